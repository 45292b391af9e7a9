#!/usr/bin/env python3
"""Micro-benchmark (GPU box): per-kernel times of the quantize/propagate loop alone."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from sleekit_amd import _lib, engine, codebook
R, n = (int(x) for x in (sys.argv[1:3] if len(sys.argv) > 2 else (4096, 4096)))
g = torch.Generator(device="cuda").manual_seed(1)
W = torch.randn(R, n, device="cuda", generator=g) * 0.5
U = torch.triu(torch.randn(n, n, device="cuda", dtype=torch.float64, generator=g) * 0.01) + torch.eye(n, device="cuda", dtype=torch.float64)
order = torch.arange(n, device="cuda")
cb = codebook.UniformCodebook(8, -1, 1)._abi()
for rep in range(2):
    engine.run_loop(W, None, order, U, cb, 32, 8)
torch.cuda.synchronize()
_lib.lib.slk_profile_reset(); _lib.lib.slk_profile_enable(1)
for rep in range(3):
    engine.run_loop(W, None, order, U, cb, 32, 8)
torch.cuda.synchronize(); _lib.lib.slk_profile_enable(0)
for k in _lib.profile_report():
    print(f"   {k['kernel']:<18s} {k['launches']//3:4d}/call  avg {1e3*k['total_ms']/k['launches']:8.2f} us  total/call {k['total_ms']/3:8.3f} ms")
