#!/usr/bin/env python3
"""Micro-benchmark (GPU box): the layer-error GEMM alone (4096 x 4096, symmetric H)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sleekit_amd import _lib, engine
R, n = 4096, 4096
g = torch.Generator(device="cuda").manual_seed(1)
W = torch.randn(R, n, device="cuda", generator=g)
Q = W + 0.1 * torch.randn(R, n, device="cuda", generator=g)
X = torch.randn(n, n, device="cuda", generator=g)
H = (X @ X.t()) / n
H = (H + H.t()) * 0.5
WANT_G = bool(int(os.environ.get("WANT_G", "0")))  # the full product G = (W - Q) H (the local search's), not the error alone
for rep in range(2):
    engine.row_errors(W, Q, H, want_G=WANT_G)
torch.cuda.synchronize()
_lib.lib.slk_profile_reset(); _lib.lib.slk_profile_enable(1)
for rep in range(5):
    engine.row_errors(W, Q, H, want_G=WANT_G)
torch.cuda.synchronize(); _lib.lib.slk_profile_enable(0)
for k in _lib.profile_report():
    print(f"   {k['kernel']:<18s} {k['launches']//5:4d}/call  avg {1e3*k['total_ms']/k['launches']:8.2f} us  {k['flops']/max(k['total_ms'],1e-9)/1e9:8.2f} TFLOP/s (algorithmic)")
