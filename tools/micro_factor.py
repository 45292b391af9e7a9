#!/usr/bin/env python3
"""Micro-benchmark (GPU box): per-kernel times of the factorisation at several n."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from sleekit_amd import _lib, engine
for n in [int(x) for x in sys.argv[1:]] or [64, 128, 512, 4096]:
    rng = np.random.default_rng(n)
    A = rng.standard_normal((n, n + 8)).astype(np.float32)
    H = torch.from_numpy((A @ A.T / n + 0.1 * np.eye(n)).astype(np.float32)).cuda()
    for rep in range(3):
        engine.factorize(H, n, 0.01, _lib.ORDER_DIAG)
    torch.cuda.synchronize()
    _lib.lib.slk_profile_reset(); _lib.lib.slk_profile_enable(1)
    for rep in range(5):
        engine.factorize(H, n, 0.01, _lib.ORDER_DIAG)
    torch.cuda.synchronize()
    _lib.lib.slk_profile_enable(0)
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for rep in range(5):
        engine.factorize(H, n, 0.01, _lib.ORDER_DIAG)
    t1.record(); torch.cuda.synchronize()
    print(f"n={n}: whole factorize {t0.elapsed_time(t1)/5*1e3:.1f} us (unprofiled)")
    for k in _lib.profile_report():
        print(f"   {k['kernel']:<18s} {k['launches']//5:4d}/call  avg {1e3*k['total_ms']/k['launches']:8.2f} us  total/call {k['total_ms']/5:8.3f} ms")
    _lib.lib.slk_profile_reset()
