#!/usr/bin/env python3
"""Micro-benchmark (GPU box): the greedy pivoted-Cholesky order (cold path, act_order="pivot")."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sleekit_amd import engine
for n in (1024, 4096):
    X = torch.randn(2 * n, n, device="cuda")
    H = (X.t() @ X) / (2 * n)
    engine.pivot_keys(H, n, 0.01); torch.cuda.synchronize()
    t0 = time.perf_counter()
    keys = engine.pivot_keys(H, n, 0.01); torch.cuda.synchronize()
    print(f"pivot order n={n}: {1e3 * (time.perf_counter() - t0):8.1f} ms  (first picks {torch.argsort(keys)[:6].tolist()})")
