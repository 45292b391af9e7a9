#!/usr/bin/env python3
"""Micro-benchmark (GPU box): the scale searches of sleekit/scaling.py on a 4096 x 4096 layer."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sleekit_amd import codebook, scaling
R = n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
W = torch.randn(R, n, device="cuda") * 0.02
X = torch.randn(2 * n, n, device="cuda")
H = ((X.t() @ X) / (2 * n)).contiguous()
cb = codebook.UniformCodebook(8, -1, 1)
for mode in ("max", "norm", "mse", "diag", "hessian", "obq"):
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        s = scaling.compute_scaling(W, cb, H=H, mode=mode)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"compute_scaling mode={mode:<8s} {1e3 * dt:9.2f} ms")
