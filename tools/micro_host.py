#!/usr/bin/env python3
"""Micro-benchmark (GPU box): how long the HOST needs to enqueue one step of the bench (8 layers) against how
long the GPU needs to run it.  If the two are close, launch overhead, not the kernels, sets the throughput."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sleekit_amd import codebook, dist as sdist, synth
torch.cuda.set_device(0)
device = torch.device("cuda", 0)
L, R, n = 8, 4096, 4096
base = [synth.make_layer_device(R, n, 1000 + i, device) if hasattr(synth, "make_layer_device") else None for i in range(2)]
if base[0] is None:
    import numpy as np
    base = []
    for i in range(2):
        W = torch.randn(R, n, device=device) * 0.02
        X = torch.randn(2 * n, n, device=device)
        H = (X.t() @ X) / (2 * n)
        base.append(dict(W=W, H=H.contiguous(), scale=(W.abs().amax(dim=1) * 0.6).contiguous()))
layers = [base[i % 2] for i in range(L)]
cb = codebook.UniformCodebook(8, -1, 1)
backend = sdist.HipBackend(cb, "diag", 0.01, 0, with_error=True, overlap=(3, 2))
for _ in range(2):
    sdist.quantize_stream(layers, backend)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    sdist.quantize_stream(layers, backend)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"step {rep}: host enqueue {1e3 * (t1 - t0):7.2f} ms, until the GPU is done {1e3 * (t2 - t0):7.2f} ms")
