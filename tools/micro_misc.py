#!/usr/bin/env python3
"""Micro-benchmark (GPU box): local search, scale searches and Hessian accumulation at 4096 x 4096."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from sleekit_amd import _lib, engine, codebook, scaling, synth, statistics
R = n = 4096
dev = torch.device("cuda")
L = synth.make_layer(R, n, 1000, device=dev)
W, H, sc = (torch.from_numpy(L[k]).to(dev) for k in ("W", "H", "scale"))
cb = codebook.UniformCodebook(8, -1, 1)
def timed(name, fn, reps=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    print(f"{name:<44s} {1e3*(time.perf_counter()-t0)/reps:9.3f} ms")
res = engine.quantize_layer(W, H, cb, sc, unscale=False)
Ws = engine.rows_divide(W, sc)
for moves in (10, 100):
    def ls():
        Q = res.Q.clone(); engine.local_search(Ws, Q, H, cb._abi(), moves)
    timed(f"local search {moves} moves (incl. gain GEMM)", ls)
timed("quantize_layer, 3-bit, no LS", lambda: engine.quantize_layer(W, H, cb, sc))
timed("quantize_layer, 3-bit, 10 moves", lambda: engine.quantize_layer(W, H, cb, sc, nb_ls_moves=10))
timed("compute_min_mse_scaling mse, grid 100", lambda: scaling.compute_min_mse_scaling(W, cb))
timed("compute_min_mse_scaling diag, grid 100", lambda: scaling.compute_min_mse_scaling(W, cb, H=H.diagonal().contiguous()))
timed("compute_min_mse_scaling full H, grid 100", lambda: scaling.compute_min_mse_scaling(W, cb, H=H), reps=1)
timed("compute_obq_scaling, grid 100", lambda: scaling.compute_obq_scaling(W, cb, 0, H), reps=1)
lin = torch.nn.Linear(n, 8).to(dev); st = statistics.Sleekit(lin)
X = torch.randn(2048, n, device=dev)
timed("add_batch 2048 tokens x 4096", lambda: st.add_batch(X))
print("  -> %.1f TFLOP/s (T n (n+1) flop)" % (2048 * n * (n + 1) / 1e12 / 1.0))
