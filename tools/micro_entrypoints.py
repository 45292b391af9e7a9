#!/usr/bin/env python3
"""Per-kernel breakdown of the entry points that the headline bench does not time (GPU box): scale searches, local
search, statistics, the other orders.  Looks for small kernels that cost more than the work they wrap."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from sleekit_amd import _lib, codebook, obq, scaling, engine

R = n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
W = torch.randn(R, n, device="cuda") * 0.02
X = torch.randn(2 * n, n, device="cuda")
H = ((X.t() @ X) / (2 * n)).contiguous()
H = ((H + H.t()) * 0.5).contiguous()
cb = codebook.UniformCodebook(8, -1, 1)
sc = scaling.compute_scaling(W, cb, H=H, mode="mse")


def profile(name, f):
    f()
    torch.cuda.synchronize()
    _lib.lib.slk_profile_reset()
    _lib.lib.slk_profile_enable(1)
    t0 = time.perf_counter()
    f()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    _lib.lib.slk_profile_enable(0)
    rep = sorted(_lib.profile_report(), key=lambda k: -k["total_ms"])
    tot = sum(k["total_ms"] for k in rep)
    print(f"{name}: wall {1e3 * dt:8.2f} ms, kernels {tot:8.2f} ms")
    for k in rep[:7]:
        print(f"     {k['kernel']:24s} {k['launches']:5d} x {1e3 * k['total_ms'] / k['launches']:9.1f} us = {k['total_ms']:8.2f} ms")


profile("scaling mse", lambda: scaling.compute_scaling(W, cb, H=H, mode="mse"))
profile("scaling diag", lambda: scaling.compute_scaling(W, cb, H=H, mode="diag"))
profile("scaling hessian (grid 20)", lambda: scaling.compute_scaling(W, cb, H=H, mode="hessian", grid_size=20))
profile("scaling obq (grid 10)", lambda: scaling.compute_scaling(W, cb, H=H, mode="obq", grid_size=10))
profile("quantize_with_scaling diag", lambda: scaling.quantize_with_scaling(W, sc, cb, H))
profile("quantize_with_scaling + 10 moves", lambda: scaling.quantize_with_scaling(W, sc, cb, H, nb_ls_moves=10))
profile("quantize_with_scaling sqerr", lambda: scaling.quantize_with_scaling(W, sc, cb, H, act_order="sqerr"))
profile("quantize_with_scaling inv_diag", lambda: scaling.quantize_with_scaling(W, sc, cb, H, act_order="inv_diag"))
m = torch.randn(n, device="cuda")
profile("remove_input_bias", lambda: obq.remove_input_bias(H, m))
profile("quantization_error", lambda: obq.quantization_error(W, W * 0.9, H))
