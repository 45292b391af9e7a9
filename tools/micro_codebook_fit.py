#!/usr/bin/env python3
"""Micro-benchmark (GPU box): codebook training -- the statistics pass (slk_codebook_stats) against the HBM roofline
(4 B per element, read twice: max|x| then the statistics), the sort, and a whole lloyd_max against the CPU oracle
on a bounded sample."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from sleekit_amd import _lib, codebook

n = 4096 * 4096
torch.manual_seed(1)
x = torch.randn(n, device="cuda")
x[::97] *= 4
xs = codebook._sorted(x)
for tag, data in (("unsorted", x), ("sorted", xs)):
    for levels in (4, 16, 256):
        cb = codebook.Codebook(np.linspace(-3, 3, levels))
        for _ in range(2):
            cb._stats(data)
        torch.cuda.synchronize()
        _lib.lib.slk_profile_reset(); _lib.lib.slk_profile_enable(1)
        for _ in range(5):
            cb._stats(data)
        torch.cuda.synchronize()
        _lib.lib.slk_profile_enable(0)
        line = [f"{tag:9s} {levels:3d} levels:"]
        for k in _lib.profile_report():
            us = 1e3 * k["total_ms"] / k["launches"]
            line.append(f"{k['kernel']} {us:7.1f} us" + (f" ({k['bytes'] / k['launches'] / us / 1e3:5.0f} GB/s)" if k["bytes"] / k["launches"] > 1e6 else ""))
        print("  ".join(line))
        _lib.lib.slk_profile_reset()
_lib.lib.slk_profile_enable(1)
for _ in range(3):
    codebook._sorted(x)
torch.cuda.synchronize()
_lib.lib.slk_profile_enable(0)
for k in _lib.profile_report():
    print(f"{k['kernel']}: {1e3 * k['total_ms'] / k['launches']:.1f} us for {n} keys")
for levels, lam in ((16, 0.0), (16, 0.05), (256, 0.0)):
    codebook.lloyd_max(x, levels, lam, max_iter=2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    cb = codebook.lloyd_max(x, levels, lam)
    torch.cuda.synchronize()
    t = time.perf_counter() - t0
    print(f"lloyd_max({n} samples, {levels} levels, lagrange {lam}): {t * 1e3:.1f} ms, {len(cb)} levels kept, mse {float(cb.mse(x)):.6f}")
# CPU oracle on a bounded sample (1M points), rounds counted
from oracle import codebook_fit as fit
sample = x[: 1 << 20].cpu().numpy()
rounds = []
t0 = time.perf_counter()
g = fit.fit_lloyd_max(sample, 16, rounds=rounds)
t = time.perf_counter() - t0
t0 = time.perf_counter()
cb = codebook.lloyd_max(x[: 1 << 20], 16)
torch.cuda.synchronize()
tg = time.perf_counter() - t0
print(f"1M samples, 16 levels: CPU oracle {t:.2f} s ({rounds[0]} rounds), GPU {tg * 1e3:.1f} ms; max |value difference| {np.abs(g.values - cb.values).max():.2e}")
