#!/usr/bin/env python3
"""Diagnostic (GPU box): is a large-case index mismatch caused by the factor or by the loop's summation order?"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from oracle import grid, obq_ref, scaling_ref
from sleekit_amd import codebook, obq, scaling, synth, engine, _lib

R, n, seed, levels, strip = [int(x) for x in sys.argv[1:6]]
L = synth.make_layer(R, n, seed)
g = grid.UniformGrid(levels, -1, 1); cb = codebook.UniformCodebook(levels, -1, 1)
H = obq_ref.strip_input_mean(L["H"], L["mean"]) if strip else L["H"]
Ws = scaling_ref.divide_rows(L["W"], L["scale"], 0)
Qo, order, Uo, Eo = obq_ref.quantize_layer_debug(Ws, H, g)
Hd = H.astype(np.float32) + 0.01 * H.diagonal().mean() * np.eye(n)
P = Hd[order][:, order]
o2, Up, info = engine.factorize(torch.from_numpy(H).cuda(), n, 0.01, _lib.ORDER_DIAG)
Up = Up.cpu().numpy(); assert np.array_equal(o2.cpu().numpy(), order)
Uo_t = np.triu(Uo)
rel = np.abs(Up - Uo_t) / np.maximum(np.abs(Uo_t), 1e-300)
big = np.abs(Uo_t) > 1e-6 * np.abs(Uo_t).max()
res = lambda U: float(np.abs(U @ P @ U.T - np.eye(n)).max())
out = dict(n=n, cond_P=float(np.linalg.cond(P)), max_abs_diff_over_max=float(np.abs(Up - Uo_t).max() / np.abs(Uo_t).max()),
           median_rel=float(np.median(rel[big])), p99_rel=float(np.quantile(rel[big], 0.99)),
           resid_product=res(Up), resid_lapack=res(Uo_t))
# loop with the ORACLE's factor: isolates summation order from factor differences
Q1 = Ws[:, order].copy(); E1 = np.zeros_like(Q1)
obq._quantize_opt_block(Q1, E1, Uo, cb, 32, 8)
Qo_p = Qo[:, order]
out["loop_with_oracle_U: values differing"] = int((Q1 != Qo_p).sum())
out["loop_with_oracle_U: E 1ulp diffs"] = int((E1 != Eo).sum())
# loop with the PRODUCT's factor run through the ORACLE loop
Q2 = Ws[:, order].copy(); E2 = np.zeros_like(Q2)
obq_ref.run_schedule(Q2, E2, Up, g, obq_ref.block_schedule(n))
out["oracle_loop_with_product_U: values differing"] = int((Q2 != Qo_p).sum())
Q3 = Ws[:, order].copy(); E3 = np.zeros_like(Q3)
obq._quantize_opt_block(Q3, E3, Up, cb, 32, 8)
out["product loop vs oracle loop, both with product U"] = int((Q3 != Q2).sum())
print(json.dumps(out))
