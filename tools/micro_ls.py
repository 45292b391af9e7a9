#!/usr/bin/env python3
"""Micro-benchmark (GPU box): the local search (a12 + a13) per kernel, on BLOOM-560M's layer shapes with 10 moves."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sleekit_amd import _lib, codebook, engine, synth

cb = codebook.UniformCodebook(8, -1, 1)
for R, n in ((4096, 1024), (3072, 1024), (1024, 1024), (1024, 4096), (4096, 4096)):
    L = synth.make_layer_device(R, n, 1006, torch.device("cuda"))
    res = engine.quantize_layer(L["W"], L["H"], cb, L["scale"], unscale=False)
    Ws = engine.rows_divide(L["W"], L["scale"])
    abi = cb._abi()
    for rep in range(2):
        Q = res.Q.clone()
        engine.local_search(Ws, Q, L["H"], abi, 10)
    torch.cuda.synchronize()
    _lib.lib.slk_profile_reset(); _lib.lib.slk_profile_enable(1)
    for rep in range(3):
        Q = res.Q.clone()
        engine.local_search(Ws, Q, L["H"], abi, 10)
    torch.cuda.synchronize()
    _lib.lib.slk_profile_enable(0)
    print(f"{R} x {n}, 10 moves:")
    for k in _lib.profile_report():
        print(f"   {k['kernel']:<18s} avg {1e3 * k['total_ms'] / k['launches']:8.2f} us   {k['bytes'] / max(k['total_ms'], 1e-9) / 1e6:8.1f} GB/s algorithmic")
    _lib.lib.slk_profile_reset()
