#!/usr/bin/env python3
"""Micro-benchmark (GPU box): the local search (a12 + a13) per kernel, on BLOOM-560M's layer shapes with 10 moves.
SHAPES=4096x4096,1024x4096 MOVES=0,1,10,100 choose other cases (moves = 0: the fixed part of the search kernel)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sleekit_amd import _lib, codebook, engine, synth

cb = codebook.UniformCodebook(8, -1, 1)
SHAPES = [tuple(int(x) for x in sh.split("x")) for sh in os.environ.get("SHAPES", "4096x1024,3072x1024,1024x1024,1024x4096,4096x4096").split(",")]
MOVES = [int(x) for x in os.environ.get("MOVES", "10").split(",")]
for R, n in SHAPES:
    L = synth.make_layer_device(R, n, 1006, torch.device("cuda"))
    res = engine.quantize_layer(L["W"], L["H"], cb, L["scale"], unscale=False)
    Ws = engine.rows_divide(L["W"], L["scale"])
    abi = cb._abi()
    for moves in MOVES:
        for rep in range(2):
            Q = res.Q.clone()
            engine.local_search(Ws, Q, L["H"], abi, moves)
        torch.cuda.synchronize()
        _lib.lib.slk_profile_reset(); _lib.lib.slk_profile_enable(1)
        for rep in range(3):
            Q = res.Q.clone()
            engine.local_search(Ws, Q, L["H"], abi, moves)
        torch.cuda.synchronize()
        _lib.lib.slk_profile_enable(0)
        print(f"{R} x {n}, {moves} moves:")
        for k in _lib.profile_report():
            print(f"   {k['kernel']:<18s} avg {1e3 * k['total_ms'] / k['launches']:8.2f} us   {k['bytes'] / max(k['total_ms'], 1e-9) / 1e6:8.1f} GB/s algorithmic")
        _lib.lib.slk_profile_reset()
