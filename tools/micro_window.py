#!/usr/bin/env python3
"""Micro-benchmark (GPU box): where workgroup 0 of the window kernel spends its cycles (run with SLK_WIN_DBG=8)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sleekit_amd import _lib, engine, codebook
R, n = int(os.environ.get('ROWS', '4096')), int(os.environ.get('COLS', '4096'))
g = torch.Generator(device="cuda").manual_seed(1)
W = torch.randn(R, n, device="cuda", generator=g) * 0.5
U = torch.triu(torch.randn(n, n, device="cuda", dtype=torch.float64, generator=g) * 0.01) + torch.eye(n, device="cuda", dtype=torch.float64)
order = torch.arange(n, device="cuda")
cb = codebook.UniformCodebook(8, -1, 1)._abi()
buf = (ctypes.c_longlong * 80)()
engine.run_loop(W, None, order, U, cb, 32, 8)
_lib.check(_lib.lib.slk_probe_window_cycles(buf, 1))
reps = 3
for rep in range(reps):
    engine.run_loop(W, None, order, U, cb, 32, 8)
_lib.check(_lib.lib.slk_probe_window_cycles(buf, 1))
names2 = {0: "prologue (chain wave 0 / first helper)", 1: "chain: leaves incl. loads/stores", 2: "chain: local updates", 3: "chain: barrier wait", 4: "helper: block staging issue", 10: "helper: table fetch", 11: "helper: rest of tile", 5: "helper: deferred update", 6: "helper: table write", 7: "helper: barrier wait", 9: "kernel total"}
names = {0: "leaf chain (wave 0)", 1: "stage + barrier before leaf", 2: "barrier after leaf (wave 0)", 3: "urgent update (wave 0)",
         4: "barrier after update", 5: "deferred update (wave 4)", 6: "wave 4: barrier after its part", 7: "tile load", 8: "tile store", 9: "kernel total", 10: "update pass set-up (wave 0)", 11: "aux 11", 12: "aux 12"}
launches = reps * int(os.environ.get('WINDOWS', '8'))
import os
if not os.environ.get("SLK_NO_WINDOW2"): names = names2
for k, name in names.items():
    print(f"  {name:<32s} {buf[k] / launches / 2400.0:8.2f} us per window launch")

if not os.environ.get("SLK_NO_WINDOW2"):
    print("  busy us per period (chain wave 0 | first helper wave):")
    for p in range(8):
        print(f"    period {p}: {buf[16 + p] / launches / 2400.0:6.2f} | {buf[48 + p] / launches / 2400.0:6.2f}")
