#!/usr/bin/env python3
"""Micro-benchmark (GPU box): where wave 0 of workgroup 1 of the panel kernel spends its cycles (run with SLK_WIN_DBG=8)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sleekit_amd import _lib, engine, synth
n = int(os.environ.get("COLS", "4096"))
L = synth.make_layer_device(8, n, 4100, torch.device("cuda"))
buf = (ctypes.c_longlong * 16)()
engine.factorize(L["H"], n, 0.01, _lib.ORDER_DIAG)
_lib.check(_lib.lib.slk_probe_panel_cycles(buf, 1))
for _ in range(3):
    engine.factorize(L["H"], n, 0.01, _lib.ORDER_DIAG)
_lib.check(_lib.lib.slk_probe_panel_cycles(buf, 1))
names = ["staging (loads -> LDS)", "diagonal-tile update", "pivot chains (4 strips)", "barriers + next-strip blocks", "tail to the last barrier", "last L21 column + stores"]
launches = max(1, buf[15])
tot = 0.0
for k, name in enumerate(names):
    us = buf[k] / launches / 2400.0
    tot += us
    print(f"  {name:<32s} {us:7.2f} us per panel launch")
print(f"  {'sum':<32s} {tot:7.2f} us   ({launches} launches counted)")
