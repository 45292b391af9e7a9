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
_lib.set_option("panel_split", int(os.environ.get("FORM", "3")))
engine.factorize(L["H"], n, 0.01, _lib.ORDER_DIAG)
_lib.check(_lib.lib.slk_probe_panel_cycles(buf, 1))
for _ in range(3):
    engine.factorize(L["H"], n, 0.01, _lib.ORDER_DIAG)
_lib.check(_lib.lib.slk_probe_panel_cycles(buf, 1))
form = int(os.environ.get("FORM", "3"))
if form in (0, 3):  # the chain of round 4: wave 0 of the workgroups r >= 1, from seeing X_{r-1} to publishing X_r
    names = ["X_{r-1}: memory -> LDS", "L(r, r-1) = C inv(L)^T", "update of T_r", "L(r, r-1) drained + row flag", "T_r -> its image", "strips + inverse",
             "X_r stored, drained, published"]
else:
    names = ["staging (loads -> LDS)", "diagonal-tile update", "pivot chains (4 strips)", "barriers + next-strip blocks", "tail to the last barrier", "last L21 column + stores"]
launches = max(1, buf[15])
tot = 0.0
for k, name in enumerate(names):
    us = buf[k] / launches / 2400.0
    tot += us
    print(f"  {name:<32s} {us:7.2f} us per panel")
print(f"  {'sum':<32s} {tot:7.2f} us   ({launches} launches counted)")
