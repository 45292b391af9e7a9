#!/usr/bin/env python3
"""Micro-benchmark (GPU box): latency of dependent VALU chains and the clock a lone wave runs at."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sleekit_amd import _lib
names = {0: "fma f64 dependent", 1: "8 x fma f64 independent", 2: "fma f32 dependent", 3: "rsq f64 + add", 4: "div f64 + add", 5: "div f32 + add", 6: "cvt f64->f32->f64 + mul", 7: "readlane + or + mul f64", 8: "mul + rint + add f32"}
out = torch.zeros(3, dtype=torch.float64, device="cuda")
iters = 20000
for mode, name in names.items():
    for rep in range(2):
        _lib.check(_lib.lib.slk_probe_chain(out.data_ptr(), iters, mode, None))
        torch.cuda.synchronize()
    cyc, ticks, _ = out.tolist()
    print(f"{name:<28s} {cyc/iters:7.1f} cycles/iter   clock {cyc/ticks*100:7.0f} MHz   ({ticks/100:.0f} us)")
