#!/usr/bin/env python3
"""Micro-benchmark (GPU box): cycles per column step of the leaf chain in isolation."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sleekit_amd import _lib
out = torch.zeros(2, dtype=torch.float64, device="cuda")
for wps in (1, 2):
    for rep in range(2):
        _lib.check(_lib.lib.slk_probe_leaf_chain(out.data_ptr(), 2000, wps, None)); torch.cuda.synchronize()
    cyc = out[0].item()
    print(f"leaf chain, {wps} wave(s)/SIMD: {cyc/2000/32:7.1f} cycles per column step ({cyc/2000/2400:.2f} us per 32-column leaf)")
