#!/usr/bin/env python3
"""Micro-benchmark (GPU box): cycles per column step of the leaf chain in isolation."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sleekit_amd import _lib
out = torch.zeros(4, dtype=torch.float64, device="cuda")
for wps in (1, 2, 3, 4, 12, 52, 22, 32, 42):
    for rep in range(2):
        _lib.check(_lib.lib.slk_probe_leaf_chain(out.data_ptr(), 2000, wps, None)); torch.cuda.synchronize()
    cyc, slow = out[0].item(), out[3].item()
    label = {1: "1 wave/SIMD", 2: "2 leaf waves/SIMD", 3: "3 leaf waves/SIMD", 4: "4 leaf waves/SIMD", 52: "+ bfloat16 MFMA companion", 12: "+ MFMA companion", 22: "+ fma-chain companion", 32: "+ LDS-read companion", 42: "leaf waves 0,1,4,5 + MFMA waves 2,3,6,7"}[wps]
    print(f"leaf chain, {label}: {cyc/2000/32:7.1f} cycles per column step ({cyc/2000/2400:.2f} us per 32-column leaf); slowest leaf wave {slow/2000/32:7.1f} cycles")
