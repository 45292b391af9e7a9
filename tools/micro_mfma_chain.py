#!/usr/bin/env python3
"""Micro-benchmark (GPU box): float64 MFMA rate of ONE wave per SIMD against the number of independent accumulator chains."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sleekit_amd import _lib
sink64 = torch.zeros(16, dtype=torch.float64, device="cuda")
for nacc in (1, 2, 4, 8):
    for blocks in (256, 512):
        iters = 4000
        _lib.check(_lib.lib.slk_probe_mfma_f64_acc(sink64.data_ptr(), blocks, iters, nacc, None)); torch.cuda.synchronize()
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        t0.record(); _lib.lib.slk_probe_mfma_f64_acc(sink64.data_ptr(), blocks, iters, nacc, None); t1.record(); torch.cuda.synchronize()
        ms = t0.elapsed_time(t1)
        waves_per_simd = blocks / 256
        cyc = ms * 1e-3 * 2.4e9 / (iters * nacc * waves_per_simd)
        print(f"f64 16x16x4, {nacc} chains/wave, {waves_per_simd:.0f} wave(s)/SIMD: {blocks*4*iters*nacc*2048/ms/1e9:8.2f} TFLOP/s  ~{cyc:6.1f} cycles per MFMA per SIMD")
