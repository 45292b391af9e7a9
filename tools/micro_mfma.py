#!/usr/bin/env python3
"""Micro-benchmark (GPU box): issue-bound MFMA rates that anchor the compute rooflines."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sleekit_amd import _lib
sink64 = torch.zeros(16, dtype=torch.float64, device="cuda"); sink32 = torch.zeros(16, dtype=torch.float32, device="cuda")
for blocks in (256, 512, 1024, 2048):
    for name, fn, sink, flop in (("f64 16x16x4", _lib.lib.slk_probe_mfma_f64, sink64, 2048), ("f32 32x32x2", _lib.lib.slk_probe_mfma_f32, sink32, 4096)):
        iters = 4000
        fn(sink.data_ptr(), blocks, iters, None); torch.cuda.synchronize()
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        t0.record(); fn(sink.data_ptr(), blocks, iters, None); t1.record(); torch.cuda.synchronize()
        ms = t0.elapsed_time(t1)
        total = blocks * 4 * iters * 4 * flop
        print(f"{name}  blocks={blocks:5d}: {total/ms/1e9:8.2f} TFLOP/s  ({ms:.3f} ms)")

for nacc in (8, 16):
    for blocks in (256, 512, 1024, 2048):
        iters = 2000
        _lib.lib.slk_probe_mfma_f64_acc(sink64.data_ptr(), blocks, iters, nacc, None); torch.cuda.synchronize()
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        t0.record(); _lib.lib.slk_probe_mfma_f64_acc(sink64.data_ptr(), blocks, iters, nacc, None); t1.record(); torch.cuda.synchronize()
        ms = t0.elapsed_time(t1)
        print(f"f64 16x16x4 {nacc} accumulators blocks={blocks:5d}: {blocks*4*iters*nacc*2048/ms/1e9:8.2f} TFLOP/s ({ms:.3f} ms)")
