#!/usr/bin/env python3
"""Micro-benchmark (GPU box): what rocBLAS DGEMM reaches on the shapes of the trailing update (for scale)."""
import torch
for (M, K, N) in ((4096, 512, 3584), (4096, 512, 2048), (4096, 512, 512), (4096, 4096, 4096), (4096, 256, 4096)):
    A = torch.randn(M, K, device="cuda", dtype=torch.float64); B = torch.randn(K, N, device="cuda", dtype=torch.float64)
    for _ in range(3): C = A @ B
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): C = A @ B
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"rocBLAS DGEMM {M}x{K}x{N}: {ms*1e3:8.1f} us  {2.0*M*K*N/ms/1e9:6.1f} TFLOP/s")
