"""Which torch streams share a hardware queue?  A long kernel on stream i, then a short one on stream j: if the short
one finishes only after the long one, they are in the same (in-order) hardware queue."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
n_streams = int(sys.argv[1]) if len(sys.argv) > 1 else 10
streams = [torch.cuda.Stream() for _ in range(n_streams)]
a = torch.randn(8192, 8192, device="cuda")
small = torch.zeros(16, device="cuda")
torch.cuda.synchronize()
def long_op():
    for _ in range(6):
        torch.mm(a, a)
names = ["default"] + [f"s{i}" for i in range(n_streams)]
objs = [torch.cuda.default_stream()] + streams
for st in objs:  # touch every stream once
    with torch.cuda.stream(st):
        small.add_(1)
torch.cuda.synchronize()
share = {}
for i, si in enumerate(objs):
    row = []
    for j, sj in enumerate(objs):
        if i == j:
            row.append("-")
            continue
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(si):
            long_op()
        with torch.cuda.stream(sj):
            e0.record()
            small.add_(1)
            e1.record()
        t0 = time.perf_counter()
        e1.synchronize()
        waited = time.perf_counter() - t0
        torch.cuda.synchronize()
        row.append("X" if waited > 0.01 else ".")
    print(f"{names[i]:8s} " + " ".join(row), flush=True)
