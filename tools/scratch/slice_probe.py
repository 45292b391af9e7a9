import time, torch, sys, os
sys.path.insert(0, os.getcwd())
ws = [torch.randn(768, 768, device="cuda") for _ in range(8)]
torch.cuda.synchronize()
def t(f, n=200):
    t0 = time.perf_counter()
    for _ in range(n): f()
    return 1e6 * (time.perf_counter() - t0) / n
print("slice x8 (idle GPU):", t(lambda: [w[0:768] for w in ws]), "us")
print("stack x8 (idle GPU):", t(lambda: torch.stack([w[0:768] for w in ws])), "us")
torch.cuda.synchronize()
# with a deep queue of pending kernels
big = torch.randn(8192, 8192, device="cuda")
for _ in range(50): big = big @ big * 1e-4
print("slice x8 (busy GPU):", t(lambda: [w[0:768] for w in ws]), "us")
print("stack x8 (busy GPU):", t(lambda: torch.stack([w[0:768] for w in ws]), 20), "us")
torch.cuda.synchronize()
from sleekit_amd import _lib, _device as dev
print("stream_handle:", t(lambda: dev.stream_handle()), "us;  torch.empty:", t(lambda: torch.empty((8, 768, 768), device="cuda")), "us")
