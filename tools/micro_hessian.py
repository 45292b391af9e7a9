import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from sleekit_amd import _device as dev, _lib
n, T = 4096, 2048
X = torch.randn(T, n, device="cuda")
H = torch.zeros(n, n, device="cuda"); mean = torch.zeros(n, device="cuda")
ws, wsb = dev.workspace(T, n)
def run():
    _lib.check(_lib.lib.slk_hessian_accumulate(H.data_ptr(), mean.data_ptr(), X.data_ptr(), n, T, 0, ws.data_ptr(), wsb, None))
for _ in range(3): run()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): run()
torch.cuda.synchronize()
print(f"hessian accumulate n={n} T={T}: {(time.perf_counter()-t0)/20*1e3:.3f} ms")
_lib.lib.slk_profile_reset()
_lib.lib.slk_profile_enable(1)
for _ in range(5): run()
torch.cuda.synchronize()
_lib.lib.slk_profile_enable(0)
for k in sorted(_lib.profile_report(), key=lambda k: -k["total_ms"]):
    print(f"   {k['kernel']:22s} {k['launches'] // 5:3d}/call  avg {1e3 * k['total_ms'] / k['launches']:8.2f} us")
