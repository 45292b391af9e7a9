import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
from sleekit_amd import _lib, engine
rng = np.random.default_rng(29)
R, n = int(os.environ.get("ROWS", 2304)), int(os.environ.get("COLS", 1024))
W = torch.from_numpy(rng.standard_normal((R, n)).astype(np.float32)).cuda()
Q = W + 0.2 * torch.from_numpy(rng.standard_normal((R, n)).astype(np.float32)).cuda()
X = rng.standard_normal((2 * n, n)).astype(np.float32)
H = (X.T @ X / (2 * n)).astype(np.float32); H = ((H + H.T) * np.float32(0.5)).astype(np.float32)
Hs = torch.from_numpy(H).cuda()
e2, G2 = engine.row_errors(W, Q, Hs, want_G=True)
with _lib.option("tall_error", 2):
    for rep in range(3):
        e1, G1 = engine.row_errors(W, Q, Hs, want_G=True)
        d = (G1 - G2).abs()
        bad = (d > 0).nonzero()
        print("rep", rep, "max abs diff", float(d.max()), "count", int((d > 0).sum()), "of", d.numel(), "rel", float((d / G2.abs().clamp_min(1e-6)).max()))
        if len(bad):
            rows = torch.unique(bad[:, 0]); cols = torch.unique(bad[:, 1])
            print("  rows", rows[:10].tolist(), "... n", len(rows), " cols", cols[:10].tolist(), "... n", len(cols))
        Gd = (W - Q).double() @ Hs.double()
        print("  vs float64: big", float((G1.double() - Gd).abs().max()), "square", float((G2.double() - Gd).abs().max()))
