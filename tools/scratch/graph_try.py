import os, sys, faulthandler
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
faulthandler.enable()
import numpy as np, torch
from sleekit_amd import codebook, graphs, synth, engine
mode = sys.argv[1] if len(sys.argv) > 1 else "a"
cb = codebook.UniformCodebook(8, -1, 1)
R, n = 96, 172
g = graphs.GraphedLayer(R, n, cb)
if mode == "a":
    g.capture()
for seed in (2003, 2004, 2003):
    L = synth.make_layer(R, n, seed)
    W, H, sc = (torch.from_numpy(L[k]).cuda() for k in ("W", "H", "scale"))
    print("replay", seed, flush=True)
    g(W, H, sc)
    torch.cuda.synchronize()
    print("synced", flush=True)
    g.check()
    res = engine.quantize_layer(W, H, cb, sc)
    err = engine.row_errors(W, res.Q, H)
    print(seed, torch.equal(g.Q, res.Q), torch.equal(g.idx, res.idx), torch.equal(g.row_err, err), flush=True)
