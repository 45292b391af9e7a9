"""One isolated 4096 x 4096 layer through the single-layer API, six times (for a kernel trace of the last ones)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from sleekit_amd import codebook, engine, synth
dev = torch.device("cuda", 0)
lay = synth.make_layer_device(4096, 4096, 1000, dev)
cb = codebook.UniformCodebook(8, -1, 1)
side = torch.cuda.Stream() if os.environ.get("STREAM") else torch.cuda.current_stream()
for i in range(6):
    torch.cuda.synchronize()
    t = time.perf_counter()
    with torch.cuda.stream(side):
        res = engine.quantize_layer(lay["W"], lay["H"], cb, lay["scale"], "diag", 0.01, 0)
        engine.row_errors(lay["W"], res.Q, lay["H"])
    torch.cuda.synchronize()
    print(f"{1e3 * (time.perf_counter() - t):.3f} ms", flush=True)
