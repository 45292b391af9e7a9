"""Latency of the pieces one rank of an 8-GPU run executes, each alone on an idle GPU (4096 x 4096 layers)."""

import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from sleekit_amd import _device as dev  # noqa: E402
from sleekit_amd import codebook, synth  # noqa: E402
from sleekit_amd import dist as sdist  # noqa: E402


def timed(name, f, reps=5):
    f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    tot = time.perf_counter() - t0
    print(f"{name:34s} {1e3 * tot / reps:8.3f} ms   (host enqueue {1e3 * host / reps:6.3f} ms)", flush=True)


def main():
    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dev.lazy_errors = True
    R = n = 4096
    cb = codebook.UniformCodebook(8, -1, 1)
    lay = synth.make_layer(R, n, 1000, device=device)
    layer = {k: torch.from_numpy(lay[k]).to(device) for k in ("W", "H", "scale")}
    backend = sdist.HipBackend(cb, "diag", 0.01, 0, with_error=True, overlap=False)
    words = backend.payload_words(n)
    fac = backend.factorize(layer)
    payload = backend.pack(fac, words)
    timed("factorize", lambda: backend.factorize(layer))
    timed("pack", lambda: backend.pack(fac, words))
    timed("unpack", lambda: backend.unpack(payload, n))
    for rows in (4096, 2048, 1024, 512):
        timed(f"loop + error, {rows} rows", lambda: backend.run_rows(layer, 0, rows, fac))
    eng = backend.engine
    W = layer["W"][:512].contiguous()
    sc = layer["scale"][:512].contiguous()
    timed("  loop only, 512 rows", lambda: eng.quantize_layer(W, layer["H"], cb, sc, factor=fac))
    res = eng.quantize_layer(W, layer["H"], cb, sc, factor=fac)
    timed("  error only, 512 rows", lambda: eng.row_errors(W, res.Q, layer["H"]))
    if "--kernels" in sys.argv:
        from sleekit_amd import _lib
        _lib.lib.slk_profile_reset()
        _lib.lib.slk_profile_enable(1)
        backend.run_rows(layer, 0, 512, fac)
        torch.cuda.synchronize()
        _lib.lib.slk_profile_enable(0)
        for k in _lib.profile_report():
            print(f"    {k['kernel']:24s} {k['launches']:4d} launches {k['total_ms']:8.3f} ms")


if __name__ == "__main__":
    main()
