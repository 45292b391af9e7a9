#!/usr/bin/env python3
"""Where the HOST spends its time enqueueing a step (cProfile around sleekit_amd.dist.quantize_stream; GPU box).

    python tools/host_profile.py [N] [--config cfgK]       (N > 1: one rank of N, rehearsed as tools/micro_rank_of_n.py does)
"""
import cProfile
import os
import pstats
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

import bench  # noqa: E402
from sleekit_amd import _device as dev  # noqa: E402
from sleekit_amd import codebook, synth  # noqa: E402
from sleekit_amd import dist as sdist  # noqa: E402


def main():
    argv = sys.argv[1:]
    N = next((int(a) for a in argv if a.isdigit()), 1)
    cfg = argv[argv.index("--config") + 1] if "--config" in argv else "cfg2"
    wl = bench.WORKLOADS[cfg]
    shapes = wl["block"] * wl["blocks"]
    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dev.lazy_errors = True
    cb = codebook.UniformCodebook(wl["levels"], -1, 1)
    layers = [{k: v for k, v in synth.make_layer_device(R, n, 1000 + i, device).items() if k in ("W", "H", "scale")} for i, (R, n) in enumerate(shapes)]
    backend = sdist.HipBackend(cb, "diag", 0.01, wl["moves"], with_error=True, overlap=(2, 1) if N >= 4 else (3, 1))
    backend.rounds_on_factor_streams = N < 4
    sdist.rehearse = (0, N) if N > 1 else None
    for _ in range(2):
        sdist.quantize_stream(layers, backend, join=False)
    torch.cuda.synchronize()
    import gc
    gc.collect()
    gc.freeze()  # (as bench.py: a full collection inside the profile would land on whichever line allocates)
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    for _ in range(3):
        sdist.quantize_stream(layers, backend, join=False)
        torch.cuda.synchronize()  # the queue never fills: what is timed is the enqueueing
    pr.disable()
    print(f"{cfg} N={N}: {1e3 * (time.perf_counter() - t0) / 3:.2f} ms per step with a synchronize after each")
    pstats.Stats(pr).sort_stats("cumulative").print_stats(40)
    pstats.Stats(pr).sort_stats("tottime").print_stats(30)


if __name__ == "__main__":
    main()
