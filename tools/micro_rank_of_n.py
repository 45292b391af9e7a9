"""What ONE rank of an N-GPU run does per step, timed on this one GPU through the real scheduling code
(sleekit_amd.dist.quantize_stream with dist.rehearse = (0, N): same rounds, roots, streams and kernels; the
all-gather is replaced by handing this rank's own packed payload over locally).

    python tools/micro_rank_of_n.py [N ...] [--config cfg2|cfg3|cfg4|cfg5] [--blocks B] [--kernels]

Default workload: bench.py's headline batch (8 layers 4096 x 4096).  Prints ms per step on the rank, how many
rounds went through the batched route (HipBackend.run_round) and how many layers layer by layer, and the whole-job
rate N such ranks would reach if the exchange hid completely behind the loops -- an upper bound for
`bench.py --gpus N`, NOT a measurement of it.
"""

import os

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # as bench.py
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import bench  # noqa: E402  (WORKLOADS)
from sleekit_amd import _device as dev  # noqa: E402
from sleekit_amd import _lib, codebook, synth  # noqa: E402
from sleekit_amd import dist as sdist  # noqa: E402


def main():
    argv = sys.argv[1:]
    Ns = [int(a) for a in argv if a.isdigit()] or [1, 2, 4, 8]
    cfg = argv[argv.index("--config") + 1] if "--config" in argv else None
    blocks = int(argv[argv.index("--blocks") + 1]) if "--blocks" in argv else 0
    if "--blocks" in argv:
        Ns = [x for x in Ns if x != blocks] or Ns
    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dev.lazy_errors = True
    if cfg:
        wl = bench.WORKLOADS[cfg]
        shapes = wl["block"] * (blocks or wl["blocks"])
        levels, moves, strip = wl["levels"], wl["moves"], wl["strip"]
    else:
        shapes, levels, moves, strip = [(4096, 4096)] * 8, 8, 0, False
    cb = codebook.UniformCodebook(levels, -1, 1)
    layers = []
    for i, (R, n) in enumerate(shapes):
        lay = synth.make_layer_device(R, n, 1000 + i, device)
        if strip:
            out = torch.empty_like(lay["H"])
            _lib.check(_lib.lib.slk_hessian_strip_mean(dev.ptr(lay["H"]), dev.ptr(lay["mean"]), n, dev.ptr(out), dev.stream_handle()))
            lay["H"] = out
        layers.append({k: lay[k] for k in ("W", "H", "scale")})
    weights = float(sum(R * n for R, n in shapes))
    for N in Ns:
        nl = int(os.environ.get("NL", "0")) or 1  # bench.py's default
        nf = int(os.environ.get("NF", "0")) or (2 if N >= 4 else 3)  # bench.py's default
        backend = sdist.HipBackend(cb, "diag", 0.01, moves, with_error=True, overlap=(nf, nl))
        calls = {"round": 0, "layers_in_rounds": 0, "rows": 0}
        run_round, run_rows = backend.run_round, backend.run_rows

        def counted_round(members, *a):
            calls["round"] += 1
            calls["layers_in_rounds"] += len(members)
            return run_round(members, *a)

        def counted_rows(*a):
            calls["rows"] += 1
            return run_rows(*a)

        backend.run_round, backend.run_rows = counted_round, counted_rows
        backend.rounds_on_factor_streams = N < 4  # bench.py's default
        if os.environ.get("ROUNDS_ON_FS"):
            backend.rounds_on_factor_streams = os.environ["ROUNDS_ON_FS"] != "0"
        sdist.rehearse = (0, N) if N > 1 else None

        def step():
            sdist.quantize_stream(layers, backend, join=False)

        for _ in range(int(os.environ.get("WARMUP", "6"))):  # (the caching allocator keeps finding new (stream, size) pairs for a few steps)
            step()
        torch.cuda.synchronize()
        calls.update(round=0, layers_in_rounds=0, rows=0)
        steps = int(os.environ.get("STEPS", "5"))
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        host_ms = 1e3 * (time.perf_counter() - t0) / steps  # enqueue only (the GPU is still running)
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / steps
        dev.raise_pending()
        routes = f"{calls['round'] // steps} batched rounds ({calls['layers_in_rounds'] // steps} layers), {calls['rows'] // steps} layers one by one"
        if "--kernels" in argv:
            _lib.lib.slk_profile_reset()
            _lib.lib.slk_profile_enable(1)
            for _ in range(steps):
                step()
            torch.cuda.synchronize()
            _lib.lib.slk_profile_enable(0)
            rep = sorted(_lib.profile_report(), key=lambda k: -k["chip_ms"])
            tot = sum(k["chip_ms"] for k in rep) / steps
            print(f"  chip-time per step {tot:.3f} ms:")
            for k in rep:
                print(f"    {k['kernel']:24s} {k['launches'] // steps:5d} launches/step {k['total_ms'] / steps:8.3f} ms  chip {k['chip_ms'] / steps:7.3f} ms")
            _lib.lib.slk_profile_reset()
        print(f"{cfg or 'headline'} N={N}: {ms:9.3f} ms per step on one rank  ->  {weights / ms / 1e3:8.0f} Mweights/s whole job if the exchange hides; "
              f"{routes}; host enqueue {host_ms:.2f} ms per step", flush=True)
    sdist.rehearse = None


if __name__ == "__main__":
    main()
