"""What ONE rank of an N-GPU run does per step, timed on this one GPU through the real scheduling code
(bench.py's own Leg -- its inputs, its streams for N ranks, its step and its flow control -- with
sleekit_amd.dist.rehearse = (0, N): same rounds, roots, streams and kernels; the all-gather is replaced by handing this
rank's own packed payload over locally).

    python tools/micro_rank_of_n.py [N ...] [--config cfg2|cfg3|cfg4|cfg5] [--blocks B] [--kernels]

Environment: STEPS (20), WARMUP (5) -- the driver's; NF / NL / ROUNDS_ON_FS override the stream layout.
Default workload: bench.py's headline batch (8 layers 4096 x 4096).  Prints ms per step on the rank, how many
rounds went through the batched route (HipBackend.run_round) and how many layers layer by layer, and the whole-job
rate N such ranks would reach if the exchange hid completely behind the loops -- an upper bound for
`bench.py --gpus N`, NOT a measurement of it.
"""

import os

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # as bench.py
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import bench  # noqa: E402  (WORKLOADS, Env, Leg)
from sleekit_amd import _device as dev  # noqa: E402
from sleekit_amd import _lib  # noqa: E402
from sleekit_amd import dist as sdist  # noqa: E402


def main():
    argv = sys.argv[1:]
    Ns = [int(a) for a in argv if a.isdigit()] or [1, 2, 4, 8]
    cfg = argv[argv.index("--config") + 1] if "--config" in argv else None
    blocks = int(argv[argv.index("--blocks") + 1]) if "--blocks" in argv else 0
    if "--blocks" in argv:
        Ns = [x for x in Ns if x != blocks] or Ns
    torch.cuda.set_device(0)
    dev.lazy_errors = True
    if cfg:
        wl = bench.WORKLOADS[cfg]
        shapes = wl["block"] * (blocks or wl["blocks"])
        levels, moves, strip = wl["levels"], wl["moves"], wl["strip"]
    else:
        shapes, levels, moves, strip = [(4096, 4096)] * 8, 8, 0, False
    weights = float(sum(R * n for R, n in shapes))
    steps, warmup = int(os.environ.get("STEPS", "20")), int(os.environ.get("WARMUP", "5"))
    env = bench.Env(None)
    env.device = torch.device("cuda", 0)
    for N in Ns:
        streams = None
        if os.environ.get("NF") or os.environ.get("NL"):
            streams = (int(os.environ.get("NF", "0")) or (2 if N >= 4 else 3), int(os.environ.get("NL", "0")) or 1)
        env.world = N  # the leg lays its streams out for N ranks ...
        leg = bench.Leg(env, cfg or "headline", shapes, levels, moves, strip, streams=streams)
        env.world = 1  # ... and is timed without a process group
        backend = leg.backend
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
        if os.environ.get("ROUNDS_ON_FS"):
            backend.rounds_on_factor_streams = os.environ["ROUNDS_ON_FS"] != "0"
        sdist.rehearse = (0, N) if N > 1 else None
        for _ in range(warmup):
            leg.step()
        env.fence()
        calls.update(round=0, layers_in_rounds=0, rows=0)
        elapsed, _ = leg.timed(steps, 0)
        ms = 1e3 * elapsed / steps
        dev.raise_pending()
        routes = f"{calls['round'] // steps} batched rounds ({calls['layers_in_rounds'] // steps} layers), {calls['rows'] // steps} layers one by one"
        if "--kernels" in argv:
            _lib.lib.slk_profile_reset()
            _lib.lib.slk_profile_enable(1)
            for _ in range(steps):
                leg.step()
            torch.cuda.synchronize()
            _lib.lib.slk_profile_enable(0)
            rep = sorted(_lib.profile_report(), key=lambda k: -k["chip_ms"])
            tot = sum(k["chip_ms"] for k in rep) / steps
            print(f"  chip-time per step {tot:.3f} ms:")
            for k in rep:
                print(f"    {k['kernel']:24s} {k['launches'] // steps:5d} launches/step {k['total_ms'] / steps:8.3f} ms  chip {k['chip_ms'] / steps:7.3f} ms")
            _lib.lib.slk_profile_reset()
        print(f"{cfg or 'headline'} N={N}: {ms:9.3f} ms per step on one rank  ->  {weights / ms / 1e3:8.0f} Mweights/s whole job if the exchange hides; "
              f"{routes}; host enqueue {leg.host_ms_per_step:.2f} ms per step; device mallocs while timed {leg.device_mallocs_while_timed}", flush=True)
        leg.release()
    sdist.rehearse = None


if __name__ == "__main__":
    main()
