"""What ONE rank of an N-GPU run does per step, timed on this one GPU (no transfer: the all-gather is
replaced by handing the packed payload over locally).

    python tools/micro_rank_of_n.py [N ...]

Per step of 8 layers 4096 x 4096: 8 / N factorisations + packs, 8 (N - 1) / N unpacks, 8 loops + errors of
4096 / N rows.  Prints ms per step and the whole-job rate N such ranks would reach if the exchange hid
completely behind the loops (an upper bound for bench.py --gpus N).
"""

import os

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # as bench.py
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))

from sleekit_amd import _device as dev  # noqa: E402
from sleekit_amd import codebook, synth  # noqa: E402
from sleekit_amd import dist as sdist  # noqa: E402


def main():
    Ns = [int(a) for a in sys.argv[1:] if a.isdigit()] or [1, 2, 4, 8]
    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dev.lazy_errors = True
    R = n = 4096
    L = 8
    cb = codebook.UniformCodebook(8, -1, 1)
    base = []
    for i in range(2):
        lay = synth.make_layer(R, n, 1000 + i, device=device)
        base.append({k: torch.from_numpy(lay[k]).to(device) for k in ("W", "H", "scale")})
    layers = [base[i % 2] for i in range(L)]
    for N in Ns:
        nl = int(os.environ.get("NL", "0")) or 1  # bench.py's default
        nf = int(os.environ.get("NF", "3"))
        backend = sdist.HipBackend(cb, "diag", 0.01, 0, with_error=True, overlap=(nf, nl))
        fstreams, cstream, lstreams = backend.streams()
        words = backend.payload_words(n)
        lo, hi = sdist.row_range(R, 0, N)

        only = os.environ.get("ONLY", "")
        cached = {}

        def step():
            here = torch.cuda.current_stream()
            facs, evs = {}, {}
            mine = list(range(0, L, N))
            if only == "loops" and cached:
                mine = []
                facs.update(cached["facs"])
                evs.update(cached["evs"])
            first = getattr(backend, "_rot", 0)
            backend._rot = (first + len(mine)) % len(fstreams)
            for k, l in enumerate(mine):
                fs = fstreams[(first + k) % len(fstreams)]
                with torch.cuda.stream(fs):
                    facs[l] = backend.factorize(layers[l])
                    evs[l] = torch.cuda.Event()
                    evs[l].record(fs)
            if only == "loops" and not cached:
                cached["facs"], cached["evs"] = dict(facs), dict(evs)
                mine = list(range(0, L, N))
            if only == "factor":
                return None
            mine = list(range(0, L, N))
            payloads = {}
            if N > 1:
                cevs = {}
                with torch.cuda.stream(cstream):
                    for l in mine:
                        cstream.wait_event(evs[l])
                        payloads[l] = backend.pack(facs[l], words)
                        for t in facs[l]:
                            t.record_stream(cstream)
                        cevs[l] = torch.cuda.Event()  # one hand-over per round, like the all-gather of that round
                        cevs[l].record(cstream)
                cev = cevs[mine[-1]]
            out = []
            if N > 1 and os.environ.get("BATCH", "1") != "0":
                # round by round through the batched loop, as sleekit_amd.dist does (own payload stands in for the peers')
                for g in range(L // N):
                    rot = getattr(backend, "_lrot", 0)
                    backend._lrot = (rot + 1) % len(lstreams)
                    ls = lstreams[rot]
                    if os.environ.get("SLOT", "1") != "0":  # (dist.py's default) the round runs on the stream that factored this rank's layer of it
                        ls = fstreams[(first + g) % len(fstreams)]
                    with torch.cuda.stream(ls):
                        ls.wait_event(cevs[mine[g]])
                        members = list(range(g * N, (g + 1) * N))
                        out.extend(backend.run_round([layers[l] for l in members], lo, hi, [payloads[mine[g]]] * N))
                        payloads[mine[g]].record_stream(ls)
                return out
            for l in range(L):
                ls = lstreams[l % len(lstreams)]
                with torch.cuda.stream(ls):
                    if N > 1:
                        ls.wait_event(cev)
                        root = mine[(l // N) % len(mine)] if l not in facs else l
                        f = facs[l] if l in facs else backend.unpack(payloads[root], n)
                        payloads[root].record_stream(ls)
                    else:
                        ls.wait_event(evs[l])
                        f = facs[l]
                    out.append(backend.run_rows(layers[l], lo, hi, f))
                    for t in f:
                        t.record_stream(ls)
            return out

        for _ in range(2):
            step()
        torch.cuda.synchronize()
        steps = int(os.environ.get("STEPS", "5"))
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        host_ms = 1e3 * (time.perf_counter() - t0) / steps  # enqueue only (the GPU is still running)
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / steps
        dev.raise_pending()
        if "--kernels" in sys.argv:
            from sleekit_amd import _lib
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
        print(f"N={N}: {ms:8.3f} ms per step on one rank  ->  {L * R * n / ms / 1e3:8.0f} Mweights/s whole job "
              f"(x{(L * R * n / ms / 1e3) / 1:.0f}), payload {words * 8 / 1e6:.0f} MB per layer; host enqueue {host_ms:.2f} ms per step", flush=True)


if __name__ == "__main__":
    main()
