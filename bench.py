#!/usr/bin/env python3
"""Headline benchmark: Mweights/s quantized (GPTQ 3-bit, 4096 x 4096 layers) on N MI355X.

    python bench.py --gpus N --steps K --warmup W [--config cfg2|cfg3|cfg4|cfg5] [--no-configs]
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...; started WITHOUT a launcher
     -- no WORLD_SIZE in the environment -- `bench.py --gpus N` starts that command itself as a child process, before it touches
     the GPU, relays rank 0's line and exits with the child's code)

One STEP = one pass of the hot path over a batch of synthetic layers, every layer with inputs of its own:
    default   8 layers 4096 x 4096, 3-bit uniform codebook, act_order="diag", damp=0.01, no local search
              (BASELINE.json's headline metric)
    --config  the layer stream of a BASELINE.json config, in MODEL order (SURVEY.md 8 preamble):
              cfg2 OPT-125M  12 x {4 x 768^2, 3072x768, 768x3072}, 3 bit
              cfg3 OPT-350M  24 x {4 x 1024^2, 4096x1024, 1024x4096}, 1.5 bit (3 levels), H - m m^T (inside the step)
              cfg4 BLOOM-560M 24 x {3072x1024, 1024^2, 4096x1024, 1024x4096}, 3 bit + 10 local-search moves
              cfg5 32 x (4096 x 11008), 2 bit
per layer
    [strip mean] -> scale -> damp + order + gather -> float64 factor -> blocked quantize/propagate loop
    [-> local search] -> un-scale -> layer error (W - Qw) H (W - Qw)^T
with W, H and the row scales resident in HBM before the clock starts.  N > 1 shards the
rows of every layer across the ranks (sleekit_amd/dist.py); the factors are made by the ranks in turn and cross
xGMI once, in one RCCL all-gather per round of N same-shaped layers.  Total work is the same at every N ("strong").

The ONE JSON line carries the headline run and, unless --no-configs / --config is given, every other BASELINE config
as a short leg of its own under "configs" (compact: value, ms_per_step, layer_errors {min, max, max_rel_diff_vs_cpu},
roofline {dominant kernel, its frac timed and alone, loop_frac {alone, timed, timed_chip_time}, top four kernels as [name, chip
share, frac]}, cpu_baseline) (cfg2: 20 steps, cfg3, cfg4: 8; cfg5: 2 steps of all 32 layers), each with its own roofline and CPU
baseline, so that one driver run backs every number of DESIGN.md's table.  The line ENDS with "summary": every config's
[Mweights/s, ms per step], the loop's roofline fractions, the single-layer latency, the Hessian accumulation's fraction and the
golden-hash checks in under 600 characters -- what a 2 kB tail of the output still holds.  Before a leg's clock starts its pipeline
runs until the caching allocator has stopped growing (`allocator_settle_steps`, untimed, BEFORE the W warm-up steps): a hipMalloc
inside the timed steps costs tens of ms; a timed region that still saw one is repeated (`timed_region_repeats`).
Fields besides the contract's:
  roofline      for the kernel with the largest share of the chip's time IN THE TIMED CONFIGURATION (HIP events
                 around every launch, on the launch stream: slk_profile_* in the C ABI; a launch of fewer than 256
                 workgroups counts for that share of its duration); achieved = ALGORITHMIC flops or bytes of its
                 launches / their summed duration; `alone` = the same kernel in a single-stream pass (no other
                 kernel shares the chip); `single_stream_leader` = the kernel that leads that pass, when another;
                 `loop` = the whole error-update loop (window + trailing kernels: SURVEY.md 8 a8 + a9) priced as one
                 thing, alone and as timed;
  latency_ms_single_layer   one isolated layer, start to finish;
  asymmetric_H  the same workload with Hessians that are not bit-symmetric (like the experiments' dumps);
  cpu_baseline  the NumPy oracle (bit-identical to the reference, tests/test_oracle_golden.py) timed on this host, rank 0,
                 N = 1 only: headline = one layer, best of 3 after a warm-up; configs = one layer per distinct shape,
                 weighted by the shape's count in the model;
  layer_errors  the layer error of EVERY layer of the last timed step (all finite), and where the CPU ran the same layer,
                 the relative difference to the oracle's error;
  golden        layers of the leg whose (shape, seed, levels, moves) tests/golden/large_cases.json holds a hash for, made by the
                 REAL reference: idx_sha_ok = the indices of the LAST timed step carry that SHA-256 (searched layers: rows whose
                 hash differs from the reference's, each a proven near-tie in tests/test_gpu_parity.py);
  rccl          (N > 1) what the collective did: world size, backend, bytes a rank receives per step, the exchange's
                 duration measured by events on the comm stream, and whether every rank's copy of the payloads carried its
                 root's checksum (sleekit_amd.dist.verify_exchange).
"""

import argparse
import gc
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
# HIP multiplexes a process's streams onto GPU_MAX_HW_QUEUES in-order hardware queues (4 by default): with 3 factor
# streams, a comm stream, loop streams and RCCL's own, some share a queue and stop overlapping -- a loop stream behind a
# factor chain, the next round's factorisation behind this round's loops.  With 8 queues every stream has its own:
# one rank's step at N = 8 takes 4.5 ms instead of 4.9 ... 7.0 (depending on how the streams fell), and on one rank
# three factor streams + ONE loop stream reach 4720 Mweights/s against 4450 for the best pairing found on 4 queues.
# Small layers are the exception: chains of launches of a few microseconds, where 4 queues and more streams measured
# better (768 x 768: 0.34 ms per layer against 0.43).  Set in main(), before HIP initialises.

import torch
import torch.distributed as dist

# MI355X peaks (/opt/skills/guides/MI355X_MICROARCH.md; float64 from the CDNA4 datasheet:
# vector = matrix = 78.6 TFLOP/s, i.e. 32 flop/clk/SIMD at 2.4 GHz on 1024 SIMDs)
PEAK = {"hbm": (8.0e12, "GB/s"), "mfma_f64": (78.6e12, "TFLOP/s"), "mfma_f32": (157.3e12, "TFLOP/s"), "mfma_bf16": (2.5e15, "TFLOP/s")}
KERNEL_DTYPE = {
    "chol_panel": "mfma_f64", "chol_syrk_inner": "mfma_f64", "chol_syrk_outer": "mfma_f64", "trtri_stage0": "mfma_f64", "trtri_stage1": "mfma_f64",
    "chol_syrk_ahead": "mfma_f64", "trtri_level": "mfma_f64", "chol_block": "mfma_f64", "chol_chain": "mfma_f64", "chol_rows_below": "mfma_f64",
    "chol_panel_below": "mfma_f64",
    "gptq_window": "mfma_f64", "gptq_window_wide": "mfma_f64", "gptq_trailing": "mfma_f64",
    "error_gemm": "mfma_f32", "error_gemm_bf16": "mfma_bf16", "hessian_syrk": "mfma_f32",
}
LOOP_KERNELS = ("gptq_window", "gptq_window_wide", "gptq_trailing")

# BASELINE.json configs as layer streams in model order (shapes from results/compare_3b.csv's layer names)
WORKLOADS = {
    "cfg2": dict(name="OPT-125M, all 72 layers", block=[(768, 768)] * 4 + [(3072, 768), (768, 3072)], blocks=12, levels=8, moves=0, strip=False),
    "cfg3": dict(name="OPT-350M, all 144 layers, bias-corrected Hessian", block=[(1024, 1024)] * 4 + [(4096, 1024), (1024, 4096)], blocks=24,
                 levels=3, moves=0, strip=True),
    "cfg4": dict(name="BLOOM-560M, all 96 layers", block=[(3072, 1024), (1024, 1024), (4096, 1024), (1024, 4096)], blocks=24, levels=8,
                 moves=10, strip=False),
    "cfg5": dict(name="Llama-FFN 4096x11008 x 32", block=[(4096, 11008)], blocks=32, levels=4, moves=0, strip=False),
}
# the short legs of the default run: (steps, warm-up, blocks of the model; 0 = all)
CONFIG_LEGS = {"cfg2": (20, 6, 0), "cfg3": (8, 4, 0), "cfg4": (8, 4, 0), "cfg5": (2, 1, 0)}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", choices=sorted(WORKLOADS), default=None, help="a BASELINE.json config's layer stream instead of the headline batch")
    ap.add_argument("--blocks", type=int, default=0, help="with --config: only this many blocks of the model (0 = all)")
    ap.add_argument("--layers", type=int, default=8, help="layers per step (headline batch)")
    ap.add_argument("--rows", type=int, default=4096)
    ap.add_argument("--cols", type=int, default=4096)
    ap.add_argument("--levels", type=int, default=8, help="codebook size (8 = 3 bit)")
    ap.add_argument("--moves", type=int, default=0, help="local-search moves")
    ap.add_argument("--distinct", type=int, default=0, help="distinct synthetic layers cycled through the batch (0 = every layer its own)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the single-layer latency, asymmetric-H and Hessian-accumulation legs")
    ap.add_argument("--no-configs", action="store_true", help="skip the legs of the other BASELINE configs (cfg2 ... cfg5)")
    ap.add_argument("--configs", type=str, default="", help="comma-separated subset of the config legs (default: all four)")
    ap.add_argument("--streams", type=str, default="", help="factor,loop stream counts of the headline leg (default 3,1; 3,3 for small-layer models)")
    ap.add_argument("--stages", action="store_true", help="also print per-kernel timing table to stderr")
    ap.add_argument("--launcher-selftest", action="store_true",
                    help="no GPU: the ranks only form a gloo group, all-reduce their ranks and rank 0 prints a line (tests the self-launch)")
    return ap.parse_args()


class Env:
    """What every leg shares: the process group, the device, the library."""

    def __init__(self, args):
        self.args = args
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))

    def start(self, launch_bound):
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "4" if launch_bound else "8")
        assert torch.cuda.is_available(), "bench.py needs the MI355X"
        dev_index = self.local_rank % torch.cuda.device_count()  # (a rehearsal may put several ranks on one GPU)
        torch.cuda.set_device(dev_index)
        self.device = torch.device("cuda", dev_index)
        self.backend_name = "none"
        if self.world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            self.backend_name = os.environ.get("SLK_DIST_BACKEND", "nccl")  # "gloo" only to rehearse on a 1-GPU box
            if self.backend_name == "nccl":
                dist.init_process_group("nccl", device_id=self.device)
            else:
                dist.init_process_group(self.backend_name)
        assert self.world == self.args.gpus, f"--gpus {self.args.gpus} but WORLD_SIZE={self.world}"

    def fence(self):
        if self.world > 1:
            dist.barrier()
        torch.cuda.synchronize()


class Leg:
    """One workload on this process group: inputs, backend, the timed loop and its measurements."""

    def __init__(self, env, tag, shapes, levels, moves, strip, streams=None, distinct=0, seeds=None):
        from sleekit_amd import codebook, synth
        from sleekit_amd import dist as sdist

        self.env, self.tag, self.shapes, self.levels, self.moves, self.strip = env, tag, shapes, levels, moves, strip
        self.L = len(shapes)
        self.widest = max(n for _, n in shapes)
        self.cb = codebook.UniformCodebook(levels, -1, 1)
        self.weights_per_step = float(sum(R * n for R, n in shapes))
        world = env.world
        # ---- inputs, resident in HBM (identical on every rank: integer-hash generator, run on the GPU -- same bytes as
        #      synth.make_layer on the host, tests/test_gpu_parity.py::test_device_generator_makes_the_same_bytes)
        t0 = time.time()
        distinct = distinct if distinct > 0 else self.L
        self.made, self.layers = {}, []
        # layer i is made from seed 1000 + i unless the caller names the seeds (tests/test_gpu_timed_path.py)
        self.seeds = [1000 + (i % distinct) for i in range(self.L)] if seeds is None else list(seeds)
        assert len(self.seeds) == self.L
        for i, (R, n) in enumerate(shapes):
            key = (R, n, self.seeds[i])
            if key not in self.made:
                lay = synth.make_layer_device(R, n, self.seeds[i], env.device)
                self.made[key] = {k: lay[k] for k in ("W", "H", "scale", "mean")}
                # checked ONCE, here: the synthetic Hessians are bit-wise symmetric (sums of exact products in float64), and
                # the layer says so -- no symmetry check per step, and a searched layer's error is the one the search carries
                self.made[key]["symmetric"] = bool(torch.equal(lay["H"], lay["H"].T))
                if strip:  # ... and so is H - m m^T (m_i m_j commutes): checked on the stripped matrix itself, once, not assumed
                    Hs = self.strip_mean(self.made[key], check=False)["H"]
                    self.made[key]["stripped_symmetric"] = bool(torch.equal(Hs, Hs.T))
                    del Hs
            self.layers.append(self.made[key])
        torch.cuda.synchronize()
        self.t_setup = time.time() - t0
        # flow control (step): steps in flight hold their factors and outputs -- 8 n^2 bytes of U per layer above all: four
        # steps ahead for the 4096-column workloads (2-3 GB each), fewer when a step's factors alone are tens of GB (cfg5: 31)
        step_bytes = sum(8.0 * n * n + 9.0 * R * n for R, n in shapes) / max(world, 1)
        self.depth = max(1, min(4, int(64e9 / max(step_bytes, 1.0))))
        # (on N > 1 ranks the row shards of a round's N layers go through the loop as one batch: one loop stream; the batched
        # rounds run on the factor streams.  From 4 ranks up a rank factors one or two layers per step: TWO factor streams and
        # the batched rounds on a loop stream of their own measured best -- one rank's step of the headline batch,
        # tools/micro_rank_of_n.py with ROUNDS_ON_FS / NF / NL: N = 8: 4.3 ms against 4.7 with the rounds on the factor
        # streams and 5.1 ... 5.4 with three factor streams; N = 4: 7.9 against 8.5 ... 9.8; N = 2 keeps three factor
        # streams and rounds on them, 14.8 against 15.4)
        few_factors = world >= 4
        # a stream of SMALL layers only (OPT-125M: nothing wider than 3072 columns) is chains of narrow launches in batched
        # rounds, which go round all the streams: six of them (40 steps after 8 warm-up steps, ms per step: 3,3 14.5-14.6;
        # 2,4 14.6; 4,4 and 6,6 15.0; 1,1 16.1; no side streams at all 21.1).  With 4096-column layers in the stream 3,1 stays
        # ahead (OPT-350M 63.7 against 64.7 with 3,3 and 73.3 with 4,2)
        launch_bound = self.widest <= 3072 and world == 1
        # (Two factor streams read 0.5 % better for the headline alone -- 25.50 against 25.65 ms per step, alternating runs -- but
        # the legs that follow in the same process then found the side streams in other roles and hardware queues: OPT-350M
        # 84 ms per step instead of 61, the single-layer latency 6.6 ms instead of 5.0.  Every one-rank leg therefore keeps
        # three factor streams in the pool's first three places.)
        self.streams = streams or ((3, 3) if launch_bound else ((2, 1) if few_factors else (3, 1)))
        # (--streams 0,0: no side streams at all, every kernel of the step in order on the caller's stream)
        self.backend = sdist.HipBackend(self.cb, "diag", 0.01, moves, with_error=True, overlap=self.streams if sum(self.streams) else False)
        if few_factors:
            self.backend.rounds_on_factor_streams = False
        self.in_flight = []  # per enqueued step: events at the tail of its streams
        self.host_seconds = 0.0  # spent in quantize_stream (reset by timed())

    # -- the step
    def strip_mean(self, lay, check=True, out=None):
        """a2, remove_input_bias (obq.py:14-25): part of cfg3's path, so inside the step.  The derived layer vouches for the
        symmetry of the STRIPPED Hessian only as far as it was verified on it at setup."""
        from sleekit_amd import _device as dev
        from sleekit_amd import _lib

        n = lay["H"].shape[0]
        out = torch.empty_like(lay["H"]) if out is None else out
        _lib.check(_lib.lib.slk_hessian_strip_mean(dev.ptr(lay["H"]), dev.ptr(lay["mean"]), n, dev.ptr(out), dev.stream_handle()))
        return dict(lay, H=out, symmetric=bool(lay.get("stripped_symmetric", False)) if check else False)

    def step(self, stream_layers=None, be=None):
        from sleekit_amd import dist as sdist

        todo = self.layers if stream_layers is None else stream_layers
        if self.strip:
            # the stripped Hessians of a step live in one of depth + 2 fixed sets of buffers, taken in turn (a step's set is
            # free again when the step depth + 1 before it has been waited for, below): the same bytes every time round, so
            # that the timed steps ask the allocator for nothing -- tensors made per step and dropped while side streams still
            # hold them (record_stream) come back to the pool at times that depend on the run
            ring = self.__dict__.setdefault("_strip_ring", [None] * (self.depth + 2))
            slot = self.__dict__.get("_strip_turn", 0) % len(ring)
            self._strip_turn = slot + 1
            if ring[slot] is None or len(ring[slot]) != len(todo) or any(b.shape != lay["H"].shape for b, lay in zip(ring[slot], todo)):
                ring[slot] = [torch.empty_like(lay["H"]) for lay in todo]
            todo = [self.strip_mean(lay, out=buf) for lay, buf in zip(todo, ring[slot])]
        # join=False: consecutive steps are independent batches, so the next step's factorisations start under
        # this step's loops (the fence waits for everything before the clock stops); the factorisation statuses of every
        # layer are registered and checked after the timed region (raise_pending)
        t_host = time.perf_counter()
        shards = sdist.quantize_stream(todo, be or self.backend, join=False)
        self.host_seconds += time.perf_counter() - t_host  # enqueueing only (nothing in there waits for the GPU)
        # flow control only: the host enqueues a step several times faster than the GPU runs it, and every step in
        # flight holds its own factors and outputs (2-3 GB); never more than self.depth (four) steps ahead
        # (the step's own inputs -- the stripped Hessians of cfg3 are made per step -- stay referenced until the step is
        # through: with join=False the side streams still read them after this returns, and memory handed back to the
        # allocator would be reused by the next step's strip_mean on the current stream)
        fstreams, _, lstreams = (be or self.backend).streams()
        if lstreams:
            evs = []
            for st in lstreams + fstreams:
                e = torch.cuda.Event()
                e.record(st)
                evs.append(e)
            self.in_flight.append((evs, todo))
            if len(self.in_flight) > self.depth:
                for e in self.in_flight.pop(0)[0]:
                    e.synchronize()
        return shards

    def timed(self, n_steps, n_warm, stream_layers=None):
        env = self.env
        # The host enqueues hundreds of launches per step from Python; a full collection of the cyclic garbage collector in the
        # middle of that (every object torch and this process ever made: 35-50 ms) is a stall a launch-bound step cannot hide
        # (one rank of 8 on OPT-125M, 20 steps: 3.5 ms per step, or 5.4-6.3 when a full collection fell into them).  What exists
        # now is long-lived: collect once, then take it out of the collector's sight (INTEGRATION.md says the same to hosts).
        gc.collect()
        gc.freeze()

        def device_mallocs():
            return torch.cuda.memory_stats().get("num_device_alloc", 0)

        # ---- sizing, untimed: the caching allocator keeps a pool per stream and finds new (stream, size) pairs until the
        #      pipeline has run at its full depth a few times; a hipMalloc costs tens of ms.  The pipeline therefore runs, exactly
        #      as it will be timed (no fence inside a batch: the steps in flight are what sizes the pools), until a whole batch
        #      of depth + 1 steps has asked the device for nothing new.  Every rank runs the same number (the slowest decides).
        self.settle_steps = 0
        for _ in range(6):
            before = device_mallocs()
            for _ in range(self.depth + 1):
                self.step(stream_layers)
            self.settle_steps += self.depth + 1
            grew = device_mallocs() - before
            if env.world > 1:
                t = torch.tensor([grew], dtype=torch.int64, device=env.device)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                grew = int(t.item())
            if grew == 0:
                break
        self.repeats = 0
        while True:
            for _ in range(n_warm):
                self.step(stream_layers)
            env.fence()
            self.host_seconds = 0.0
            mallocs = device_mallocs()
            t0 = time.perf_counter()
            for _ in range(n_steps):
                out = self.step(stream_layers)
            self.host_ms_per_step = 1e3 * self.host_seconds / n_steps
            env.fence()
            el = time.perf_counter() - t0
            # (a hipMalloc inside the timed region costs tens of ms: a region that saw one is measured again -- the pools have
            # grown by what it needed -- at most four times; the count of the region that is REPORTED goes into the line)
            grew = device_mallocs() - mallocs
            if env.world > 1:
                t = torch.tensor([el, float(grew)], dtype=torch.float64, device=env.device)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                el, grew = float(t[0].item()), int(t[1].item())
            self.device_mallocs_while_timed = grew
            if grew == 0 or self.repeats >= 4:
                break
            self.repeats += 1
        return el, out

    # -- measurements
    def layer_errors(self, shards):
        """Layer error of EVERY layer of a step (mean over all rows; a scalar all-reduce per layer on N > 1 ranks)."""
        from sleekit_amd import dist as sdist

        vals = torch.stack([sdist.layer_error(sh["row_err"], self.shapes[i][0]) for i, sh in enumerate(shards)])
        return [float(x) for x in vals.cpu().tolist()]

    def profile(self, n_steps, full):
        """The SAME steps once more (same streams, same overlap) with a pair of HIP events around every launch, recorded on
        the launch's own stream; then (one rank) one single-stream step: every kernel alone on the chip."""
        from sleekit_amd import _lib
        from sleekit_amd import dist as sdist

        env, args = self.env, self.env.args
        env.fence()
        _lib.lib.slk_profile_reset()
        _lib.lib.slk_profile_enable(1)
        if env.world > 1:
            self.backend.exchange_log = []
        t0p = time.perf_counter()
        for _ in range(n_steps):
            self.step()
        env.fence()
        t_prof = time.perf_counter() - t0p
        _lib.lib.slk_profile_enable(0)
        table = _lib.profile_report()
        _lib.lib.slk_profile_reset()
        exchange = None
        if env.world > 1:
            log, self.backend.exchange_log = self.backend.exchange_log, None
            if log:
                exchange = dict(ms_per_step=sum(a.elapsed_time(b) for a, b, _ in log) / n_steps, bytes_per_step=sum(x[2] for x in log) / n_steps,
                                collectives_per_step=len(log) / n_steps)
        traffic_db, traffic_src = {}, None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if full and os.path.exists(tpath):
            tj = json.load(open(tpath))
            traffic_db = tj.get("bytes_per_launch", {})
            traffic_src = f"profiles/pmc_traffic.json ({tj.get('made_by', 'tools/profile_round.sh')}, commit {tj.get('commit', 'unrecorded')}): " \
                          "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the headline batch, not measured in this run"
        seq = []
        if env.world == 1 and table:
            _lib.lib.slk_profile_enable(1)
            self.step(be=sdist.HipBackend(self.cb, "diag", 0.01, self.moves, with_error=True, overlap=False))
            torch.cuda.synchronize()
            _lib.lib.slk_profile_enable(0)
            seq = _lib.profile_report()
            _lib.lib.slk_profile_reset()
        if not table:
            return None, exchange, seq
        for k in table + seq:
            k.setdefault("chip_ms", k["total_ms"])

        def describe(k, tab):
            secs = k["total_ms"] * 1e-3
            kind = KERNEL_DTYPE.get(k["kernel"])
            t_flops = k["flops"] / PEAK[kind][0] if kind else 0.0
            t_bytes = k["bytes"] / PEAK["hbm"][0]
            if kind and t_flops >= t_bytes:
                achieved, peak, unit, bound = k["flops"] / secs / 1e12, PEAK[kind][0] / 1e12, "TFLOP/s", "mfma"
                per_launch = k["flops"] / k["launches"]
            else:
                achieved, peak, unit, bound = k["bytes"] / secs / 1e9, PEAK["hbm"][0] / 1e9, "GB/s", "hbm"
                per_launch = k["bytes"] / k["launches"]
            return {
                "kernel": k["kernel"], "bound": bound, "achieved": round(achieved, 3), "peak": peak, "unit": unit,
                "frac": round(achieved / peak, 4), "traffic": traffic_db.get(k["kernel"]),
                "traffic_source": traffic_src if traffic_db.get(k["kernel"]) is not None else None,
                "launches": k["launches"], "avg_launch_us": round(1e3 * k["total_ms"] / k["launches"], 2),
                "algorithmic_per_launch": per_launch, "peak_kind": kind or "hbm",
                "share_of_chip_time": round(k["chip_ms"] / sum(x["chip_ms"] for x in tab), 3),
                "share_of_kernel_time": round(k["total_ms"] / sum(x["total_ms"] for x in tab), 3),
            }

        # "dominant" = most CHIP time in the timed configuration: a launch's duration weighs by the share of the
        # 256 CUs it can occupy (chip_ms, from the library: min(1, workgroups / 256) x duration), so that the
        # one-workgroup latency chains of the factorisation, which run beside the wide kernels of other layers,
        # do not pose as the bottleneck of the step.  (Overlapped durations include queueing for CUs: `alone` below
        # is the kernel's own rate.)
        top = max(table, key=lambda k: k["chip_ms"])
        roofline = describe(top, table)
        if not full:
            for key in ("traffic_source", "peak_kind", "share_of_kernel_time", "algorithmic_per_launch", "launches"):
                roofline.pop(key, None)
        roofline["steps_with_events_ms"] = round(1e3 * t_prof / n_steps, 3)
        # how full the chip was: the launches' chip time (duration x share of the 256 CUs a launch can occupy) over the wall
        # time of the same steps -- above 1 when kernels of several streams share CUs, far below 1 when the step is a chain
        # of narrow or short launches
        roofline["chip_time_over_wall"] = round(sum(k["chip_ms"] for k in table) / (1e3 * t_prof), 3)
        roofline["kernel_time_over_wall"] = round(sum(k["total_ms"] for k in table) / (1e3 * t_prof), 3)

        def frac_of(k):
            kd = KERNEL_DTYPE.get(k["kernel"])
            tf = k["flops"] / PEAK[kd][0] if kd else 0.0
            return max(tf, k["bytes"] / PEAK["hbm"][0]) / max(k["total_ms"] * 1e-3, 1e-12)

        tot_ms = sum(k["total_ms"] for k in table)
        tot_chip = sum(k["chip_ms"] for k in table)
        roofline["top_kernels"] = [
            {"kernel": k["kernel"], "chip_share": round(k["chip_ms"] / tot_chip, 3), "time_share": round(k["total_ms"] / tot_ms, 3),
             "frac": round(frac_of(k), 4), "avg_launch_us": round(1e3 * k["total_ms"] / k["launches"], 2)}
            for k in sorted(table, key=lambda k: -k["chip_ms"])[:8 if full else 5]
        ]
        if seq:
            alone = next((k for k in seq if k["kernel"] == roofline["kernel"]), None)
            if alone:
                d = describe(alone, seq)
                roofline["alone"] = {"avg_launch_us": d["avg_launch_us"], "achieved": d["achieved"], "frac": d["frac"],
                                     "share_of_chip_time": d["share_of_chip_time"]}
                if full:
                    roofline["alone"]["note"] = "single-stream pass after the timed region: no other kernel shares the chip"

            # the whole error-update loop (SURVEY.md 8 a8 + a9: leaf chains and every blocked update = window + trailing
            # kernels; north_star prices the loop as one thing): its algorithmic float64 flops over the summed launch time
            def loop_rate(tab, key="total_ms"):
                ks = [k for k in tab if k["kernel"] in LOOP_KERNELS]
                return sum(k[key] for k in ks), sum(k["flops"] for k in ks)

            ms_a, fl_a = loop_rate(seq)
            ms_t, fl_t = loop_rate(table)
            ms_c, _ = loop_rate(table, "chip_ms")
            if ms_a > 0 and ms_t > 0:
                peak64 = PEAK["mfma_f64"][0]
                roofline["loop"] = {
                    "kernels": "gptq_window + gptq_trailing", "bound": "mfma", "peak": peak64 / 1e12, "unit": "TFLOP/s",
                    "algorithmic_flops_per_layer": fl_a / self.L,
                    "alone": {"ms_per_layer": round(ms_a / self.L, 4), "achieved": round(fl_a / ms_a / 1e9, 3), "frac": round(fl_a / ms_a / 1e9 / (peak64 / 1e12), 4)},
                    "timed": {"ms_per_layer": round(ms_t / n_steps / self.L, 4), "achieved": round(fl_t / ms_t / 1e9, 3),
                              "frac": round(fl_t / ms_t / 1e9 / (peak64 / 1e12), 4)},
                }
                # the same by CHIP time: a window launch of 32-row workgroups occupies R / 32 of the 256 CUs (half of them
                # at 4096 rows) and leaves the rest to the other streams' kernels: it costs its duration times that share
                roofline["loop"]["timed_chip_time"] = {"ms_per_layer": round(ms_c / n_steps / self.L, 4), "achieved": round(fl_t / ms_c / 1e9, 3),
                                                       "frac": round(fl_t / ms_c / 1e9 / (peak64 / 1e12), 4)}
                if full:
                    roofline["loop"]["timed"]["note"] = "launch durations while other layers' kernels share the CUs"
                    roofline["loop"]["timed_chip_time"]["note"] = "launch durations x the share of the 256 CUs the launch can occupy"
            lead = max(seq, key=lambda k: k["chip_ms"])
            if lead["kernel"] != roofline["kernel"]:
                d = describe(lead, seq)
                keys = ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_us", "algorithmic_per_launch", "share_of_chip_time")
                roofline["single_stream_leader"] = {k: d[k] for k in (keys if full else ("kernel", "frac", "avg_launch_us", "share_of_chip_time"))}
            if full:
                tot_seq = sum(k["total_ms"] for k in seq)
                roofline["single_stream_ms_per_layer"] = {k["kernel"]: round(k["total_ms"] / self.L, 4)
                                                          for k in sorted(seq, key=lambda k: -k["total_ms"])[:10]}
                roofline["single_stream_ms_per_layer"]["all kernels"] = round(tot_seq / self.L, 4)
        if args.stages and env.rank == 0 and env.world == 1 and seq:
            tot = sum(k["total_ms"] for k in seq)
            print(f"  [{self.tag}] single-stream pass: {tot:.3f} ms of kernels for {self.L} layers", file=sys.stderr)
            for k in sorted(seq, key=lambda k: -k["total_ms"]):
                print(
                    f"  {k['kernel']:<20s} {k['launches']:5d} launches {k['total_ms']:9.3f} ms {100 * k['total_ms'] / tot:5.1f}%"
                    f"  {k['flops'] / max(k['total_ms'], 1e-9) / 1e9:9.2f} TFLOP/s {k['bytes'] / max(k['total_ms'], 1e-9) / 1e6:9.1f} GB/s",
                    file=sys.stderr,
                )
        return roofline, exchange, seq

    def golden(self, shards):
        """Layers of this leg that tests/golden/large_cases.json holds the REAL reference's index hash for (same generator,
        seed, shape, levels, moves, Hessian treatment): the indices the LAST timed step produced, held to that SHA-256.  A
        searched layer may differ from it in proven near-tie rows (tests/test_gpu_parity.py proves each one from
        the reference's move record): there the number of rows whose hash differs from the reference's per-row hashes
        (tests/golden/ls_traces.npz) is reported instead.  On N > 1 ranks the row shards are gathered on every rank first."""
        import hashlib

        gdir = os.path.join(ROOT, "tests", "golden")
        try:
            cases = json.load(open(os.path.join(gdir, "large_cases.json")))["cases"]
        except OSError:
            return None
        want = {(c["R"], c["n"], c["seed"], c["levels"], c["moves"], bool(c["strip_mean"])): c for c in cases if c["order"] == "diag" and c["damp"] == 0.01}
        out = []
        traces = None
        for i, (R, n) in enumerate(self.shapes):
            c = want.get((R, n, self.seeds[i], self.levels, self.moves, bool(self.strip)))
            if c is None or shards[i].get("idx") is None:
                continue
            idx = shards[i]["idx"]
            if self.env.world > 1:  # equal-sized pieces: pad the shard to the tallest one's height
                tall = -(-R // self.env.world)
                piece = torch.zeros((tall, n), dtype=torch.uint8, device=idx.device)
                piece[:idx.shape[0]] = idx
                parts = [torch.empty_like(piece) for _ in range(self.env.world)]
                dist.all_gather(parts, piece)
                from sleekit_amd import dist as sdist

                idx = torch.cat([parts[r][:sdist.row_range(R, r, self.env.world)[1] - sdist.row_range(R, r, self.env.world)[0]] for r in range(self.env.world)])
            host = idx.cpu().numpy()
            ok = hashlib.sha256(np.ascontiguousarray(host).tobytes()).hexdigest() == c["sha_idx"]
            rec = {"layer": i, "shape": f"{R}x{n}", "seed": self.seeds[i], "idx_sha_ok": ok}
            if not ok and self.moves > 0:
                try:
                    traces = traces if traces is not None else np.load(os.path.join(gdir, "ls_traces.npz"))
                    name = f"r{R}_n{n}_s{self.seeds[i]}_N{self.levels}_diag_ls{self.moves}"
                    rows = np.array([int.from_bytes(hashlib.sha256(np.ascontiguousarray(r).tobytes()).digest()[:8], "little") for r in host], dtype=np.uint64)
                    bad = np.flatnonzero(rows != traces[name + "/row_hash"])
                    rec["rows_differing"] = int(len(bad))
                    rec["all_recorded_near_ties"] = bool(np.isin(bad, traces[name + "/rows"]).all())
                except (OSError, KeyError):
                    pass
            out.append(rec)
        return out or None

    def cpu_baseline(self, shards, sample_layers, best_of):
        """The oracle on the host cores for `sample_layers` = [(layer index, count in the model)], each timed `best_of` times
        after ONE warm-up of LAPACK / BLAS; rate = weights of the sample (x counts) / its time (x counts).  Where the oracle
        ran a layer, its error is held against the GPU's for the same rows."""
        from oracle import grid, obq_ref, scaling_ref
        from sleekit_amd import synth

        g = grid.UniformGrid(self.levels, -1, 1)
        small = synth.make_layer(512, 512, 999)
        scaling_ref.quantize_scaled(small["W"], small["scale"], g, small["H"])  # LAPACK/BLAS warm-up
        tot_w = tot_t = 0.0
        parts, checks = [], []
        for idx, count in sample_layers:
            R, n = self.shapes[idx]
            rows = min(R, 512) if n > 8192 else R  # (the n^3 factorisation of an 11008-column layer alone is most of a minute)
            lay = self.layers[idx]
            host = {k: (lay[k][:rows] if k in ("W", "scale") else lay[k]).cpu().numpy() for k in ("W", "H", "scale", "mean")}
            best, e_cpu = None, None
            for _ in range(best_of):
                t1 = time.perf_counter()
                H0 = obq_ref.strip_input_mean(host["H"], host["mean"]) if self.strip else host["H"]
                out = scaling_ref.quantize_scaled(host["W"], host["scale"], g, H0, "diag", 0.01, self.moves)
                e_cpu = float(obq_ref.mean_error(host["W"], out, H0))
                t = time.perf_counter() - t1
                best = t if best is None else min(best, t)
            tot_w += count * rows * n
            tot_t += count * best
            parts.append(f"{rows}x{n}" + (f" (first {rows} rows of {R})" if rows != R else "") + f" x{count}: {best:.2f} s")
            e_gpu = float(shards[idx]["row_err"][:rows].double().mean().item())
            checks.append({"layer": idx, "shape": [rows, n], "cpu": e_cpu, "gpu": e_gpu, "rel_diff": abs(e_gpu - e_cpu) / abs(e_cpu)})
        cpu = {
            "value": round(tot_w / tot_t / 1e6, 3), "unit": "Mweights/s", "cores": os.cpu_count(), "kind": "port",
            "sample": f"one layer per distinct shape, weighted by its count in the model: {'; '.join(parts)}; best of {best_of} after a 512x512 warm-up; "
                      f"quantize_with_scaling + quantization_error, {self.levels} levels, moves={self.moves}{', H - m m^T' if self.strip else ''}, "
                      f"NumPy {np.__version__} OpenBLAS threads=all",
        }
        return cpu, checks

    def release(self):
        from sleekit_amd import _device as dev

        torch.cuda.synchronize()
        self.layers, self.made, self.backend, self.in_flight = [], {}, None, []
        self.__dict__.pop("_strip_ring", None)
        dev.release_workspaces()
        torch.cuda.empty_cache()


def distinct_shape_sample(shapes):
    """[(index of the first layer of each distinct shape, how many layers have it)] in order of appearance."""
    first, count = {}, {}
    for i, sh in enumerate(shapes):
        first.setdefault(sh, i)
        count[sh] = count.get(sh, 0) + 1
    return [(first[sh], count[sh]) for sh in first]


def config_leg(env, name):
    """One BASELINE config as a short leg: value, roofline (compact), CPU baseline, layer-error check."""
    from sleekit_amd import _device as dev

    args = env.args
    wl = WORKLOADS[name]
    steps, warm, blocks = CONFIG_LEGS[name]
    shapes = wl["block"] * (blocks or wl["blocks"])
    leg = Leg(env, name, shapes, wl["levels"], wl["moves"], wl["strip"])
    try:
        elapsed, shards = leg.timed(steps, warm)
        dev.raise_pending()
        errs = leg.layer_errors(shards)
        out = {
            "workload": f"{wl['name']} in model order" + (f", first {len(shapes)} of {len(wl['block']) * wl['blocks']} layers" if blocks else "")
                        + f", {wl['levels']} levels, moves={wl['moves']}{', H - m m^T' if wl['strip'] else ''}",
            "value": round(leg.weights_per_step / (elapsed / steps) / 1e6, 2), "unit": "Mweights/s", "ms_per_step": round(1e3 * elapsed / steps, 3),
            "steps": steps, "warmup": warm, "layers": len(shapes), "setup_seconds": round(leg.t_setup, 1),
            "host_enqueue_ms_per_step": round(leg.host_ms_per_step, 3),
            "device_mallocs_while_timed": leg.device_mallocs_while_timed, "allocator_settle_steps": leg.settle_steps,
            "timed_region_repeats": leg.repeats,
            "layer_errors": {"layers": len(errs), "all_finite": bool(np.all(np.isfinite(errs))), "min": min(errs), "max": max(errs)},
            "golden": leg.golden(shards),
        }
        if not args.no_profile:
            roofline, exchange, _ = leg.profile(min(steps, 2), full=False)
            dev.raise_pending()
            if roofline:
                # compact (the one JSON line carries five workloads): the dominant kernel, the loop's three rates, the top four
                r = roofline
                loop = r.get("loop") or {}
                out["roofline"] = {
                    "kernel": r["kernel"], "bound": r["bound"], "achieved": r["achieved"], "peak": r["peak"], "unit": r["unit"], "frac": r["frac"],
                    "chip_share": r["share_of_chip_time"], "avg_launch_us": r["avg_launch_us"], "alone_frac": (r.get("alone") or {}).get("frac"),
                    "chip_time_over_wall": r.get("chip_time_over_wall"),
                    "loop_frac": {k: (loop.get(k) or {}).get("frac") for k in ("alone", "timed", "timed_chip_time")} if loop else None,
                    "top": [[k["kernel"], k["chip_share"], k["frac"]] for k in r["top_kernels"][:4]],
                }
            if exchange:
                out["exchange"] = {k: round(v, 3) for k, v in exchange.items()}
        if env.rank == 0 and env.world == 1 and not args.no_cpu_baseline:
            cpu, checks = leg.cpu_baseline(shards, distinct_shape_sample(shapes), 1)
            cpu["sample"] = cpu["sample"].split("; best of")[0].replace("one layer per distinct shape, weighted by its count in the model: ", "per shape x count: ")
            out["cpu_baseline"] = cpu
            out["layer_errors"]["checked_against_cpu"] = len(checks)
            out["layer_errors"]["max_rel_diff_vs_cpu"] = max(c["rel_diff"] for c in checks)
        return out
    finally:
        leg.release()


def free_port():
    import socket

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


def launch_ranks(args):
    """`bench.py --gpus N` without a launcher around it: start `python -m torch.distributed.run --nproc-per-node N bench.py
    ...` as a CHILD process (this parent has not touched the GPU and never will; never exec), hand rank 0's JSON line on and
    return the child's exit code."""
    import subprocess

    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True, bufsize=1)
    last = None
    for text in child.stdout:  # rank 0 prints ONE JSON line; anything else the ranks say goes to stderr untouched
        text = text.rstrip("\n")
        if text.startswith("{") and text.endswith("}"):
            last = text
        elif text:
            print(text, file=sys.stderr, flush=True)
    rc = child.wait()
    if last is not None:
        print(last, flush=True)
    return rc if rc != 0 or last is not None else 4  # (no line at all is a failure too)


def launcher_selftest(args):
    """The ranks of a (self-)launched job without any GPU work: a gloo group, one all-reduce, one line from rank 0."""
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
        t = torch.tensor([rank + 1], dtype=torch.int64)
        dist.all_reduce(t)
        total = int(t.item())
        dist.destroy_process_group()
    else:
        total = 1
    if rank == 0:
        print(json.dumps({"launcher_selftest": True, "world_size": world, "gpus_asked": args.gpus, "sum_of_ranks_plus_one": total}), flush=True)
    return 0 if world == args.gpus else 5


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    if args.launcher_selftest:
        sys.exit(launcher_selftest(args))
    env = Env(args)
    if args.config:
        wl = dict(WORKLOADS[args.config])
        if args.blocks:
            wl["blocks"] = args.blocks
        shapes = wl["block"] * wl["blocks"]
        levels, moves, strip = wl["levels"], wl["moves"], wl["strip"]
    else:
        shapes = [(args.rows, args.cols)] * args.layers
        levels, moves, strip = args.levels, args.moves, False
    widest = max(n for _, n in shapes)
    run_configs = not args.config and not args.no_configs
    # small shapes alone (see the note on hardware queues at the top); with the config legs in the same process: 8 queues
    env.start(launch_bound=widest <= 1024 and env.world == 1 and not run_configs)
    rank, world, device = env.rank, env.world, env.device

    from sleekit_amd import _device as dev
    from sleekit_amd import _lib
    from sleekit_amd import dist as sdist

    dev.lazy_errors = True
    # (measurement knob: which hardware queue a stream lands in follows from how many streams were made before it)
    _spare = [torch.cuda.Stream() for _ in range(int(os.environ.get("SLK_BENCH_STREAM_OFFSET", "0")))]
    streams = tuple(int(x) for x in args.streams.split(",")) if args.streams else None
    head = Leg(env, args.config or "headline", shapes, levels, moves, strip, streams=streams, distinct=args.distinct)
    L = head.L

    elapsed, shards = head.timed(args.steps, args.warmup)
    host_ms = head.host_ms_per_step
    head_mallocs, head_settle, head_repeats = head.device_mallocs_while_timed, head.settle_steps, head.repeats
    dev.raise_pending()  # a Hessian that is not positive definite in any layer of any step: LinAlgError naming the layer
    peak_hbm = torch.cuda.max_memory_reserved(device)  # after the timed region (at most five steps in flight)
    ms_per_step = 1e3 * elapsed / args.steps
    value = head.weights_per_step / (elapsed / args.steps) / 1e6

    # layer error of EVERY layer of the last step (sanity: finite, GPTQ-sized) -- bookkeeping, not timed
    errs = head.layer_errors(shards)
    assert all(np.isfinite(errs)), errs
    layer_errors = {"layers": len(errs), "all_finite": True, "min": min(errs), "max": max(errs)}
    if len(errs) <= 16:
        layer_errors["values"] = errs
    golden = head.golden(shards)

    roofline, exchange = None, None
    if not args.no_profile:
        roofline, exchange, _ = head.profile(args.steps, full=True)
        dev.raise_pending()

    # ---- N > 1: what the collective did, and that every rank received its roots' bytes
    rccl = None
    if world > 1:
        check = sdist.verify_exchange(head.layers if not strip else [head.strip_mean(lay) for lay in head.layers], head.backend)
        env.fence()
        rccl = {"world_size": check["world_size"], "backend": check["backend"], "payload_bytes": check["payload_bytes"],
                "payload_checksums_agree": check["agree"], "rounds_checked": check["rounds_checked"]}
        if exchange:
            rccl.update(all_gather_bytes_per_step=int(exchange["bytes_per_step"]), exchange_ms=round(exchange["ms_per_step"], 3),
                        collectives_per_step=exchange["collectives_per_step"],
                        note="bytes a rank RECEIVES per step; exchange_ms = summed duration of the step's all-gathers, events on the comm "
                             "stream, in the pass repeated with events (it overlaps the loops of the round before)")
        assert check["agree"], "a rank's copy of a packed factor differs from its root's"

    extras = rank == 0 and world == 1 and not args.no_extras
    # ---- one isolated layer, start to finish ("report the single-layer number separately")
    latency = None
    if extras:
        # through the single-layer API (sleekit_amd.engine.quantize_layer + row_errors: what scaling.quantize_with_scaling
        # and obq.quantization_error run on device tensors), where the factorisation looks ahead on a helper stream
        from sleekit_amd import engine

        lat = []
        for i in range(6):
            lay = head.strip_mean(head.layers[i % L]) if strip else head.layers[i % L]
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            res = engine.quantize_layer(lay["W"], lay["H"], head.cb, lay["scale"], "diag", 0.01, moves)
            engine.row_errors(lay["W"], res.Q, lay["H"])
            torch.cuda.synchronize()
            lat.append(1e3 * (time.perf_counter() - t1))
        dev.raise_pending()
        latency = {"ms": round(float(np.median(lat[1:])), 3), "shape": list(shapes[0]) if len(set(shapes)) == 1 else [list(shapes[i % L]) for i in range(1, 6)],
                   "note": "median of 5 after one warm-up: a single layer alone on the GPU through engine.quantize_layer + row_errors, host call to synchronize"}

    # ---- the same workload on Hessians that are NOT bit-symmetric (made by a library GEMM, like the experiments' dumps
    #      of torch `inp @ inp.t()`): the layer error keeps its half-product route by averaging H with its transpose in the split
    asym = None
    if extras and not args.config:
        n = args.cols
        alt = {}
        gemm_symmetric = True
        for i, lay in enumerate(head.layers):
            if id(lay) not in alt:
                X = torch.randn(2 * n, n, device=device) * (0.5 + 2.0 * torch.rand(n, device=device))
                X[:, :8] *= 8.0
                Ha = (X.T @ X) / float(2 * n)
                # (hipBLASLt's X.T @ X came out bit-symmetric here -- every C[i][j] and C[j][i] sums the same products in the
                # same order -- so the asymmetry a dump from another BLAS may carry is put in by hand: noise of 1e-6 of the
                # mean magnitude above the diagonal)
                gemm_symmetric = gemm_symmetric and bool(torch.equal(Ha, Ha.T))
                Ha = Ha + torch.triu(torch.randn_like(Ha), 1) * (1e-6 * float(Ha.abs().mean()))
                alt[id(lay)] = dict(lay, H=Ha, symmetric=False)
        alt_layers = [alt[id(lay)] for lay in head.layers]
        symmetric = all(bool(torch.equal(a["H"], a["H"].T)) for a in alt.values())
        el, _ = head.timed(args.steps, 1, alt_layers)
        dev.raise_pending()
        asym = {"value": round(head.weights_per_step / (el / args.steps) / 1e6, 2), "unit": "Mweights/s", "ms_per_layer": round(1e3 * el / args.steps / L, 3),
                "H": "torch X.T @ X / T in float32 + 1e-6 relative noise above the diagonal", "bitwise_symmetric": symmetric,
                "library_gemm_result_was_symmetric": gemm_symmetric}
        del alt, alt_layers

    # ---- a13, the local search alone at 10 and 100 moves (sleekit_heavy runs 100: sleekit/statistics.py:143) on one layer of
    #      the headline's shape: the move kernel's own time (events around its launch) against its algorithmic bytes
    search = None
    if extras and not args.config:
        from sleekit_amd import engine

        lay = head.layers[0]
        Ws = engine.rows_divide(lay["W"], lay["scale"])
        base = engine.quantize_layer(lay["W"], lay["H"], head.cb, lay["scale"], "diag", 0.01, 0, unscale=False)
        search = {"shape": list(shapes[0]), "unit": "us", "bound": "hbm", "peak_gb_s": PEAK["hbm"][0] / 1e9}
        for mv in (10, 100):
            for rep in range(2):  # (the second is timed)
                Qs = base.Q.clone()
                torch.cuda.synchronize()
                _lib.lib.slk_profile_reset()
                _lib.lib.slk_profile_enable(1)
                engine.local_search(Ws, Qs, lay["H"], engine.require_uniform(head.cb), mv)
                torch.cuda.synchronize()
                _lib.lib.slk_profile_enable(0)
                rep_ = {k["kernel"]: k for k in _lib.profile_report()}
                _lib.lib.slk_profile_reset()
            k = rep_.get("local_search")
            if k:
                # the moves the rows actually took (a row stops when nothing is left to gain): an untimed run with the record
                Qs = base.Q.clone()
                rec = engine.local_search(Ws, Qs, lay["H"], engine.require_uniform(head.cb), mv, want_trace=True)
                taken = int((rec >= 0).sum().item())
                R_, n_ = shapes[0]
                moved_bytes = 4.0 * n_ * taken + 13.0 * R_ * n_
                search[f"moves_{mv}"] = {"us": round(1e3 * k["total_ms"], 1), "moves_taken_per_row": round(taken / R_, 2),
                                         "us_per_move_taken": round(1e3 * k["total_ms"] / max(taken / R_, 1e-9), 2),
                                         "achieved_gb_s": round(moved_bytes / (k["total_ms"] * 1e-3) / 1e9, 1),
                                         "frac": round(moved_bytes / (k["total_ms"] * 1e-3) / PEAK["hbm"][0], 4)}
                del rec
        search["algorithmic_bytes"] = "moves TAKEN x 4 n (one streamed row of H per row and move taken) + 13 R n (state in and out), SURVEY.md 8d"
        del Ws, base

    # ---- a1, Hessian accumulation, timed as its own stage: 2048-token batches into an n x n Hessian
    hess = None
    if extras:
        n = widest
        Hacc = torch.zeros((n, n), dtype=torch.float32, device=device)
        macc = torch.zeros(n, dtype=torch.float32, device=device)
        X = torch.randn(2048, n, device=device)
        s_ = dev.stream_handle()
        hws, hws_bytes = dev.workspace(0, n)
        for it in range(2):
            _lib.check(_lib.lib.slk_hessian_accumulate(Hacc.data_ptr(), macc.data_ptr(), X.data_ptr(), n, 2048, 2048 * it, dev.ptr(hws), hws_bytes, s_))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for it in range(4):
            _lib.check(_lib.lib.slk_hessian_accumulate(Hacc.data_ptr(), macc.data_ptr(), X.data_ptr(), n, 2048, 4096 + 2048 * it, dev.ptr(hws), hws_bytes, s_))
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 4
        flops = 2048.0 * n * (n + 1)
        hess = {"features": n, "tokens_per_batch": 2048, "ms_per_batch": round(ms, 3), "achieved_tflops": round(flops / ms / 1e9, 2),
                "peak_tflops": PEAK["mfma_f32"][0] / 1e12, "frac": round(flops / ms / 1e9 / (PEAK["mfma_f32"][0] / 1e12), 4),
                "algorithmic_flops": "T n (n + 1): one triangle, float32 by definition; executed as six bfloat16 products each when n % 128 == 0"}
        del Hacc, macc, X

    # ---- CPU baseline: the oracle on the host cores, one layer of the same workload (best of 3), or one per shape
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sample = distinct_shape_sample(shapes) if args.config else [(0, 1)]
        cpu, checks = head.cpu_baseline(shards, sample, 1 if args.config else 3)
        layer_errors["against_cpu"] = checks
        cpu["layer_error"], cpu["gpu_layer_error"] = checks[0]["cpu"], checks[0]["gpu"]

    if args.config:
        wl = WORKLOADS[args.config]
        metric = f"Mweights/sec quantized ({wl['name']}, {np.log2(levels):.3g}-bit)"
        workload = (f"{args.config}: {wl['name']} in model order, {L} layers, {levels}-level uniform codebook, act_order=diag, damp=0.01, "
                    f"local-search moves={moves}, {'H - m m^T, ' if strip else ''}layer error included")
    else:
        metric = "Mweights/sec quantized (GPTQ 3-bit, 4096x4096 layer)"
        workload = (f"{L} layers {args.rows}x{args.cols}, {levels}-level uniform codebook (GPTQ {np.log2(levels):g}-bit), "
                    f"act_order=diag, damp=0.01, local-search moves={moves}, layer error included")
    line = {
        "metric": metric, "value": round(value, 2), "unit": "Mweights/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": workload, "layers_per_step": L, "distinct_inputs": len(head.made),
            "shapes": sorted({f"{R}x{n}" for R, n in shapes}), "row_sharding": f"{world} ranks",
            "ms_per_layer": round(ms_per_step / L, 3), "setup_seconds": round(head.t_setup, 1),
            "host_enqueue_ms_per_step": round(host_ms, 3),
            "device_mallocs_while_timed": head_mallocs, "allocator_settle_steps": head_settle, "timed_region_repeats": head_repeats,
            "peak_hbm_gb": round(peak_hbm / 2**30, 2),
            "streams": {"factor": head.streams[0], "loop": head.streams[1]},
            "gpu_max_hw_queues": int(os.environ.get("GPU_MAX_HW_QUEUES", "4")),
        },
        "roofline": roofline, "cpu_baseline": cpu, "layer_error": errs[0], "layer_errors": layer_errors, "latency_ms_single_layer": latency,
        "asymmetric_H": asym, "hessian_accumulate": hess, "local_search": search, "golden": golden,
    }
    if rccl:
        line["rccl"] = rccl

    # ---- the other BASELINE configs, a short leg each.  The headline above is complete: whatever happens in a leg (an
    #      exception, or on N > 1 ranks a collective that never returns) the line is printed -- a watchdog prints it and
    #      ends the process if the legs take longer than their allowance.
    printed = threading.Event()

    def summary():
        """What must survive a 2 kB tail of the output: the LAST key of the line, under 600 characters."""
        loop = (roofline or {}).get("loop") or {}
        s = {"headline": [round(value, 1), round(ms_per_step, 2)]}
        for name, leg in (line.get("configs") or {}).items():
            if isinstance(leg, dict):
                s[name] = [leg["value"], leg["ms_per_step"]] if "value" in leg else "error"
        s["loop"] = {k: (loop.get(k) or {}).get("frac") for k in ("alone", "timed")} if loop else None
        s["latency_ms"] = latency["ms"] if latency else None
        s["hessian_frac"] = hess["frac"] if hess else None
        s["ls_us"] = [(search.get(f"moves_{m}") or {}).get("us") for m in (10, 100)] if search else None
        gold = list(golden or []) + [g for leg in (line.get("configs") or {}).values() if isinstance(leg, dict) for g in (leg.get("golden") or [])]
        s["golden_ok"] = [sum(1 for g in gold if g["idx_sha_ok"] or g.get("all_recorded_near_ties")), len(gold)]
        s["mallocs_timed"] = [head_mallocs] + [leg.get("device_mallocs_while_timed") for leg in (line.get("configs") or {}).values() if isinstance(leg, dict)]
        if "error" in (line.get("configs") or {}):
            s["error"] = line["configs"]["error"][:80]
        return s

    def emit():
        if not printed.is_set():
            printed.set()
            if rank == 0:
                line.pop("summary", None)
                line["summary"] = summary()  # last key: the end of the line
                print(json.dumps(line), flush=True)

    if run_configs:
        head.release()
        names = [c for c in (args.configs.split(",") if args.configs else sorted(CONFIG_LEGS)) if c in CONFIG_LEGS]
        line["configs"] = {}
        allowance = float(os.environ.get("SLK_BENCH_CONFIGS_SECONDS", "420"))

        def bail():
            # a leg that never came back (on N > 1 ranks: a collective): the headline is printed, and the process ends
            # NON-ZERO -- the run did not complete
            line["configs"]["error"] = f"config legs exceeded {allowance:.0f} s; stopped"
            emit()
            sys.stdout.flush()
            os._exit(3)

        dog = threading.Timer(allowance, bail)
        dog.daemon = True
        dog.start()
        t_legs = time.time()
        for name in names:
            try:
                line["configs"][name] = config_leg(env, name)
            except Exception as exc:  # the headline stands; the leg says what went wrong
                line["configs"][name] = {"error": f"{type(exc).__name__}: {exc}"[:400]}
                if world > 1:
                    break  # the ranks are out of step: no further collectives
        line["configs"]["seconds"] = round(time.time() - t_legs, 1)
        dog.cancel()
    emit()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
