"""Row-sharded quantization of a stream of layers over the GPUs of one node.

Given the shared factor U, every output row of W goes through the loop, the local search
and the error on its own (sleekit/obq.py:106-137, 264-346, 89-95 have no cross-row term),
so the path shards by rows with no data-path collective.  What does NOT shard is the
n x n factorisation; over a stream of layers it is spread instead: rank (l mod G) factors
layer l and its packed factor (status, order, upper triangle of U) crosses xGMI exactly once,
while every rank runs rows [r R/G, (r+1) R/G) of every layer.

One process per GPU; `torch.distributed` must be initialised by the caller (backend "nccl"
is RCCL on ROCm).  The exchange is one asynchronous all-gather per round of G layers (G
simultaneous broadcasts, one root each), issued up front so the loops of round g overlap the
factorisations and the transfer of round g+1; on a single rank nothing is communicated.

The module is engine-agnostic: `backend` supplies factorize / run_rows, which lets the CPU
test-suite drive the same scheduling code over gloo with a stand-in backend.
"""

import os

import torch
import torch.distributed as dist


# Test hook: exchange (pack, all-gather, unpack) even on a single rank, so that a one-GPU box can drive the
# RCCL path end to end (tests/test_gpu_parity.py::test_exchange_over_rccl_single_rank).
always_exchange = False


# Measurement hook (tools/micro_rank_of_n.py): (rank, size) makes this process behave as ONE rank of a larger job on a
# single GPU -- same rounds, roots, streams and kernels -- with the all-gather replaced by a local hand-over of this
# rank's own payload in every slot (the results are meaningless, the work is that of the real rank).
rehearse = None
_rehearsal_payloads = {}
_stream_pool = {}  # device index -> the side streams every HipBackend of this process hands out (HipBackend.streams)


class _Done:
    """Stands in for the work handle of a collective: wait() orders the current stream behind `event`."""

    def __init__(self, event=None):
        self.event = event

    def wait(self):
        if self.event is not None:
            torch.cuda.current_stream().wait_event(self.event)
        return True


def world():
    if rehearse is not None:
        return rehearse
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def row_range(R, rank, size):
    """Rows [lo, hi) of rank `rank`: contiguous, sizes differ by at most one, covers [0, R) exactly."""
    base, extra = divmod(R, size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def factor_root(position, size):
    """Root of the layer at `position` of the processing order (plan_rounds): positions go round the ranks."""
    return position % size


def plan_rounds(layers, size, bucket=True):
    """Rounds of at most `size` layers, each of ONE shape, and every layer's factor root.

    A model's layers come in an order that mixes shapes (an OPT / BLOOM block is {d x d ..., 4d x d, d x 4d};
    reference: experiments/compare.py:37-53 walks one directory per layer), but they are independent
    (compare.py:50-131), so the stream is bucketed by (rows, columns, scaled?) and each bucket cut into rounds: the
    row shards of a round's layers then go through the kernels as one batch (HipBackend.run_round).  Roots follow
    the position in this processing order, so a bucket's partial last round does not leave the same ranks idle
    every time.  Returns (rounds, root): rounds = lists of layer indices, root[l] = the rank that factors layer l.
    """
    n_layers = len(layers)
    if size <= 1 or not bucket:
        groups = [list(range(n_layers))]
    else:
        by_shape = {}
        for l, lay in enumerate(layers):
            key = (tuple(lay["W"].shape), tuple(lay["H"].shape), lay.get("scale") is not None)
            by_shape.setdefault(key, []).append(l)
        groups = list(by_shape.values())  # in order of first appearance
    rounds, root, position = [], [None] * n_layers, 0
    for members in groups:
        for i in range(0, len(members), max(size, 1)):
            chunk = members[i:i + max(size, 1)]
            for l in chunk:
                root[l] = factor_root(position, max(size, 1))
                position += 1
            rounds.append(chunk)
    return rounds, root


class HipBackend:
    """The real thing: sleekit_amd.engine on the current device."""

    def __init__(self, quantizer, act_order="diag", damp=0.01, nb_ls_moves=0, with_error=True, overlap=True):
        from . import engine

        self.engine, self.quantizer, self.overlap = engine, quantizer, overlap
        self.act_order, self.damp, self.moves, self.with_error = act_order, damp, nb_ls_moves, with_error
        self.local_batch = int(os.environ.get("SLK_LOCAL_BATCH", self.local_batch))            # (measurement knobs)
        self.local_batch_cols = int(os.environ.get("SLK_LOCAL_BATCH_COLS", self.local_batch_cols))
        self.group_rows = int(os.environ.get("SLK_GROUP_ROWS", self.group_rows))
        self.short_factor_batch = int(os.environ.get("SLK_SHORT_FACTOR_BATCH", self.short_factor_batch))
        self.short_rows = int(os.environ.get("SLK_SHORT_ROWS", self.short_rows))
        self.group_wide_rows = int(os.environ.get("SLK_GROUP_WIDE_ROWS", self.group_wide_rows))
        self.group_bytes = int(os.environ.get("SLK_GROUP_BYTES", self.group_bytes))

    def streams(self):
        """(factor streams, comm stream, loop streams), created once; (None, None, None) = everything in order."""
        if not self.overlap:
            return None, None, None
        if not hasattr(self, "_streams"):
            nf, nl = (self.overlap if isinstance(self.overlap, tuple) else (3, 1))  # best on 8 hardware queues
            # ONE set of side streams per device and process, shared by every backend: with 8 hardware queues
            # (GPU_MAX_HW_QUEUES, sleekit_amd/__init__.py) the first seven streams a process makes -- and the default stream --
            # have a queue each, later ones double up with them and stop overlapping.  A second backend with streams of its own
            # ran like a different program (bench.py: OPT-350M 69.8 ms per step after the headline's backend, 61.2 on its streams).
            dev_index = torch.cuda.current_device()
            pool = _stream_pool.setdefault(dev_index, [])
            while len(pool) < nf + nl + 1:
                pool.append(torch.cuda.Stream())
            self._streams = (pool[:nf], pool[nf], pool[nf + 1:nf + 1 + nl])
        return self._streams

    def payload_words(self, n):
        """slk_factor_pack's words, plus one: whether H is bit-wise symmetric (the root checks it once for all ranks)."""
        from . import _lib

        return int(_lib.lib.slk_factor_payload_words(n)) + 1

    def alloc_payload(self, words, device):
        return torch.empty(words, dtype=torch.int64, device=device)

    def pack(self, factor, words):
        """(order, U, info) -> the first payload_words(n) of a `words`-long int64 buffer:
        status, order, packed upper triangle of U."""
        from . import _device as dev
        from . import _lib

        order, U, info = factor[:3]
        n = U.shape[0]
        payload = self.alloc_payload(words, U.device)
        _lib.check(_lib.lib.slk_factor_pack(dev.ptr(U), dev.ptr(order), dev.ptr(info), n, dev.ptr(payload), dev.stream_handle()))
        mark = self.payload_words(n) - 1
        if len(factor) > 3:
            payload[mark:mark + 1].copy_(factor[3])  # int32 -> int64 on the way
        else:
            payload[mark:mark + 1].fill_(-1)  # not checked by the root
        return payload

    def unpack(self, payload, n):
        from . import _device as dev
        from . import _lib

        order = torch.empty(n, dtype=torch.int64, device=payload.device)
        U = torch.empty((n, n), dtype=torch.float64, device=payload.device)
        info = torch.empty(1, dtype=torch.int32, device=payload.device)
        _lib.check(_lib.lib.slk_factor_unpack(dev.ptr(payload), n, dev.ptr(U), dev.ptr(order), dev.ptr(info), dev.stream_handle()))
        mark = self.payload_words(n) - 1
        return order, U, info, payload[mark:mark + 1].to(torch.int32)  # + the root's symmetry verdict (-1: none)

    def factorize(self, layer):
        eng = self.engine
        W, H, n = layer["W"], layer["H"], layer["H"].shape[0]
        mode = eng.order_mode_code(self.act_order)
        miss = None
        if mode == 4:  # inv_diag / combined_diag / pivot: sort keys from a kernel of their own
            miss = eng.order_keys(H, n, self.damp, self.act_order)
        elif mode >= 2:  # err / sqerr need the statistics of ALL rows, before sharding
            cb = eng.require_uniform(self.quantizer)
            Ws = eng.rows_divide(W, layer["scale"]) if layer.get("scale") is not None else W
            miss = eng.column_miss(Ws, cb, mode == 3)
        factor = eng.factorize(H, n, self.damp, mode, miss)
        if self.with_error:  # the layer error wants to know whether H is symmetric: decided here, once per layer
            factor = factor + (self._symmetry(layer),)
        return factor

    def _symmetry(self, layer):
        """int32[1] on the device: 1 iff H is bit-wise symmetric.  A layer dict may VOUCH for it (`symmetric=True`: the caller
        has checked, e.g. once when the statistics were loaded): no check per call then, and a searched layer's error can be
        the one the search carries (see _run_stacked)."""
        if layer.get("symmetric") is True:
            dev_ = layer["H"].device
            if getattr(self, "_yes", None) is None or self._yes.device != dev_:
                self._yes = torch.ones(1, dtype=torch.int32, device=dev_)
            return self._yes
        return self.engine.symmetry_flag(layer["H"])

    def factorize_many(self, layers):
        """The factors of several layers of ONE width in launches that cover them all (engine.factorize_batch: bit-equal to
        factorize, one by one): what a rank owes to a group of rounds of small layers."""
        eng = self.engine
        n = layers[0]["H"].shape[0]
        mode = eng.order_mode_code(self.act_order)
        if mode not in (0, 1) or any(lay["H"].shape[0] != n for lay in layers) or len(layers) > 64:
            return [self.factorize(lay) for lay in layers]
        order, U, info = eng.factorize_batch([lay["H"] for lay in layers], n, self.damp, mode)
        out = []
        for b, lay in enumerate(layers):
            fac = (order[b], U[b], info[b:b + 1])
            if self.with_error:
                fac = fac + (self._symmetry(lay),)
            out.append(fac)
        return out

    group_rows = 8192  # stacked (padded) rows a loop batch of small shards may reach
    group_bytes = 1 << 32  # ... and the bytes of their stacked factors (24 factors of 4096 columns)
    # WIDE layers join groups too when a rank's shard of them is this few rows (the 1024 x 4096 layers of OPT-350M /
    # BLOOM-560M from 4 ranks up): a rank then owes a group several 4096-column factorisations, and they share ONE launch
    # chain (factorize_many) instead of following one another round by round -- one rank of 8, rehearsed: OPT-350M 12.15 ->
    # 11.06 ms per step, BLOOM-560M 12.9 -> 11.4; one of 4 on OPT-350M 20.3 -> 17.0.
    group_wide_rows = 256
    # Wide layers of few rows on one rank (the 1024 x 4096 layers of OPT-350M / BLOOM-560M: a 4096-column factorisation
    # each, for a loop of 1024 rows) go in rounds of `short_rows` stacked rows, and a round's factorisations share ONE launch
    # chain (factorize_many): the chain of 72 narrow launches is what such a layer costs, and six matrices ride it as well
    # as one.  OPT-350M 59.5 -> 51.8 ms per step, BLOOM-560M 69.2 -> 61.6 (four per chain and 4096 rows: 52.7 / 62.3; two:
    # 56.5 / 65.7).  Full-height layers gain nothing from sharing a chain (headline 25.6 ms either way: the chip is full
    # of their wide kernels), 4096 x 11008 layers lose (506 -> 518).
    short_factor_batch = 8  # at most this many factorisations per launch chain (1: one by one, on rotating streams)
    short_rows = 6144

    def group_limit(self, layer, rows):
        """How many layers of this shape, `rows` of them on this rank, go through the loop as one batch (0: round by round).
        Small layers only: their shards are chains of short launches (wants_local_batch), and the stacked factors must fit."""
        if rows <= 0:
            return 0
        if not self.wants_local_batch(layer):
            # wide layers whose SHARDS are a few rows: a rank's factorisations of consecutive rounds in one chain (below)
            if not (self.group_wide_rows and rows <= self.group_wide_rows and self.engine.order_mode_code(self.act_order) in (0, 1)):
                return 0
        n = layer["H"].shape[0]
        padded = (rows + 127) // 128 * 128
        return int(min(64, self.group_rows // padded, self.group_bytes // (8 * n * n)))

    def note_statuses(self, infos, layers, defer=False):
        """The status words of the layers' factorisations (0, or 1 + the failing pivot), one per layer of the stream:
        LinAlgError like the reference's np.linalg.cholesky (sleekit/obq.py:49-50), naming the layer -- now (ONE read
        of all the words), or at _device.raise_pending() when deferred (the words may still be in flight on a side
        stream) or when _device.lazy_errors is set."""
        from . import _device as dev

        def label(l):
            R, n = layers[l]["W"].shape
            return f"quantize_stream: layer {l} ({R} x {n}), compute_hessian_chol"

        have = [l for l, info in enumerate(infos) if info is not None]
        if not have:
            return
        if defer or dev.lazy_errors:
            for l in have:
                dev.note_info(infos[l], label(l), defer=True)
            return
        codes = torch.cat([infos[l].reshape(1) for l in have]).cpu().tolist()
        for l, code in zip(have, codes):
            if code != 0:
                dev.raise_not_pd(code, label(l))

    def run_rows(self, layer, lo, hi, factor):
        eng = self.engine
        if hi <= lo:  # fewer rows than ranks: this rank holds none of this layer
            n, device = layer["W"].shape[1], layer["W"].device
            return dict(Q=torch.empty((0, n), dtype=torch.float32, device=device), idx=torch.empty((0, n), dtype=torch.uint8, device=device),
                        row_err=torch.empty(0, dtype=torch.float32, device=device) if self.with_error else None, rows=(lo, hi))
        W = layer["W"][lo:hi].contiguous()
        sc = layer["scale"][lo:hi].contiguous() if layer.get("scale") is not None else None
        # (lookahead = "alone on the GPU": with overlapping streams the loop takes the window kernel's least-chip-time form)
        res = eng.quantize_layer(W, layer["H"], self.quantizer, sc, self.act_order, self.damp, self.moves, factor=factor[:3],
                                 lookahead=not self.overlap,
                                 want_ls_error=self.with_error and self.moves > 0 and layer.get("symmetric") is True)
        err = None
        if res.ls_error is not None:  # carried through the search (scaled domain: times scale^2)
            err = res.ls_error if sc is None else (res.ls_error * sc) * sc
        elif self.with_error and len(factor) > 3:  # the verdict on H's symmetry came with the factor
            err = eng.row_errors_batch(W[None], res.Q[None], [layer["H"]], factor[3])[0]
        elif self.with_error:
            err = eng.row_errors(W, res.Q, layer["H"])
        return dict(Q=res.Q, idx=res.idx, row_err=err, rows=(lo, hi))


    # -- a whole round at once: the row shards of the G layers of a round go through every kernel together
    min_batch = 2  # (tests set 1 to send single-layer rounds through run_round as well)

    def can_batch(self, round_layers, lo, hi):
        if len(round_layers) < self.min_batch or len(round_layers) > 64 or hi == lo:
            return False
        first = round_layers[0]
        scaled = first.get("scale") is not None
        return all(lay["W"].shape == first["W"].shape and lay["H"].shape == first["H"].shape
                   and (lay.get("scale") is not None) == scaled for lay in round_layers)

    def run_round(self, round_layers, lo, hi, payloads):
        """Shards of the round's layers from their packed factors: unpack into one stacked factor, ONE loop and
        ONE error evaluation over all the layers (engine.run_loop_batch) -- R / G rows of a single layer leave
        most of the chip idle, the round's G shards together are a full layer's worth of rows."""
        from . import _device as dev
        from . import _lib

        B, n = len(round_layers), round_layers[0]["H"].shape[0]
        device = round_layers[0]["W"].device
        # the stacked factors live in a buffer per stream, zeroed once: only the upper triangles are rewritten each round
        # (use on one stream is ordered; rounds on other streams have buffers of their own)
        if not hasattr(self, "_ustacks"):
            self._ustacks = {}
        # (ONE buffer per stream and width, grown to the largest batch seen and sliced: a buffer per batch size pinned up to
        # 4 GiB for every distinct B, and a shorter last group added a second one)
        key = (device.index, dev.stream_handle(), n)
        U = self._ustacks.get(key)
        if U is None or U.shape[0] < B:
            U = self._ustacks[key] = torch.zeros((B, n, n), dtype=torch.float64, device=device)
        U = U[:B]
        import ctypes

        order = torch.empty((B, n), dtype=torch.int64, device=device)
        info = torch.empty(B, dtype=torch.int32, device=device)
        # (every rank runs the same backend settings, so with_error here means the roots packed real verdicts)
        known = torch.empty(B, dtype=torch.int32, device=device) if self.with_error else None
        ptrs = (ctypes.c_void_p * B)(*[dev.ptr(p) for p in payloads])
        # one launch for the round's payloads (a launch per layer is bound by the number of launches: 16 us each)
        _lib.check(_lib.lib.slk_factor_unpack_upper_batch(ptrs, B, n, dev.ptr(U), dev.ptr(order), dev.ptr(info), dev.ptr(known),
                                                          dev.stream_handle()))
        return self._run_stacked(round_layers, lo, hi, order, U, info, known)

    # -- one rank, small layers: a round of same-shaped layers is factored AND looped in launches that cover them all
    local_batch = 8          # layers per such round (1: off)
    local_batch_cols = 1536  # widest layer that takes this route whatever its rows (up to twice that with <= 1024 rows)

    def wants_local_batch(self, layer):
        """Small layers are bound by the host's launch rate (a 768-column layer is ~50 launches of microseconds each):
        batched by shape, a round of them costs the launches of one (OPT-125M, 72 layers: 47.8 -> 17.6 ms; rounds of 16
        or 4096-column layers in them changed nothing).  Big layers fill the chip alone and overlap better on separate
        streams (factor chains beside loops)."""
        R, n = layer["W"].shape
        small = n <= self.local_batch_cols or (n <= 2 * self.local_batch_cols and R <= 1024)
        return self.local_batch > 1 and small and self.engine.order_mode_code(self.act_order) in (0, 1)

    def wants_stacked_loop(self, layer):
        """Wide layers with few rows (OPT-350M / BLOOM-560M's 1024 x 4096): the n^3 factorisation fills the chip, the loop
        does not -- its window kernel runs one workgroup per 16 rows, 64 of them for 1024 rows.  Such layers are
        factored one by one on the factor streams and LOOPED in stacks of 4096 rows (run_round_stacked)."""
        R, n = layer["W"].shape
        return (self.local_batch > 1 and not self.wants_local_batch(layer) and R <= 2048
                and self.engine.order_mode_code(self.act_order) in (0, 1))

    def run_round_stacked(self, round_layers, factors):
        """All rows of a round's layers from their own factors (order, U, info[, symmetry flag]) made elsewhere on this
        GPU: stacked, then ONE loop / error over all of them."""
        order = _stack_views([f[0] for f in factors])
        U = _stack_views([f[1] for f in factors])  # (factors that came out of ONE batched factorisation are a stack already)
        info = torch.cat([f[2] for f in factors])
        known = torch.cat([f[3] for f in factors]) if all(len(f) > 3 for f in factors) else None
        return self._run_stacked(round_layers, 0, round_layers[0]["W"].shape[0], order, U, info, known)

    def run_round_local(self, round_layers):
        """All rows of a round's layers on this rank: batched factorisation (engine.factorize_batch), then the
        stacked loop / local search / error of run_round."""
        eng = self.engine
        n = round_layers[0]["H"].shape[0]
        order, U, info = eng.factorize_batch([lay["H"] for lay in round_layers], n, self.damp, eng.order_mode_code(self.act_order))
        return self._run_stacked(round_layers, 0, round_layers[0]["W"].shape[0], order, U, info, None)

    def _run_stacked(self, round_layers, lo, hi, order, U, info, known):
        """Rows [lo, hi) of every layer of the round through ONE loop and ONE error evaluation, from stacked factors
        order (B, n), U (B, n, n), info (B,); known: the symmetry verdicts of the Hessians, or None (checked here)."""
        eng = self.engine
        B, n = len(round_layers), round_layers[0]["H"].shape[0]
        device = round_layers[0]["W"].device
        rows = hi - lo
        Rp = (rows + 127) // 128 * 128  # the batch entry points want whole 128-row tiles per layer (96 rows at 768 / 8)
        scaled = round_layers[0].get("scale") is not None
        # ragged shard: every layer's rows padded to whole tiles (zero weights, unit scale); rows never interact, so the
        # padding rows are wasted work and nothing else -- they are cut off below.  One launch for the stack either way
        # (engine.stack_rows: a copy per layer was 120 small launches per step for one rank of 8 on OPT-125M).
        W = eng.stack_rows([lay["W"][lo:hi] for lay in round_layers], Rp, 0.0)
        sc = eng.stack_rows([lay["scale"][lo:hi] for lay in round_layers], Rp, 1.0) if scaled else None
        cb = eng.require_uniform(self.quantizer)
        want_idx = cb[0] <= 256  # (the kernels emit uint8 indices)
        if self.moves > 0:
            # local search works in the scaled domain (engine.quantize_layer): scaled copy in, ONE search over the stack
            # (engine.local_search_batch: a search per layer is ten small launches, and the shards of a round on several
            # ranks are a few hundred rows each), de-scale on the way out.  (Padding rows of a ragged shard search too:
            # rows never interact, they are cut off below.)
            Ws = eng.rows_divide(W.view(B * Rp, n), sc.reshape(-1)).view(B, Rp, n) if sc is not None else W
            Q, idx = eng.run_loop_batch(Ws, None, order, U, cb, 32, 8, want_idx=want_idx)
            # Hessians the caller vouches to be symmetric (layer["symmetric"] is True): the search carries every row's error with
            # it (obq.py:254, 290 -- the gain of a move IS the change of the error when H is symmetric), so the layer error
            # needs no product of its own; it comes out in the scaled domain, (W - Qw) = scale (Ws - Q): times scale^2.
            # Per row it is as exact as the reference's own `ls.err` (float32 gains: a few 1e-5 relative on a rare row), the
            # layer's mean agrees with the recomputed product to ~1e-8 (BLOOM-560M, all 96 layers).
            carried = self.with_error and all(lay.get("symmetric") is True for lay in round_layers)
            err = torch.empty((B, Rp), dtype=torch.float32, device=device) if carried else None
            eng.local_search_batch(Ws, Q, [lay["H"] for lay in round_layers], cb, self.moves, idx if want_idx else None, known, err)
            if sc is not None:
                Q = eng.rows_divide(Q.view(B * Rp, n), sc.reshape(-1), invert=True).view(B, Rp, n)
                if err is not None:
                    err = (err * sc) * sc
            if self.with_error and err is None:
                err = eng.row_errors_batch(W, Q, [lay["H"] for lay in round_layers], known)
        else:
            Q, idx = eng.run_loop_batch(W, sc, order, U, cb, 32, 8, want_idx=want_idx, unscale=sc is not None)
            err = eng.row_errors_batch(W, Q, [lay["H"] for lay in round_layers], known) if self.with_error else None
        return [dict(Q=Q[b, :rows], idx=idx[b, :rows] if want_idx else None, row_err=None if err is None else err[b, :rows], rows=(lo, hi),
                     info=info[b:b + 1]) for b in range(B)]


def _guard_inputs(layers, stream):
    """The caller's tensors of `layers` (W, H, scale, mean) are read on the side stream `stream`: tell the caching allocator,
    so that a caller who drops them right after a join=False call does not hand memory the side streams still read back for
    reuse (the allocator then holds the block until `stream` has passed this point)."""
    if stream is None:
        return
    for lay in layers:
        for key in ("W", "H", "scale", "mean"):
            t = lay.get(key)
            if isinstance(t, torch.Tensor) and t.is_cuda:
                t.record_stream(stream)


def _stack_views(tensors):
    """torch.stack -- or, when the tensors already lie one behind the other in one allocation (slices of a batched result),
    a view of that allocation: no copy (six 4096-column factors are 0.8 GB)."""
    t0 = tensors[0]
    step = t0.numel()
    base = t0.untyped_storage().data_ptr()
    if t0.is_contiguous() and all(t.is_contiguous() and t.shape == t0.shape and t.dtype == t0.dtype and t.untyped_storage().data_ptr() == base
                                  and t.storage_offset() == t0.storage_offset() + i * step for i, t in enumerate(tensors)):
        return t0.as_strided((len(tensors),) + tuple(t0.shape), (step,) + tuple(t0.stride()))
    return torch.stack(tensors)


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def _all_gather_words(payload, size):
    """All-gather equal-sized int64 buffers; returns (list of per-rank views, async work)."""
    if rehearse is not None:
        ev = None
        if payload.is_cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
        return [payload] * size, _Done(ev)
    if dist.get_backend() == "nccl":
        out = torch.empty(size * payload.numel(), dtype=payload.dtype, device=payload.device)
        work = dist.all_gather_into_tensor(out, payload, async_op=True)
        return list(out.chunk(size)), work
    outs = [torch.empty_like(payload) for _ in range(size)]
    return outs, dist.all_gather(outs, payload, async_op=True)


def _quantize_stream_local(layers, small, short, backend, join):
    """One rank, a stream with small or short layers in it (quantize_stream).  `small` layers go in batched rounds by
    shape -- factored AND looped together -- one round after the other on alternating streams (a round is a chain of short
    launches: two or three in flight fill the gaps); `short` ones (wide, few rows) go in rounds whose factorisations share
    one launch chain on a factor stream (HipBackend.short_factor_batch) and whose rows are looped as one stack; the rest
    through the usual route."""
    n_layers = len(layers)
    out = [None] * n_layers
    fstreams, _, lstreams = backend.streams()
    side = fstreams is not None
    pool = (fstreams + lstreams) if side else [None]
    here = torch.cuda.current_stream() if side else None

    def on(st):
        return torch.cuda.stream(st) if st is not None else _NullCtx()

    def keep(shards, idxs):
        for l, shard in zip(idxs, shards):
            out[l] = shard
            if here is not None:
                for t in shard.values():
                    if isinstance(t, torch.Tensor):
                        t.record_stream(here)

    for st in pool:
        if st is not None:
            st.wait_stream(here)
    if small:
        rounds, _ = plan_rounds([layers[l] for l in small], backend.local_batch)
        for i, members in enumerate(rounds):
            st = pool[i % len(pool)]
            idxs = [small[m] for m in members]
            if not join:
                _guard_inputs([layers[l] for l in idxs], st)
            with on(st):
                if len(idxs) == 1:
                    lay = layers[idxs[0]]
                    fac = backend.factorize(lay)
                    shards = [dict(backend.run_rows(lay, 0, lay["W"].shape[0], fac), info=fac[2])]
                else:
                    shards = backend.run_round_local([layers[l] for l in idxs])
            keep(shards, idxs)
    if short:
        # a round's factorisations in one launch chain (chains rotate over the factor streams), its stacked loop on the loop
        # stream behind them
        ls = lstreams[0] if side else None
        rotation = getattr(backend, "_factor_rotation", 0) if side else 0
        for members in _short_rounds(layers, short, backend):
            facs, events = [], []
            n_ = layers[members[0]]["H"].shape[0]
            # (a chain of B factorisations holds B x (A, U and two workspace images) of n x n doubles: capped at group_bytes)
            fb = max(1, min(int(getattr(backend, "short_factor_batch", 1)), int(getattr(backend, "group_bytes", 1 << 32)) // (32 * n_ * n_)))
            for i in range(0, len(members), fb):
                job = members[i:i + fb]
                fs = fstreams[rotation % len(fstreams)] if side else None
                rotation += 1
                if not join:
                    _guard_inputs([layers[l] for l in job], fs)
                with on(fs):
                    if len(job) > 1 and hasattr(backend, "factorize_many"):
                        facs.extend(backend.factorize_many([layers[l] for l in job]))
                    elif len(job) > 1:
                        facs.extend(backend.factorize(layers[l]) for l in job)
                    else:
                        facs.append(backend.factorize(layers[job[0]]))
                    if side:
                        ev = torch.cuda.Event()
                        ev.record(fs)
                        events.append(ev)
            if not join:
                _guard_inputs([layers[l] for l in members], ls)
            with on(ls):
                for ev in events:
                    ls.wait_event(ev)
                if len(members) == 1:
                    lay = layers[members[0]]
                    shards = [dict(backend.run_rows(lay, 0, lay["W"].shape[0], facs[0]), info=facs[0][2])]
                else:
                    shards = backend.run_round_stacked([layers[l] for l in members], facs)
                if side:
                    for f in facs:
                        for t in f:
                            t.record_stream(ls)
            keep(shards, members)
        if side:
            backend._factor_rotation = rotation % len(fstreams)
    taken = set(small) | set(short)
    rest = [l for l in range(n_layers) if l not in taken]
    if rest:
        for l, shard in zip(rest, _quantize_stream([layers[l] for l in rest], backend, join=join, _local=False)):
            out[l] = shard
    if join and here is not None:
        for st in pool:
            here.wait_stream(st)
    return out


def _short_rounds(layers, short, backend):
    """Rounds of same-shaped `short` layers whose stacked rows come to about a full layer's worth (4096)."""
    by_shape = {}
    for l in short:
        by_shape.setdefault((tuple(layers[l]["W"].shape), layers[l].get("scale") is not None), []).append(l)
    rounds = []
    for (shape, _), members in by_shape.items():
        per = max(2, min(backend.local_batch, int(getattr(backend, "short_rows", 4096)) // max(shape[0], 1)))
        rounds.extend(members[i:i + per] for i in range(0, len(members), per))
    return rounds


def quantize_stream(layers, backend, comm_device=None, join=True):
    """Quantize `layers` (see _quantize_stream) and REGISTER every layer's factorisation status.

    A Hessian that is not positive definite makes the reference raise numpy.linalg.LinAlgError
    (sleekit/obq.py:49-50, np.linalg.cholesky).  Here the status word of every layer's factorisation --
    made on this rank, or carried by the packed factor another rank sent -- is handed to the backend
    (`backend.note_status`): HipBackend raises LinAlgError naming the layer, at once when `join` is true and
    sleekit_amd._device.lazy_errors is off (the default), otherwise at `_device.raise_pending()`.  Every rank sees
    every layer's status, so every rank raises.

    join=True ends in ONE blocking device -> host read of the status words (a host synchronisation per call).  join=False
    blocks nowhere: `_device.raise_pending()` is then REQUIRED, after synchronising, to see the statuses (the list it reads is
    bounded: _device.PENDING_LIMIT); the inputs may be dropped at once (_guard_inputs).
    """
    out = _quantize_stream(layers, backend, comm_device, join, True)
    note = getattr(backend, "note_statuses", None)
    if note is not None:
        note([None if shard is None else shard.get("info") for shard in out], layers, defer=not join)
    return out


def _group_rounds(rounds, layers, backend, rank, size):
    """Consecutive rounds of one shape joined into groups of at most backend.group_limit(layer, shard rows) layers (the
    loop batch the backend wants for such shards); without that hook, or for big layers (limit below two rounds), every
    round is a group of its own."""
    limit_of = getattr(backend, "group_limit", None)
    groups, key_now, count, limit = [], None, 0, 0
    for g, members in enumerate(rounds):
        first = layers[members[0]]
        same = len({(tuple(layers[l]["W"].shape), layers[l].get("scale") is not None) for l in members}) == 1
        key = (tuple(first["W"].shape), first.get("scale") is not None) if same else None
        if limit_of is not None and key is not None and key == key_now and count + len(members) <= limit:
            groups[-1].append(g)
            count += len(members)
            continue
        groups.append([g])
        key_now, count = key, len(members)
        # The limit must be the SAME on every rank: it decides how many all-gathers there are and how large.  Shard heights
        # differ by one when R % size != 0 (1030 rows on 8 ranks: 129 / 128, on either side of the 128-row padding), so every
        # rank asks with the TALLEST shard, rank 0's = ceil(R / size) -- also for a rank that has no rows of the layer at all.
        lo, hi = row_range(first["W"].shape[0], 0, size)
        limit = int(limit_of(first, hi - lo)) if (limit_of is not None and key is not None) else 0
    return groups


def _quantize_stream(layers, backend, comm_device=None, join=True, _local=True):
    """Quantize `layers` (list of dicts with W (R, n), H (n, n), optional scale (R,)) across the ranks.

    Returns, per layer, this rank's shard: dict(Q, idx, row_err, rows=(lo, hi), info).
    Every rank holds every layer's inputs (W, H are inputs of the path and resident before
    it starts); only the factor travels.

    Layers are taken in ROUNDS of at most G (= world size) layers of ONE shape (plan_rounds: the stream is
    bucketed by shape first, since a model's layer order mixes shapes and the layers are independent); the
    ranks factor one layer of the round each, then ONE collective per round exchanges the packed factors
    (status, order, upper triangle of U: n (n + 1) / 2 + n + 1 words per layer).  It is an all-gather -- G simultaneous one-to-all
    broadcasts -- because xGMI is a point-to-point mesh: every rank then receives over all of its
    7 links at once, where G separate ring broadcasts would each crawl through one link per hop.

    Three kinds of queues per rank when the backend provides device streams (`backend.streams()`):
      factor streams : the n x n factorisations this rank owns (latency-bound chains)
      comm stream    : pack + all-gather of each round, behind that round's factor event
      loop streams   : the row loops, each behind its round's collective
    so the factorisation of round g+1 runs under the loops and errors of round g.

    `join` (default): the caller's stream waits for the loop streams before this returns, so the results can
    be used on it at once.  With join=False nothing waits: the NEXT call's factorisations then start under
    this call's loops (a throughput loop over independent batches, bench.py) -- synchronise the device, or
    the loop streams, before reading the results.  The INPUT tensors may be dropped when the call returns: every side
    stream that reads them is recorded on them (_guard_inputs), so their memory is not reused before those streams pass.
    """
    rank, size = world()
    n_layers = len(layers)
    if size == 1 and not always_exchange and _local and hasattr(backend, "run_round_local"):
        # one rank: small layers go in rounds of one shape, factored and looped in launches that cover the round
        small = [l for l in range(n_layers) if backend.wants_local_batch(layers[l])]
        short = [l for l in range(n_layers) if backend.wants_stacked_loop(layers[l])]
        small, short = (small if len(small) > 1 else []), (short if len(short) > 1 else [])
        if small or short:
            return _quantize_stream_local(layers, small, short, backend, join)
    fstreams, cstream, lstreams = backend.streams() if hasattr(backend, "streams") else (None, None, None)
    side = fstreams is not None
    here = torch.cuda.current_stream() if side else None
    exchange = size > 1 or always_exchange
    # rounds of one shape each; root[l] = the rank that factors layer l (plan_rounds)
    rounds, root = plan_rounds(layers, size, bucket=exchange and getattr(backend, "bucket_by_shape", True))
    n_rounds = len(rounds)
    factors = [None] * n_layers
    ready = [None] * n_layers

    def on(stream):
        return torch.cuda.stream(stream) if stream is not None else _NullCtx()

    # Rounds of SMALL layers are taken in GROUPS (the backend says how many layers it wants in one loop batch:
    # group_limit): a rank factors all its layers of a group in one batched factorisation, the group's payloads cross in ONE
    # all-gather (a slot per round in every rank's buffer) and its row shards go through the loop as ONE stack.  The shards
    # of a small layer on several ranks are a chain of short launches whatever their height (one rank of 8 on OPT-125M: 10
    # rounds x 15 loop launches and 9 factorisations x 45 launches per step, as long as the whole model on one GPU);
    # groups make it 3 x 15 and 3 x 45.  A group of one round is the plain scheme.
    groups = _group_rounds(rounds, layers, backend, rank, size) if exchange else [[g] for g in range(n_rounds)]
    slot = {}  # layer -> its round's slot in the group's buffers
    for members_g in groups:
        for j, g in enumerate(members_g):
            for l in rounds[g]:
                slot[l] = j

    # 1. every rank factors the layers it is the root of (concurrently across ranks), in processing order; the
    #    factorisations are latency-bound chains, so consecutive ones alternate between streams
    #    (the rotation carries on from the previous call: with one layer per rank and call -- a round of G
    #    layers on G ranks -- consecutive calls would otherwise queue on the same stream, one chain behind the other)
    jobs = []  # lists of layers factored together
    for members_g in groups:
        own = [l for g in members_g for l in rounds[g] if root[l] == rank]
        if len(members_g) > 1 and len(own) > 1 and hasattr(backend, "factorize_many"):
            # (a chain of B factorisations holds B x (A, U and two workspace images) of n x n doubles: capped like the one-rank
            # path's chains, _quantize_stream_local)
            n_ = layers[own[0]]["H"].shape[0]
            cap = max(1, int(getattr(backend, "group_bytes", 1 << 32)) // (32 * n_ * n_))
            jobs.extend(own[i:i + cap] for i in range(0, len(own), cap))
        else:
            jobs.extend([l] for l in own)
    first = getattr(backend, "_factor_rotation", 0) if side else 0
    if side:
        backend._factor_rotation = (first + len(jobs)) % len(fstreams)
    stream_of = {}  # layer -> the stream its factorisation went to
    for k, job in enumerate(jobs):
        fs = fstreams[(first + k) % len(fstreams)] if side else None
        if side and k < len(fstreams):
            fs.wait_stream(here)
        if not join:
            _guard_inputs([layers[l] for l in job], fs)
        with on(fs):
            made = backend.factorize_many([layers[l] for l in job]) if len(job) > 1 else [backend.factorize(layers[job[0]])]
            ev = None
            if side:
                ev = torch.cuda.Event()
                ev.record(fs)
        for l, fac in zip(job, made):
            factors[l], ready[l], stream_of[l] = fac, ev, fs
    # 2. one asynchronous all-gather per group of rounds, issued in order
    gathered = [None] * len(groups)  # (per-rank buffers, work, words per payload, words per slot)
    keep = []
    if exchange:
        if cstream is not None:
            cstream.wait_stream(here)
        with on(cstream):
            for gi, members_g in enumerate(groups):
                members = [l for g in members_g for l in rounds[g]]
                # equal-sized contributions: a group's layers share one shape (a lone round of mixed shapes: the widest)
                words = max(backend.payload_words(layers[j]["H"].shape[0]) for j in members)
                stride = (words + 1) // 2 * 2 if len(members_g) > 1 else words
                dev_ = comm_device if comm_device is not None else layers[0]["H"].device
                buffer = None if len(members_g) == 1 else backend.alloc_payload(stride * len(members_g), dev_)
                for j, g in enumerate(members_g):
                    own = [l for l in rounds[g] if root[l] == rank]  # at most one: a round has at most `size` layers
                    if own:
                        l = own[0]
                        if ready[l] is not None:
                            cstream.wait_event(ready[l])
                        payload = backend.pack(factors[l], words)
                        if side:
                            for t in factors[l]:
                                t.record_stream(cstream)  # made on a factor stream, read here
                    elif rehearse is not None:
                        # rehearsal: a real factor of the right shape must stand in for the peers' (a blank one is not a
                        # permutation + triangle the kernels can run on); made once per shape, outside what is measured
                        n_ = layers[rounds[g][0]]["H"].shape[0]
                        if (n_, words) not in _rehearsal_payloads:
                            _rehearsal_payloads[(n_, words)] = backend.pack(backend.factorize(layers[rounds[g][0]]), words)
                        payload = _rehearsal_payloads[(n_, words)]
                    elif buffer is None:  # no layer of this round is this rank's: contribute a blank
                        payload = backend.alloc_payload(words, dev_).zero_()
                    else:
                        payload = None  # (a slot nobody reads)
                    if buffer is None:
                        buffer = payload
                    elif payload is not None:
                        buffer[j * stride:j * stride + words].copy_(payload)
                log = getattr(backend, "exchange_log", None)
                if log is not None and buffer.is_cuda:
                    # measurement (bench.py's pass with events): the comm stream brackets the collective with two events
                    # and waits for it, so that their distance is the exchange itself
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(torch.cuda.current_stream())
                    parts, work = _all_gather_words(buffer, size)
                    work.wait()
                    e1.record(torch.cuda.current_stream())
                    log.append((e0, e1, buffer.numel() * buffer.element_size() * (size - 1)))
                else:
                    parts, work = _all_gather_words(buffer, size)
                gathered[gi] = (parts, work, words, stride)
                keep.append(buffer)

    def payload_of(gi, l):
        parts, _, words, stride = gathered[gi]
        return parts[root[l]][slot[l] * stride:slot[l] * stride + words]

    # 3. every rank runs its rows of every layer as the factors land, group by group.  A group goes through the
    #    kernels as ONE batch when the backend can do that (run_round): the shards are R / G rows each, too few to
    #    fill the chip alone.  Otherwise layer by layer, consecutive ones on alternating streams (their leaf chains
    #    are latency-bound).
    out = [None] * n_layers
    if side:
        for st in lstreams:
            st.wait_stream(here)
    # (batched rounds rotate over the loop streams across calls, like the factorisations above)
    batched_rounds = getattr(backend, "_loop_rotation", 0) if side else 0
    for gi, members_g in enumerate(groups):
        members = [l for g in members_g for l in rounds[g]]
        lo, hi = row_range(layers[members[0]]["W"].shape[0], rank, size)
        if exchange and hasattr(backend, "run_round") and backend.can_batch([layers[l] for l in members], lo, hi):
            ls = lstreams[batched_rounds % len(lstreams)] if side else None
            own = [l for l in members if l in stream_of]
            if side and own and getattr(backend, "rounds_on_factor_streams", True):
                # The round runs on the stream that factored this rank's layer of it: one stream fewer to pair badly.
                # HIP multiplexes all streams of a process onto GPU_MAX_HW_QUEUES (4 by default) in-order hardware queues;
                # a loop stream that shares its queue with the factor stream of the NEXT round puts that factorisation
                # behind this round's loops and the pipeline collapses into factor -> loop -> factor (4.4 ... 7.0 ms per
                # step at N = 8 depending on how the streams fell).  Multi-rank runs should raise GPU_MAX_HW_QUEUES to 8
                # before HIP starts (bench.py does): every stream, RCCL's included, then has a queue of its own.
                ls = stream_of[own[0]]
            batched_rounds += 1
            if side:
                backend._loop_rotation = batched_rounds % len(lstreams)
            if not join:
                _guard_inputs([layers[l] for l in members], ls)
            with on(ls):
                parts, work = gathered[gi][:2]
                work.wait()  # orders the stream behind the transfer; no host block on GPU
                shards = backend.run_round([layers[l] for l in members], lo, hi, [payload_of(gi, l) for l in members])
                if side:
                    for t in parts:
                        t.record_stream(ls)
            for l, shard in zip(members, shards):
                out[l] = shard
            continue
        for l in members:
            layer = layers[l]
            ls = lstreams[l % len(lstreams)] if side else None
            if not join:
                _guard_inputs([layer], ls)
            with on(ls):
                if exchange:
                    parts, work = gathered[gi][:2]
                    work.wait()
                    if factors[l] is None or always_exchange:
                        factors[l] = backend.unpack(payload_of(gi, l), layer["H"].shape[0])
                elif ready[l] is not None:
                    ls.wait_event(ready[l])
                lo, hi = row_range(layer["W"].shape[0], rank, size)
                shard = backend.run_rows(layer, lo, hi, factors[l])
                if side:  # made on a factor / comm stream, read on this one
                    for t in factors[l]:
                        t.record_stream(ls)
                    if exchange:
                        for t in gathered[gi][0]:
                            t.record_stream(ls)
            shard["info"] = factors[l][2]
            out[l] = shard
    if side:
        if join:
            for st in lstreams + fstreams:  # (batched rounds run on factor streams)
                here.wait_stream(st)
        # tensors made on the side streams are consumed on the caller's stream: keep the allocator honest
        extra = [(p,) for p in keep if p is not None] + [tuple(g[0]) for g in gathered if g is not None]
        for f in factors + extra:
            for t in f or ():
                t.record_stream(here)
        for shard in out:
            for t in shard.values():
                if isinstance(t, torch.Tensor):
                    t.record_stream(here)
    return out


def verify_exchange(layers, backend, max_rounds=2):
    """Self-check of the collective: for the first rounds of the stream, every rank packs the factor of its own layer,
    the payloads are all-gathered exactly as quantize_stream does, and every rank's copy of every payload is held to its
    ROOT's own buffer by two wrapping int64 checksums (plain and position-weighted), compared across ranks through a
    second, tiny all-gather.  Returns dict(world_size, backend, rounds_checked, payload_bytes, agree)."""
    rank, size = world()
    out = dict(world_size=size, backend=dist.get_backend() if size > 1 and rehearse is None else "none", rounds_checked=0,
               payload_bytes=0, agree=True)
    if size <= 1 or rehearse is not None:
        return out
    rounds, root = plan_rounds(layers, size)
    for members in rounds[:max_rounds]:
        words = max(backend.payload_words(layers[j]["H"].shape[0]) for j in members)
        own = [l for l in members if root[l] == rank]
        payload = backend.pack(backend.factorize(layers[own[0]]), words) if own else backend.alloc_payload(words, layers[0]["H"].device).zero_()
        parts, work = _all_gather_words(payload, size)
        work.wait()
        weight = torch.arange(1, words + 1, dtype=torch.int64, device=payload.device)

        def sums(t):
            return torch.stack([t.sum(), (t * weight).sum()])

        seen = torch.stack([sums(p) for p in parts])  # (size, 2): this rank's view of every rank's payload
        views = [torch.empty_like(seen) for _ in range(size)]
        dist.all_gather(views, seen)
        views = torch.stack(views).cpu()  # [viewer][source][2]
        mine = sums(payload).cpu()
        ok = bool(torch.equal(views[rank][rank], mine))
        for src in range(size):
            ok = ok and all(bool(torch.equal(views[viewer][src], views[src][src])) for viewer in range(size))
        out["agree"] = out["agree"] and ok
        out["rounds_checked"] += 1
        out["payload_bytes"] = int(words * 8)
    return out


def layer_error(shards_row_err, R):
    """Mean over all R rows of a layer from this rank's row errors (sum-all-reduce of one scalar).

    Bookkeeping outside the timed path: the reference's layer error is the host-side mean.
    """
    total = shards_row_err.double().sum()
    rank, size = world()
    if size > 1:
        dist.all_reduce(total)
    return total / R
