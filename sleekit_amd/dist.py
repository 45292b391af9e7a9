"""Row-sharded quantization of a stream of layers over the GPUs of one node.

Given the shared factor U, every output row of W goes through the loop, the local search
and the error on its own (sleekit/obq.py:106-137, 264-346, 89-95 have no cross-row term),
so the path shards by rows with no data-path collective.  What does NOT shard is the
n x n factorisation; over a stream of layers it is spread instead: rank (l mod G) factors
layer l and broadcasts (order, U, status) once -- one RCCL broadcast per layer over xGMI --
while every rank runs rows [r R/G, (r+1) R/G) of every layer.

One process per GPU; `torch.distributed` must be initialised by the caller (backend "nccl"
is RCCL on ROCm).  The broadcasts are issued asynchronously up front, so layer l's loop
overlaps the transfer of layer l+1's factor; on a single rank nothing is communicated.

The module is engine-agnostic: `backend` supplies factorize / run_rows, which lets the CPU
test-suite drive the same scheduling code over gloo with a stand-in backend.
"""

import torch
import torch.distributed as dist


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def row_range(R, rank, size):
    """Rows [lo, hi) of rank `rank`: contiguous, sizes differ by at most one, covers [0, R) exactly."""
    base, extra = divmod(R, size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def factor_root(layer_index, size):
    return layer_index % size


class HipBackend:
    """The real thing: sleekit_amd.engine on the current device."""

    def __init__(self, quantizer, act_order="diag", damp=0.01, nb_ls_moves=0, with_error=True, overlap=True):
        from . import engine

        self.engine, self.quantizer, self.overlap = engine, quantizer, overlap
        self.act_order, self.damp, self.moves, self.with_error = act_order, damp, nb_ls_moves, with_error

    def streams(self):
        """(factor streams, comm stream, loop streams), created once; (None, None, None) = everything in order."""
        if not self.overlap:
            return None, None, None
        if not hasattr(self, "_streams"):
            nf, nl = (self.overlap if isinstance(self.overlap, tuple) else (2, 2))
            self._streams = ([torch.cuda.Stream() for _ in range(nf)], torch.cuda.Stream(),
                             [torch.cuda.Stream() for _ in range(nl)])
        return self._streams

    def alloc_payload(self, n, device):
        from . import _lib

        return torch.empty(int(_lib.lib.slk_factor_payload_words(n)), dtype=torch.int64, device=device)

    def pack(self, factor):
        """(order, U, info) -> one int64-word buffer: status, order, packed upper triangle of U."""
        from . import _device as dev
        from . import _lib

        order, U, info = factor
        n = U.shape[0]
        payload = self.alloc_payload(n, U.device)
        _lib.check(_lib.lib.slk_factor_pack(dev.ptr(U), dev.ptr(order), dev.ptr(info), n, dev.ptr(payload), dev.stream_handle()))
        return payload

    def unpack(self, payload, n):
        from . import _device as dev
        from . import _lib

        order = torch.empty(n, dtype=torch.int64, device=payload.device)
        U = torch.empty((n, n), dtype=torch.float64, device=payload.device)
        info = torch.empty(1, dtype=torch.int32, device=payload.device)
        _lib.check(_lib.lib.slk_factor_unpack(dev.ptr(payload), n, dev.ptr(U), dev.ptr(order), dev.ptr(info), dev.stream_handle()))
        return order, U, info

    def factorize(self, layer):
        eng = self.engine
        W, H, n = layer["W"], layer["H"], layer["H"].shape[0]
        mode = eng.order_mode_code(self.act_order)
        miss = None
        if mode >= 2:  # err / sqerr need the statistics of ALL rows, before sharding
            cb = eng.require_uniform(self.quantizer)
            Ws = eng.rows_divide(W, layer["scale"]) if layer.get("scale") is not None else W
            miss = eng.column_miss(Ws, cb, mode == 3)
        return eng.factorize(H, n, self.damp, mode, miss)

    def run_rows(self, layer, lo, hi, factor):
        eng = self.engine
        W = layer["W"][lo:hi].contiguous()
        sc = layer["scale"][lo:hi].contiguous() if layer.get("scale") is not None else None
        res = eng.quantize_layer(W, layer["H"], self.quantizer, sc, self.act_order, self.damp, self.moves, factor=factor)
        err = eng.row_errors(W, res.Q, layer["H"]) if self.with_error else None
        return dict(Q=res.Q, idx=res.idx, row_err=err, rows=(lo, hi))


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def quantize_stream(layers, backend, comm_device=None):
    """Quantize `layers` (list of dicts with W (R, n), H (n, n), optional scale (R,)) across the ranks.

    Returns, per layer, this rank's shard: dict(Q, idx, row_err, rows=(lo, hi), info).
    Every rank holds every layer's inputs (W, H are inputs of the path and resident before
    it starts); only the factor travels.

    Three queues per rank when the backend provides device streams (`backend.streams()`):
      factor stream : the n x n factorisations this rank owns, back to back
      comm stream   : one broadcast per layer, each behind its factor's event
      main stream   : the row loops, each behind its layer's factor / broadcast
    so the latency-bound factorisation of layer l+1 runs under the loop and error of layer l.
    """
    rank, size = world()
    n_layers = len(layers)
    factors = [None] * n_layers
    ready = [None] * n_layers
    fstreams, cstream, lstreams = backend.streams() if hasattr(backend, "streams") else (None, None, None)
    side = fstreams is not None
    here = torch.cuda.current_stream() if side else None

    def on(stream):
        return torch.cuda.stream(stream) if stream is not None else _NullCtx()

    # 1. every rank factors the layers it is the root of (concurrently across ranks); the
    #    factorisations are latency-bound chains, so consecutive ones alternate between streams
    mine = list(range(rank, n_layers, size))
    for k, l in enumerate(mine):
        fs = fstreams[k % len(fstreams)] if side else None
        if side and k < len(fstreams):
            fs.wait_stream(here)
        with on(fs):
            factors[l] = backend.factorize(layers[l])
            if side:
                ready[l] = torch.cuda.Event()
                ready[l].record(fs)
    # 2. ONE asynchronous broadcast per layer from its root, issued in layer order: the root packs
    #    (status, order, upper triangle of U) into a single buffer, the others unpack it
    pending = [None] * n_layers
    payloads = [None] * n_layers
    if size > 1:
        if cstream is not None:
            cstream.wait_stream(here)
        with on(cstream):
            for l in range(n_layers):
                root = factor_root(l, size)
                n = layers[l]["H"].shape[0]
                if root == rank:
                    if ready[l] is not None:
                        cstream.wait_event(ready[l])
                    payloads[l] = backend.pack(factors[l])
                else:
                    dev_ = comm_device if comm_device is not None else layers[l]["H"].device
                    payloads[l] = backend.alloc_payload(n, dev_)
                pending[l] = dist.broadcast(payloads[l], src=root, async_op=True)
    # 3. every rank runs its rows of every layer as the factors land (loops of consecutive layers
    #    alternate between streams too: their leaf chains are latency-bound as well)
    out = []
    for l, layer in enumerate(layers):
        ls = lstreams[l % len(lstreams)] if side else None
        if side and l < len(lstreams):
            ls.wait_stream(here)
        with on(ls):
            if pending[l] is not None:
                pending[l].wait()  # orders the stream behind the transfer; no host block on GPU
                if factors[l] is None:
                    factors[l] = backend.unpack(payloads[l], layer["H"].shape[0])
            elif ready[l] is not None:
                ls.wait_event(ready[l])
            lo, hi = row_range(layer["W"].shape[0], rank, size)
            shard = backend.run_rows(layer, lo, hi, factors[l])
        shard["info"] = factors[l][2]
        out.append(shard)
    if side:
        for st in lstreams:
            here.wait_stream(st)
        # tensors made on the side streams are consumed on the caller's stream: keep the allocator honest
        for f in factors + [(p,) for p in payloads if p is not None]:
            for t in f:
                t.record_stream(here)
        for shard in out:
            for t in shard.values():
                if isinstance(t, torch.Tensor):
                    t.record_stream(here)
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
