"""Device-resident pipeline for one layer: the hot path end to end.

    [W / scale] -> column statistics -> damp + order + gather -> factor (float64)
               -> blocked quantize / propagate loop -> [local search] -> [un-scale]

Every stage is one call into libsleekit_amd.so on the current torch stream; tensors stay
in HBM and nothing synchronises unless the caller asks for NumPy results.  This is the
function `sleekit_amd.obq.quantize_opt` and `sleekit_amd.scaling.quantize_with_scaling`
(the reference's entry points, sleekit/obq.py:169 and sleekit/scaling.py:58) are thin
wrappers of, and what bench.py times.
"""

import ctypes

import torch

from . import _device as dev
from . import _lib
from .codebook import DeviceCodebook

_INVERSE_ORDERS = {"inv_diag": 0, "combined_diag": 1}  # need diag(Hd^-1): two factorisations
_KEY_ORDERS = ("inv_diag", "combined_diag", "pivot")     # sort keys computed by a kernel of their own


def order_mode_code(act_order):
    if act_order in _lib.ORDER_MODES:
        return _lib.ORDER_MODES[act_order]
    if act_order in _KEY_ORDERS:
        return _lib.ORDER_KEYS
    raise RuntimeError(f"Invalid act_order value {act_order}")


def inverse_diag_keys(H, n, damp, combined):
    """Sort keys of inv_diag / combined_diag (obq.py:70-75) from a first factorisation in the original order."""
    _, U0, info = factorize(H, n, damp, _lib.ORDER_NONE)
    dev.note_info(info, "compute_hessian_chol")
    ws, ws_bytes = dev.workspace(0, n)
    keys = torch.empty(n, dtype=torch.float64, device=H.device)
    _lib.check(
        _lib.lib.slk_inverse_diag_keys(
            dev.ptr(U0), dev.ptr(H), n, float(damp), int(combined), dev.ptr(keys), dev.ptr(ws), ws_bytes, dev.stream_handle()
        )
    )
    return keys


def pivot_keys(H, n, damp):
    """Sort keys of the greedy pivoted-Cholesky order (obq.py:78, 140-166): the step at which each column is picked."""
    ws, ws_bytes = dev.workspace(0, n)
    keys = torch.empty(n, dtype=torch.float64, device=H.device)
    _lib.check(_lib.lib.slk_pivot_keys(dev.ptr(H), n, float(damp), dev.ptr(keys), dev.ptr(ws), ws_bytes, dev.stream_handle()))
    return keys


def order_keys(H, n, damp, act_order):
    """Keys of the orders that are not a function of the damped diagonal and the column statistics alone."""
    if act_order == "pivot":
        return pivot_keys(H, n, damp)
    return inverse_diag_keys(H, n, damp, _INVERSE_ORDERS[act_order])


def require_uniform(quantizer):
    """(levels, lo, hi, table) of a UniformCodebook or a general Codebook; anything else has no device form."""
    if not isinstance(quantizer, DeviceCodebook):
        raise NotImplementedError(
            "sleekit_amd runs UniformCodebook / Codebook quantizers on the GPU and has no CPU fallback for arbitrary "
            f"callables (got {type(quantizer).__name__})"
        )
    abi = quantizer._abi()
    if len(quantizer) > 256 and abi[3] is not None:
        raise NotImplementedError("general (table) codebooks on the device hold at most 256 entries")
    return abi


class LayerResult:
    """Device tensors produced for one layer."""

    __slots__ = ("Q", "idx", "order", "U", "info", "E", "ls_trace", "ls_error")

    def __init__(self):
        self.Q = self.idx = self.order = self.U = self.info = self.E = self.ls_trace = self.ls_error = None


def factorize(H, n, damp, mode, miss=None, keep=None, lookahead=False):
    """Damping + order + float64 factor for a float32 device Hessian. Returns (order, U, info).
    lookahead: one layer at a time (latency matters, nothing else is in flight): the factorisation forks the bulk of
    its outer updates onto a helper stream (slk_chol_inverse_upper_lookahead: an argument of THIS call, no process-wide
    switch; off for the multi-stream pipelines of sleekit_amd.dist, where the extra streams cost more than they save)."""
    ws, ws_bytes = dev.workspace(0, n)
    s = dev.stream_handle()
    ld = _lib.lib.slk_factor_ld(n)
    order = torch.empty(n, dtype=torch.int64, device=H.device)
    A = torch.empty(ld * ld, dtype=torch.float64, device=H.device)
    U = torch.empty((n, n), dtype=torch.float64, device=H.device)
    info = torch.empty(1, dtype=torch.int32, device=H.device)
    _lib.check(
        _lib.lib.slk_hessian_prepare(
            dev.ptr(H), n, float(damp), mode, dev.ptr(miss), dev.ptr(order), dev.ptr(A), dev.ptr(ws), ws_bytes, s
        )
    )
    chol = _lib.lib.slk_chol_inverse_upper_lookahead if lookahead else _lib.lib.slk_chol_inverse_upper
    _lib.check(chol(dev.ptr(A), n, dev.ptr(U), dev.ptr(info), dev.ptr(ws), ws_bytes, s))
    return order, U, info


def factorize_batch(Hs, n, damp, mode):
    """Damping + order + float64 factor of B same-sized float32 device Hessians in launches that cover them all
    (slk_hessian_prepare_batch / slk_chol_inverse_upper_batch).  Returns order (B, n), U (B, n, n), info (B,):
    the results of B calls of factorize."""
    B = len(Hs)
    device = Hs[0].device
    assert all(H.shape == (n, n) and H.is_contiguous() and H.dtype == torch.float32 for H in Hs)
    ws, ws_bytes = dev.scratch(_lib.lib.slk_factor_workspace_bytes_batch(B, n), "factor_batch")
    s = dev.stream_handle()
    ld = _lib.lib.slk_factor_ld(n)
    order = torch.empty((B, n), dtype=torch.int64, device=device)
    A = torch.empty(B * ld * ld, dtype=torch.float64, device=device)
    U = torch.empty((B, n, n), dtype=torch.float64, device=device)
    info = torch.empty(B, dtype=torch.int32, device=device)
    ptrs = (ctypes.c_void_p * B)(*[dev.ptr(H) for H in Hs])
    _lib.check(_lib.lib.slk_hessian_prepare_batch(ptrs, B, n, float(damp), mode, dev.ptr(order), dev.ptr(A), dev.ptr(ws), ws_bytes, s))
    _lib.check(_lib.lib.slk_chol_inverse_upper_batch(dev.ptr(A), B, n, dev.ptr(U), dev.ptr(info), dev.ptr(ws), ws_bytes, s))
    return order, U, info


def factorize_order_only(H, n, mode, miss=None):
    """Column order of an already damped float32 Hessian (compute_hessian_order, obq.py:58-86)."""
    ws, ws_bytes = dev.workspace(0, n)
    ld = _lib.lib.slk_factor_ld(n)
    order = torch.empty(n, dtype=torch.int64, device=H.device)
    A = torch.empty(ld * ld, dtype=torch.float64, device=H.device)
    _lib.check(
        _lib.lib.slk_hessian_prepare(
            dev.ptr(H), n, 0.0, mode, dev.ptr(miss), dev.ptr(order), dev.ptr(A), dev.ptr(ws), ws_bytes,
            dev.stream_handle(),
        )
    )
    return order, None, None


def run_loop(W, scale, order, U, cb_abi, min_block, num_blocks, want_idx=True, want_E=False, unscale=False, latency=False):
    """The column-sequential loop on device tensors. Returns (Q, idx, E); `unscale`: Q comes back de-scaled; `latency`:
    this layer is alone on the GPU (SLK_LOOP_LATENCY: 16-row window workgroups; same results)."""
    R, n = W.shape
    levels, lo, hi, table = cb_abi
    ws, ws_bytes = dev.workspace(R, n)
    Q = torch.empty((R, n), dtype=torch.float32, device=W.device)
    idx = torch.empty((R, n), dtype=torch.uint8, device=W.device) if want_idx else None
    E = torch.empty((R, n), dtype=torch.float32, device=W.device) if want_E else None
    _lib.check(
        _lib.lib.slk_gptq_quantize(
            dev.ptr(W), dev.ptr(scale), dev.ptr(order), dev.ptr(U), R, n, levels, lo, hi, dev.ptr(table), int(min_block),
            int(num_blocks), (1 if unscale else 0) | (2 if latency else 0), dev.ptr(Q), dev.ptr(idx), dev.ptr(E), dev.ptr(ws), ws_bytes,
            dev.stream_handle(),
        )
    )
    return Q, idx, E


def run_loop_batch(W, scale, order, U, cb_abi, min_block, num_blocks, want_idx=True, unscale=False):
    """The loop over a batch of layers stacked by rows (slk_gptq_quantize_batch).

    W (B, R, n) float32, scale (B, R) or None, order (B, n) int64, U (B, n, n) float64, all contiguous.
    Returns (Q, idx) shaped (B, R, n): what B calls of run_loop return, in launches that cover all B layers.
    """
    B, R, n = W.shape
    assert order.shape == (B, n) and U.shape == (B, n, n) and (scale is None or scale.shape == (B, R))
    assert W.is_contiguous() and order.is_contiguous() and U.is_contiguous() and (scale is None or scale.is_contiguous())
    levels, lo, hi, table = cb_abi
    ws, ws_bytes = dev.workspace(R, n, batch=B)
    Q = torch.empty((B, R, n), dtype=torch.float32, device=W.device)
    idx = torch.empty((B, R, n), dtype=torch.uint8, device=W.device) if want_idx else None
    _lib.check(
        _lib.lib.slk_gptq_quantize_batch(
            dev.ptr(W), dev.ptr(scale), dev.ptr(order), dev.ptr(U), B, R, n, levels, lo, hi, dev.ptr(table), int(min_block),
            int(num_blocks), 1 if unscale else 0, dev.ptr(Q), dev.ptr(idx), None, dev.ptr(ws), ws_bytes, dev.stream_handle(),
        )
    )
    return Q, idx


def row_errors_batch(W, Q, Hs, symmetric=None):
    """Row errors of a batch of layers stacked by rows: W, Q (B, R, n); Hs a list of B (n, n) float32 tensors.
    symmetric: (B,) int32 verdicts of symmetry_flag (None: checked here)."""
    B, R, n = W.shape
    assert len(Hs) == B and W.is_contiguous() and Q.is_contiguous()
    assert symmetric is None or (symmetric.dtype == torch.int32 and symmetric.numel() == B and symmetric.is_contiguous())
    ws, ws_bytes = dev.workspace(R, n, batch=B)
    out = torch.empty((B, R), dtype=torch.float32, device=W.device)
    ptrs = (ctypes.c_void_p * B)(*[dev.ptr(H) for H in Hs])
    _lib.check(
        _lib.lib.slk_row_errors_batch(dev.ptr(W), dev.ptr(Q), ptrs, B, R, n, dev.ptr(symmetric), dev.ptr(out), dev.ptr(ws), ws_bytes,
                                      dev.stream_handle())
    )
    return out


def symmetry_flag(H):
    """int32[1] on the device: 1 iff H is bit-wise symmetric (what lets the layer error skip half its products)."""
    flag = torch.empty(1, dtype=torch.int32, device=H.device)
    _lib.check(_lib.lib.slk_symmetry_flag(dev.ptr(H), H.shape[0], dev.ptr(flag), dev.stream_handle()))
    return flag


def local_search(W, Q, H, cb_abi, moves, idx=None, want_trace=False, gains=None, gains_mode=0, row_err=None):
    """In place on Q (and idx).  want_trace: returns the (R, moves) int32 record of the moves taken
    (2 * column + up, -1 = none); gains / gains_mode: the carried state of a stateful search (slk_local_search);
    row_err (R,) float32: receives the rows' errors (W - Q) H (W - Q)^T after the moves."""
    R, n = W.shape
    levels, lo, hi, table = cb_abi
    ws, ws_bytes = dev.workspace(R, n)
    trace = torch.empty((R, int(moves)), dtype=torch.int32, device=W.device) if want_trace else None
    assert gains is None or (gains.shape == (R, 2, n) and gains.dtype == torch.float32 and gains.is_contiguous())
    _lib.check(
        _lib.lib.slk_local_search(
            dev.ptr(W), dev.ptr(Q), dev.ptr(H), R, n, levels, lo, hi, dev.ptr(table), int(moves), dev.ptr(idx), dev.ptr(trace),
            dev.ptr(gains), int(gains_mode), dev.ptr(row_err), dev.ptr(ws), ws_bytes, dev.stream_handle(),
        )
    )
    return trace


def local_search_batch(W, Q, Hs, cb_abi, moves, idx=None, symmetric=None, row_err=None):
    """local_search over a batch of layers stacked by rows: W, Q (B, R, n) (idx (B, R, n) uint8 or None), Hs a list of B
    Hessians; in place on Q and idx, the results of B separate searches.  row_err (B, R) float32: receives the rows' errors
    after the moves, in the domain of W and Q (carried through the search: no product of its own)."""
    B, R, n = W.shape
    assert len(Hs) == B and W.is_contiguous() and Q.is_contiguous() and (idx is None or idx.is_contiguous())
    levels, lo, hi, table = cb_abi
    ws, ws_bytes = dev.workspace(R, n, batch=B)
    ptrs = (ctypes.c_void_p * B)(*[dev.ptr(H) for H in Hs])
    _lib.check(
        _lib.lib.slk_local_search_batch(dev.ptr(W), dev.ptr(Q), ptrs, B, R, n, levels, lo, hi, dev.ptr(table), int(moves), dev.ptr(idx),
                                        dev.ptr(symmetric), dev.ptr(row_err), dev.ptr(ws), ws_bytes, dev.stream_handle())
    )


def stack_rows(parts, rows_padded, fill=0.0):
    """torch.stack of the layers' row shards (each rows x cols, or rows,) with every layer padded to `rows_padded` rows of
    `fill`: ONE launch (slk_stack_rows) instead of a copy per layer."""
    B = len(parts)
    rows = parts[0].shape[0]
    cols = parts[0].shape[1] if parts[0].dim() == 2 else 1
    for t in parts:
        if t.dtype != torch.float32 or not t.is_contiguous() or t.shape != parts[0].shape:
            raise ValueError("stack_rows wants contiguous float32 shards of one shape")
    out = torch.empty((B, rows_padded, cols) if parts[0].dim() == 2 else (B, rows_padded), dtype=torch.float32, device=parts[0].device)
    ptrs = (ctypes.c_void_p * B)(*[dev.ptr(t) for t in parts])
    _lib.check(_lib.lib.slk_stack_rows(ptrs, B, rows, int(rows_padded), cols, float(fill), dev.ptr(out), dev.stream_handle()))
    return out


def rows_divide(x, scale, invert=False):
    R, n = x.shape
    out = torch.empty_like(x)
    _lib.check(
        _lib.lib.slk_rows_divide(dev.ptr(x), dev.ptr(scale), R, n, 1 if invert else 0, dev.ptr(out), dev.stream_handle())
    )
    return out


def column_miss(W, cb_abi, squared):
    R, n = W.shape
    levels, lo, hi, table = cb_abi
    out = torch.empty(n, dtype=torch.float32, device=W.device)
    _lib.check(
        _lib.lib.slk_column_miss(
            dev.ptr(W), R, n, levels, lo, hi, dev.ptr(table), 1 if squared else 0, dev.ptr(out), dev.stream_handle()
        )
    )
    return out


def row_errors(W, Q, H, want_G=False):
    R, n = W.shape
    ws, ws_bytes = dev.workspace(R, n)
    out = torch.empty(R, dtype=torch.float32, device=W.device)
    G = torch.empty((R, n), dtype=torch.float32, device=W.device) if want_G else None
    _lib.check(
        _lib.lib.slk_row_errors(
            dev.ptr(W), dev.ptr(Q), dev.ptr(H), R, n, dev.ptr(out), dev.ptr(G), dev.ptr(ws), ws_bytes, dev.stream_handle()
        )
    )
    return (out, G) if want_G else out


def quantize_layer(
    W, H, quantizer, scale=None, act_order="diag", damp=0.01, nb_ls_moves=0, min_block_size=32, num_blocks=8,
    factor=None, unscale=True, want_idx=True, want_ls_trace=False, lookahead=True, want_ls_error=False,
):
    """One layer through the whole path, on device tensors.

    W (R, n) float32, H (n, n) float32, scale (R,) float32 or None.  `factor` = (order, U, info)
    re-uses a factor computed elsewhere (another GPU: see sleekit_amd.dist).  Returns a
    LayerResult whose Q is de-scaled when `scale` is given and `unscale` is true
    (sleekit/scaling.py:58-81), else the codebook values in the scaled domain
    (sleekit/obq.py:169-217).  want_ls_trace: res.ls_trace = the local search's moves (slk_local_search).
    lookahead: this layer is alone on the GPU (the default of this single-layer API; sleekit_amd.dist passes False): the
    factorisation looks ahead (see factorize) and the loop's window kernel takes 16-row workgroups (SLK_LOOP_LATENCY).
    """
    assert W.ndim == 2
    assert H.ndim == 2
    assert H.shape[0] == H.shape[1]
    assert H.shape[0] == W.shape[1]
    assert min_block_size >= 1
    cb_abi = require_uniform(quantizer)
    if cb_abi[0] > 256:
        want_idx = False  # (the kernels emit uint8 indices; wider ones come from quantizer.quantize_index on the values)
    mode = order_mode_code(act_order)
    R, n = W.shape
    res = LayerResult()

    need_scaled_copy = scale is not None and (mode in (_lib.ORDER_ERR, _lib.ORDER_SQERR) or nb_ls_moves > 0)
    Ws, loop_scale = (rows_divide(W, scale), None) if need_scaled_copy else (W, scale)

    if factor is None:
        if mode == _lib.ORDER_KEYS:
            miss = order_keys(H, n, damp, act_order)
        else:
            miss = column_miss(Ws, cb_abi, mode == _lib.ORDER_SQERR) if mode >= _lib.ORDER_ERR else None
        factor = factorize(H, n, damp, mode, miss, lookahead=lookahead)
        check_factor = True
    else:
        check_factor = False
    res.order, res.U, res.info = factor

    # without local search the loop's last kernel de-scales on the way out (one pass over Q less)
    fused = scale is not None and unscale and nb_ls_moves == 0 and loop_scale is not None
    res.Q, res.idx, _ = run_loop(Ws, loop_scale, res.order, res.U, cb_abi, min_block_size, num_blocks, want_idx, unscale=fused, latency=lookahead)
    if nb_ls_moves > 0:
        if want_ls_error:  # the rows' errors after the moves, in the scaled domain (res.ls_error)
            res.ls_error = torch.empty(R, dtype=torch.float32, device=W.device)
        res.ls_trace = local_search(Ws, res.Q, H, cb_abi, nb_ls_moves, res.idx, want_trace=want_ls_trace, row_err=res.ls_error)
    if scale is not None and unscale and not fused:
        res.Q = rows_divide(res.Q, scale, invert=True)
    # The factorisation's status word is read back only now, with the loop (and the search) already enqueued behind it: read
    # right after the factorisation, the round trip to the host kept the loop's first launch waiting 35-45 us per layer.  A
    # matrix that is not positive definite still raises here (numpy.linalg.LinAlgError, as np.linalg.cholesky does in
    # sleekit/obq.py:49-50); the loop then ran on a void factor, memory-safe, its results never returned.
    if check_factor:
        dev.note_info(res.info, "compute_hessian_chol")
    return res
