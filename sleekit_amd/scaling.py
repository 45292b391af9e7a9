"""Drop-in for `sleekit.scaling`: per-row scaling around the quantization loop, on the GPU.

Same names, arguments and defaults as the reference module (sleekit/scaling.py).  The kernels
scale the rows of a 2-D matrix (axis 0: the layout every caller uses -- `Sleekit.quantize` flattens conv
weights first, statistics.py:166); any other axis or rank is brought to that form by a device-side
transpose and back (a cold path).  Row sums follow NumPy's pairwise order for axis 0 of a 2-D array; for
other axes NumPy reduces in another order, so `compute_norm_scaling` then agrees to float32 rounding, not
bit for bit (minima and maxima are exact either way).

Known deviations, all confined to the scale SEARCH helpers (the quantization path itself is
bit-exact): Hessians are taken in float32 (the reference's `hessianN` / float64-diagonal variants
promote to float64), and the searches whose errors come out of a Hessian product (`hessian*`, `obq`)
pick among grid points whose errors are GEMM results -- summed in this library's order, not the BLAS's --
so a row whose two best grid points tie to within that rounding may take the other one
(tests/test_gpu_parity.py::test_scale_selection proves each such row against the reference's own errors).
"""

import numpy as np  # noqa: F401  (star-importers of the reference rely on `np` leaking from here)
import torch

from . import _device as dev
from . import _lib
from . import engine
from .obq import _quantize_opt_block, compute_hessian_chol, compute_hessian_order, quantize_opt  # noqa: F401


def _as_rows(data, axis):
    """`data` on the device as a contiguous 2-D matrix whose rows are the slices along `axis` (scaling.py:11-18:
    a scale has one entry per index of that axis)."""
    x = dev.to_device(data)
    assert -x.ndim <= axis < x.ndim
    axis %= x.ndim
    if x.ndim == 2 and axis == 0:
        return x
    return x.movedim(axis, 0).reshape(x.shape[axis], -1).contiguous()


def _from_rows(rows, shape, axis):
    """Inverse of _as_rows: back to an array of `shape` with the rows along `axis`."""
    axis %= len(shape)
    if len(shape) == 2 and axis == 0:
        return rows
    moved = (shape[axis],) + tuple(d for i, d in enumerate(shape) if i != axis)
    return rows.view(moved).movedim(0, axis).contiguous()


def _need_matrix(data, axis):
    """The GPTQ-style functions take a 2-D weight matrix (rows = output channels along `axis`)."""
    assert data.ndim == 2
    return _as_rows(data, axis)


def apply_scaling(data, scale, axis=0):
    """data / scale broadcast along `axis` (sleekit/scaling.py:21-25)."""
    assert scale.ndim == 1
    rows = _as_rows(data, axis)
    assert rows.shape[0] == scale.shape[0]
    out = engine.rows_divide(rows, dev.to_device(scale))
    return dev.like_input(_from_rows(out, tuple(data.shape), axis), data)


def apply_scaling_in_place(data, scale, axis=0):
    """In-place variant (sleekit/scaling.py:28-32)."""
    assert scale.ndim == 1
    rows = _as_rows(data, axis)
    assert rows.shape[0] == scale.shape[0]
    out = _from_rows(engine.rows_divide(rows, dev.to_device(scale)), tuple(data.shape), axis)
    if dev.is_device_tensor(data):
        data.copy_(out)
    else:
        data[...] = dev.like_input(out, data)


def compute_norm_scaling(data, axis=0):
    """Scale that brings the mean square of every slice along `axis` to 1 (sleekit/scaling.py:35-41)."""
    W = _as_rows(data, axis)
    out = torch.empty(W.shape[0], dtype=torch.float32, device=W.device)
    _lib.check(_lib.lib.slk_scale_norm(dev.ptr(W), W.shape[0], W.shape[1], dev.ptr(out), dev.stream_handle()))
    return dev.like_input(out, data)


def _no_clip_scale(W, codebook):
    if codebook.min() >= 0 or codebook.max() <= 0:
        raise RuntimeError("Codebook should have both negative and positive values.")
    out = torch.empty(W.shape[0], dtype=torch.float32, device=W.device)
    _lib.check(
        _lib.lib.slk_scale_minmax(
            dev.ptr(W), W.shape[0], W.shape[1], float(codebook.min()), float(codebook.max()), dev.ptr(out), dev.stream_handle()
        )
    )
    return out


def compute_non_saturating_scaling(data, codebook, axis=0):
    """Largest-magnitude scale with no saturation (sleekit/scaling.py:44-55)."""
    return dev.like_input(_no_clip_scale(_as_rows(data, axis), codebook), data)


def quantize_with_scaling(data, scale, quantizer, H=None, act_order="diag", damp=0.01, nb_ls_moves=0):
    """Quantize the weights after applying a per-row scaling factor (sleekit/scaling.py:58-81).

    Returns the de-quantized weights (float32) in the original domain.
    """
    assert data.ndim == 2
    assert scale.ndim == 1
    assert data.shape[0] == scale.shape[0]
    Wd, sd = dev.to_device(data), dev.to_device(scale)
    if H is None:
        engine.require_uniform(quantizer)
        q = quantizer.quantize_value(engine.rows_divide(Wd, sd))
        return dev.like_input(engine.rows_divide(q, sd, invert=True), data)
    res = engine.quantize_layer(Wd, dev.to_device(H), quantizer, sd, act_order, damp, nb_ls_moves, want_idx=False)
    return dev.like_input(res.Q, data)


def _times(a, b=None, c=0.0):
    out = torch.empty_like(a)
    _lib.check(_lib.lib.slk_scale_times(dev.ptr(a), dev.ptr(b), float(c), a.shape[0], dev.ptr(out), dev.stream_handle()))
    return out


_STACK_BYTES = 1 << 30  # budget for one stacked copy of W in the grid searches below


def _search_stacked(W, base, factors, H, quantize_rows):
    """Grid search whose row errors come out of a Hessian product (full-Hessian and OBQ-aware variants,
    scaling.py:98-134 and 160-190), with the grid points stacked by ROWS: rows never interact and W, H (and the
    factor U) are shared, so a chunk of G grid points is ONE matrix of G R rows with scales f_g * base -- one call
    of `quantize_rows(W_stack, scales)` (-> de-quantized rows) and one layer-error product per chunk instead of one
    per grid point, nothing but kernel launches from the host.  The first minimum wins, like the reference's `<`."""
    R, n = W.shape
    s = dev.stream_handle()
    best_err = torch.empty(R, dtype=torch.float32, device=W.device)
    best_f = torch.empty(R, dtype=torch.float32, device=W.device)
    _lib.check(_lib.lib.slk_search_step(None, 0.0, R, dev.ptr(best_err), dev.ptr(best_f), 1, s))
    per = int(max(1, min(len(factors), _STACK_BYTES // (4 * R * n), (1 << 30) // max(R, 1))))
    fac = torch.from_numpy(np.ascontiguousarray(factors, dtype=np.float32)).to(W.device)
    for c0 in range(0, len(factors), per):
        f = fac[c0:c0 + per]
        G = f.numel()
        if G < per and c0 > 0:
            # a short last chunk is padded to the size of the others (its last point repeated, results unused): the layer
            # error picks its K split from the stack's tile count, and every grid point of a row must be summed in the
            # same order for the first-best `<` below to rank like errors
            f = torch.cat([f, f[-1:].expand(per - G)])
        scales = (f[:, None] * base[None, :]).reshape(-1)  # float32 products, like `s * initial_scale`
        Wst = W.unsqueeze(0).expand(f.numel(), R, n).reshape(f.numel() * R, n)
        err = engine.row_errors(quantize_rows(Wst, scales), Wst, H)
        for g in range(G):
            _lib.check(_lib.lib.slk_search_step(err[g * R:(g + 1) * R].data_ptr(), float(factors[c0 + g]), R, dev.ptr(best_err),
                                                dev.ptr(best_f), 0, s))
    return _times(base, best_f)


def compute_min_mse_scaling(data, codebook, axis=0, H=None, min_factor=0.05, max_factor=1.0, grid_size=100):
    """Scale minimising the (Hessian-weighted) squared error of round-to-nearest (sleekit/scaling.py:98-134)."""
    cb_abi = engine.require_uniform(codebook)
    W = _as_rows(data, axis) if H is None else _need_matrix(data, axis)
    R, n = W.shape
    base = _no_clip_scale(W, codebook)
    factors = np.linspace(min_factor, max_factor, grid_size, dtype=np.float32)
    if H is None or H.ndim == 1:
        hd = None if H is None else dev.to_device(H)
        if hd is not None:
            assert hd.shape[0] == n
        fac = torch.from_numpy(factors).to(W.device)
        out = torch.empty(R, dtype=torch.float32, device=W.device)
        levels, lo, hi, table = cb_abi
        _lib.check(
            _lib.lib.slk_scale_search(
                dev.ptr(W), dev.ptr(base), dev.ptr(fac), len(factors), dev.ptr(hd), R, n, levels, lo, hi, dev.ptr(table), dev.ptr(out),
                dev.stream_handle(),
            )
        )
        return dev.like_input(out, data)
    assert H.ndim == 2 and H.shape[0] == H.shape[1] == n
    Hd = dev.to_device(H)

    def rtn_rows(Wst, scales):
        q = codebook.quantize_value(engine.rows_divide(Wst, scales))
        return engine.rows_divide(q, scales, invert=True)

    return dev.like_input(_search_stacked(W, base, factors, Hd, rtn_rows), data)


def compute_obq_scaling(data, codebook, axis, H, damp=0.01, act_order="diag", min_factor=0.05, max_factor=1.0, grid_size=100):
    """Scale minimising the error AFTER the GPTQ loop (sleekit/scaling.py:137-190): one factor, 100 loops."""
    cb_abi = engine.require_uniform(codebook)
    W, Hd = _need_matrix(data, axis), dev.to_device(H)
    R, n = W.shape
    base = _no_clip_scale(W, codebook)
    mode = engine.order_mode_code(act_order)
    if mode == _lib.ORDER_KEYS:
        miss = engine.inverse_diag_keys(Hd, n, damp, engine._INVERSE_ORDERS[act_order])
    elif mode >= _lib.ORDER_ERR:
        miss = engine.column_miss(engine.rows_divide(W, base), cb_abi, mode == _lib.ORDER_SQERR)
    else:
        miss = None
    order, U, info = engine.factorize(Hd, n, damp, mode, miss)
    dev.note_info(info, "compute_hessian_chol")
    factors = np.linspace(min_factor, max_factor, grid_size, dtype=np.float32)
    # like the reference (scaling.py:170-172): W and H go into the processing order ONCE, the loop then runs in place
    # (identity order) and the errors are evaluated on the permuted pair
    Wp = W.index_select(1, order)
    Hp = Hd.index_select(0, order).index_select(1, order)

    def loop_rows(Wst, scales):
        return engine.run_loop(Wst, scales, None, U, cb_abi, 32, 8, want_idx=False, unscale=True)[0]

    return dev.like_input(_search_stacked(Wp, base, factors, Hp, loop_rows), data)


def compute_scaling(data, codebook, H, mode="mse", axis=0, min_factor=0.05, max_factor=1.0, grid_size=100):
    """Mode dispatcher (sleekit/scaling.py:193-238)."""
    if mode == "max":
        return compute_non_saturating_scaling(data, codebook, axis)
    if mode == "norm":
        return compute_norm_scaling(data, axis)
    kw = dict(grid_size=grid_size, min_factor=min_factor, max_factor=max_factor)
    if mode == "obq":
        return compute_obq_scaling(data, codebook, axis, H=H, **kw)
    if mode == "mse":
        H = None
    elif mode.startswith("hessian"):
        if len(mode) > 7:
            Hd = dev.to_device(H).clone()
            Hd.diagonal().add_(np.float32(0.01 * float(mode[7:])) * Hd.diagonal().mean())
            H = Hd if dev.is_device_tensor(H) else dev.like_input(Hd, H)
    elif mode.startswith("diag"):
        Hd = dev.to_device(H).diagonal().contiguous()
        if len(mode) > 4:
            Hd = Hd + np.float32(0.01 * float(mode[4:])) * Hd.mean()
        H = Hd if dev.is_device_tensor(H) else dev.like_input(Hd, H)
    else:
        raise RuntimeError(f"Unknown scaling mode {mode}")
    return compute_min_mse_scaling(data, codebook, axis, H=H, **kw)
