"""Oracle: per-row scaling around the quantization loop.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates `sleekit/scaling.py`:
    divide_rows          scaling.py:11-25   (apply_scaling, axis 0 of a 2-D array)
    norm_scale           scaling.py:35-41   (compute_norm_scaling)
    no_clip_scale        scaling.py:44-55   (compute_non_saturating_scaling)
    quantize_scaled      scaling.py:58-81   (quantize_with_scaling)
    grid_error           scaling.py:84-95   (_compute_mse)
    best_grid_scale      scaling.py:98-134  (compute_min_mse_scaling)
    best_obq_scale       scaling.py:137-190 (compute_obq_scaling)
    pick_scale           scaling.py:193-238 (compute_scaling)
"""

import numpy as np

from . import obq_ref


def _along(data, scale, axis):
    assert scale.ndim == 1
    shape = [1] * data.ndim
    shape[axis] = -1
    return scale.reshape(shape)


def divide_rows(data, scale, axis=0):
    return data / _along(data, scale, axis)


def norm_scale(data, axis=0):
    rest = tuple(i for i in range(data.ndim) if i != axis)
    return np.sqrt(np.maximum(np.square(data).mean(axis=rest), 1.0e-16))


def no_clip_scale(data, grid, axis=0):
    if grid.min() >= 0 or grid.max() <= 0:
        raise RuntimeError("Codebook should have both negative and positive values.")
    rest = tuple(i for i in range(data.ndim) if i != axis)
    lo, hi = data.min(axis=rest), data.max(axis=rest)
    scale = np.maximum(hi / grid.max(), lo / grid.min())
    return np.maximum(scale, np.float32(1.0e-16))


def quantize_scaled(data, scale, grid, H=None, order_mode="diag", damp=0.01, ls_moves=0, ties="numpy", ls_records=None):
    """Divide rows by `scale`, quantize (GPTQ loop when H is given), undo the scale.

    Note the un-scaling is a division by the float32 reciprocal (scaling.py:80),
    i.e. two IEEE divides, not a multiplication.
    """
    assert data.ndim == 2 and scale.ndim == 1 and data.shape[0] == scale.size
    q = divide_rows(data, scale, 0)
    if H is not None:
        q = obq_ref.quantize_layer(q, H, grid, order_mode=order_mode, damp=damp, ls_moves=ls_moves, ties=ties, ls_records=ls_records)
    else:
        q = grid(q)
    return divide_rows(q, 1 / scale, 0)


def grid_error(H, D):
    if H is None:
        return np.square(D).sum(axis=1)
    if H.ndim == 1:
        assert D.shape[1] == H.shape[0]
        return (np.expand_dims(H, 0) * np.square(D)).sum(axis=1)
    assert H.ndim == 2 and D.shape[1] == H.shape[0] == H.shape[1]
    return ((D @ H) * D).sum(axis=-1)


def _first_best(base, factors, row_errors):
    """The grid search shared by scaling.py:98-134 and 160-190: walk the factors in order, keep for every row the FIRST
    factor whose error is strictly smaller than everything before it (float32 bookkeeping, both start at +inf)."""
    chosen = np.full(base.size, np.inf, dtype=np.float32)
    lowest = np.full(base.size, np.inf, dtype=np.float32)
    for f in factors:
        e = row_errors(f * base)
        improved = e < lowest
        lowest[improved] = e[improved]
        chosen[improved] = f
    return base * chosen


def _rows_first(data, axis):
    others = tuple(i for i in range(data.ndim) if i != axis)
    return np.transpose(data, [axis, *others])


def best_grid_scale(data, grid, axis=0, H=None, min_factor=0.05, max_factor=1.0, grid_size=100):
    flat = _rows_first(data, axis)
    factors = np.linspace(min_factor, max_factor, grid_size, dtype=np.float32)
    return _first_best(no_clip_scale(flat, grid, 0), factors, lambda sc: grid_error(H, quantize_scaled(flat, sc, grid) - flat))


def best_obq_scale(
    data, grid, axis, H, damp=0.01, order_mode="diag", min_factor=0.05, max_factor=1.0, grid_size=100
):
    W = _rows_first(data, axis)
    base = no_clip_scale(W, grid, 0)
    n = H.shape[0]
    H_damped = H + damp * H.diagonal().mean() * np.eye(n)
    order = obq_ref.column_order(divide_rows(W, base, 0), H_damped, grid, order_mode)
    # everything in processing order from here on, the error included (scaling.py:170-172)
    W = W[:, order]
    H = H[order][:, order]
    U = obq_ref.inverse_factor_upper(H_damped[order][:, order])
    ops = obq_ref.block_schedule(n, 32, 8)

    def loop_error(sc):
        Q = divide_rows(W, sc, 0)
        obq_ref.run_schedule(Q, np.zeros_like(W), U, grid, ops)
        return grid_error(H, divide_rows(Q, 1 / sc, 0) - W)

    return _first_best(base, np.linspace(min_factor, max_factor, grid_size, dtype=np.float32), loop_error)


def pick_scale(data, grid, H, mode="mse", axis=0, min_factor=0.05, max_factor=1.0, grid_size=100):
    """scaling.py:193-238: "max" | "norm" | "obq" | "mse" | "hessian[N]" | "diag[N]" (N: per cent of the mean diagonal added)."""
    search = dict(grid_size=grid_size, min_factor=min_factor, max_factor=max_factor)
    if mode in ("max", "norm"):
        return no_clip_scale(data, grid, axis) if mode == "max" else norm_scale(data, axis)
    if mode == "obq":
        return best_obq_scale(data, grid, axis, H=H, **search)
    for family in ("hessian", "diag"):
        if mode.startswith(family):
            extra = mode[len(family):]
            if family == "diag":
                H = H.diagonal()
                if extra:
                    H = H + 0.01 * float(extra) * H.mean()
            elif extra:
                H = H + 0.01 * float(extra) * H.diagonal().mean() * np.eye(H.shape[0])
            return best_grid_scale(data, grid, axis, H=H, **search)
    if mode != "mse":
        raise RuntimeError(f"Unknown scaling mode {mode}")
    return best_grid_scale(data, grid, axis, H=None, **search)
