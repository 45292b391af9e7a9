"""Drop-in for `sleekit.scaling`: per-row scaling around the quantization loop.

`apply_scaling*` and `quantize_with_scaling` (sleekit/scaling.py:11-33, 58-81) wrap the hot
path and run on the GPU.  The scale *searches* (`compute_min_mse_scaling`,
`compute_obq_scaling`, `compute_scaling` and the two closed-form scales; scaling.py:35-55,
84-238) are the callers' pre-step, listed as "next" in SURVEY.md section 8(f): the names
exist and raise until their kernels land.
"""

import numpy as np  # noqa: F401  (star-importers of the reference rely on `np` leaking from here)
import torch

from . import _device as dev
from . import engine
from .obq import _quantize_opt_block, compute_hessian_chol, compute_hessian_order, quantize_opt  # noqa: F401


def _check_axis0(data, scale, axis):
    assert scale.ndim == 1
    if axis != 0 or data.ndim != 2:
        raise NotImplementedError("the device path scales axis 0 of a 2-D weight matrix (what the hot path uses)")
    assert data.shape[0] == scale.shape[0]


def apply_scaling(data, scale, axis=0):
    """data / scale broadcast along `axis` (sleekit/scaling.py:21-25)."""
    _check_axis0(data, scale, axis)
    out = engine.rows_divide(dev.to_device(data), dev.to_device(scale))
    return dev.like_input(out, data)


def apply_scaling_in_place(data, scale, axis=0):
    """In-place variant (sleekit/scaling.py:28-32)."""
    _check_axis0(data, scale, axis)
    out = engine.rows_divide(dev.to_device(data), dev.to_device(scale))
    if dev.is_device_tensor(data):
        data.copy_(out)
    else:
        data[...] = dev.like_input(out, data)


def quantize_with_scaling(data, scale, quantizer, H=None, act_order="diag", damp=0.01, nb_ls_moves=0):
    """Quantize the weights after applying a per-row scaling factor (sleekit/scaling.py:58-81).

    Returns the de-quantized weights (float32) in the original domain.
    """
    assert data.ndim == 2
    assert scale.ndim == 1
    assert data.shape[0] == scale.shape[0]
    Wd, sd = dev.to_device(data), dev.to_device(scale)
    if H is None:
        cb_abi = engine.require_uniform(quantizer)
        q = quantizer.quantize_value(engine.rows_divide(Wd, sd))
        del cb_abi
        return dev.like_input(engine.rows_divide(q, sd, invert=True), data)
    res = engine.quantize_layer(Wd, dev.to_device(H), quantizer, sd, act_order, damp, nb_ls_moves, want_idx=False)
    return dev.like_input(res.Q, data)


def _next(name, where):
    def stub(*args, **kwargs):
        raise NotImplementedError(
            f"{name} ({where}) is the callers' scale search, scheduled after the hot path (SURVEY.md 8f); "
            "sleekit_amd has no CPU fallback for it"
        )

    stub.__name__ = name
    return stub


compute_norm_scaling = _next("compute_norm_scaling", "sleekit/scaling.py:35-41")
compute_non_saturating_scaling = _next("compute_non_saturating_scaling", "sleekit/scaling.py:44-55")
compute_min_mse_scaling = _next("compute_min_mse_scaling", "sleekit/scaling.py:98-134")
compute_obq_scaling = _next("compute_obq_scaling", "sleekit/scaling.py:137-190")
compute_scaling = _next("compute_scaling", "sleekit/scaling.py:193-238")
