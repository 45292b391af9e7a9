"""`Sleekit`: the GPTQ-style layer adapter of the reference (sleekit/statistics.py:12-199), kept on the GPU.

    st = Sleekit(layer); st.add_batch(x) ...; st.quantize_sleekit_light(3)      # or .quantize(nbits, ...)

What each piece maps to:

* `add_batch` -- the running mean and Hessian of statistics.py:76-87 -- is ONE call of `slk_hessian_accumulate`
  per batch (float32-grade products of bfloat16 pieces on the matrix cores when the feature count is a multiple
  of 128, the float32 MFMA otherwise).  Samples are handed over as ROWS (T, n); the reference holds them as
  columns (n, T): same statistics.  Convolutions are seen through `torch.nn.functional.unfold` like there
  (statistics.py:44-69): pure data movement.
* `quantize` (statistics.py:146-190): scale selection (`sleekit_amd.scaling.compute_scaling`), the device
  pipeline (`engine.quantize_layer`) and the bias correction `bias += ((W - Q) * mean).sum(1)`, all on device
  tensors -- the reference goes through `.numpy()` for every one of them (statistics.py:162-166).
* the three presets (statistics.py:107-144) are rows of `_PRESETS`.
"""

import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _device as dev
from . import _lib
from . import engine
from .codebook import UniformCodebook
from .scaling import compute_scaling

_SUPPORTED = (nn.Linear, nn.Conv1d, nn.Conv2d)

# keyword arguments of Sleekit.quantize behind quantize_<name>(nbits)
_PRESETS = {
    "basic": dict(scaling_mode="mse", order_mode="diag", bias_correction=False, damp=0.01, nb_ls_moves=0),
    "sleekit_light": dict(scaling_mode="diag", order_mode="sqerr", bias_correction=True, damp=0.03, nb_ls_moves=0),
    "sleekit_heavy": dict(scaling_mode="hessian", order_mode="sqerr", bias_correction=True, damp=0.03, nb_ls_moves=100),
}
_PRESET_NOTES = {
    "basic": "plain GPTQ-like settings, none of the improvements (statistics.py:107-118)",
    "sleekit_light": 'the "light" recipe (statistics.py:120-131)',
    "sleekit_heavy": 'the "heavy" recipe, 100 local-search moves (statistics.py:133-144)',
}


def _windows_as_rows(x, layer):
    """Every receptive field of a convolution as one row of length channels * prod(kernel) (statistics.py:44-69)."""
    kernel, dilation, padding, stride = layer.kernel_size, layer.dilation, layer.padding, layer.stride
    if isinstance(layer, nn.Conv1d):
        # a (k, 1) two-dimensional convolution over a trailing axis of length one
        x = x[None] if x.ndim == 2 else x
        x = x[..., None]
        kernel, dilation, padding, stride = (kernel[0], 1), (dilation[0], 1), (padding[0], 0), (stride[0], 1)
    elif x.ndim == 3:
        x = x[None]
    cols = F.unfold(x, kernel, dilation, padding, stride)  # (batch, features, positions)
    return cols.transpose(1, 2).reshape(-1, cols.shape[1])


def _preset_method(name):
    def run(self, nbits):
        return self.quantize(nbits, **_PRESETS[name])

    run.__name__ = "quantize_" + name
    run.__doc__ = f"quantize(nbits) with {_PRESET_NOTES[name]}."
    return run


class Sleekit:
    """Running statistics of one layer's inputs and its quantization; interface of the reference's class."""

    def __init__(self, layer):
        if not isinstance(layer, _SUPPORTED):
            raise ValueError(f"Unsupported layer type {type(layer)}")
        if not layer.weight.is_cuda:
            raise RuntimeError("sleekit_amd.Sleekit accumulates on the GPU: move the layer to the device first")
        self.layer = layer
        features = layer.weight[0].numel()  # columns of the weight seen as an (out, features) matrix
        self.count = 0
        self.mean = torch.zeros(features, dtype=torch.float32, device=self.device)
        self.hessian = torch.zeros((features, features), dtype=torch.float32, device=self.device)

    @property
    def device(self):
        return self.layer.weight.device

    def _prepare_input(self, inp):
        """(T, features) float32 samples of a batch of activations, one per row (statistics.py:37-74 gives the transpose)."""
        x = inp.to(self.device)
        rows = x.reshape(-1, x.shape[-1]) if isinstance(self.layer, nn.Linear) else _windows_as_rows(x, self.layer)
        assert rows.ndim == 2
        return rows.float().contiguous()

    def add_batch(self, inp, out=None):
        """mean <- mean c / (c + T) + sum / (c + T), hessian likewise with X^T X: statistics.py:76-87, one kernel call."""
        samples = self._prepare_input(inp)
        tokens, features = samples.shape
        assert features == self.mean.numel()
        scratch, scratch_bytes = dev.workspace(0, features)
        rc = _lib.lib.slk_hessian_accumulate(dev.ptr(self.hessian), dev.ptr(self.mean), dev.ptr(samples), features, tokens,
                                             int(self.count), dev.ptr(scratch), scratch_bytes, dev.stream_handle())
        _lib.check(rc)
        self.count += tokens

    def export(self, path, npy_format=False):
        """bias / weight / mean / hessian as four .pt (or .npy) files in `path`: what the experiments load (statistics.py:89-105)."""
        os.makedirs(path, exist_ok=True)
        for name in ("bias", "weight", "mean", "hessian"):
            value = getattr(self.layer, name) if name in ("bias", "weight") else getattr(self, name)
            host = value.detach().cpu()
            if npy_format:
                import numpy as np

                np.save(os.path.join(path, f"{name}.npy"), host.numpy())
            else:
                torch.save(host, os.path.join(path, f"{name}.pt"))

    quantize_basic = _preset_method("basic")
    quantize_sleekit_light = _preset_method("sleekit_light")
    quantize_sleekit_heavy = _preset_method("sleekit_heavy")

    def quantize(self, nbits, scaling_mode="mse", order_mode="diag", bias_correction=False, damp=0.01, nb_ls_moves=0,
                 grid_size=100, min_factor=0.05, max_factor=1.0, scale=None):
        """The layer's weight replaced by its `nbits` quantization, in place (statistics.py:146-190).

        bias_correction: quantize against H - mean mean^T and move the expected output shift into the bias.
        `scale` (optional, (out,) float32): this per-row scale instead of the scale search.
        """
        codebook = UniformCodebook(2**nbits, -1, 1)
        weight = self.layer.weight.data.flatten(1).float().contiguous()
        H = self.hessian
        if bias_correction:
            centred = torch.empty_like(H)
            _lib.check(_lib.lib.slk_hessian_strip_mean(dev.ptr(H), dev.ptr(self.mean), H.shape[0], dev.ptr(centred), dev.stream_handle()))
            H = centred
        if scale is None:
            scale = compute_scaling(weight, codebook, H=H, mode=scaling_mode, grid_size=grid_size, min_factor=min_factor,
                                    max_factor=max_factor)
        result = engine.quantize_layer(weight, H, codebook, dev.to_device(scale), order_mode, damp, nb_ls_moves)
        target = self.layer.weight
        target.data = result.Q.reshape(target.shape).to(target.dtype)
        if bias_correction:
            shift = ((weight - result.Q) * self.mean).sum(dim=1)
            self.layer.bias.data += shift.to(self.layer.bias.dtype)
        return result

    def free(self):
        self.layer = self.mean = self.hessian = None
        self.count = 0
