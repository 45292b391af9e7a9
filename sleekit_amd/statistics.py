"""Drop-in for `sleekit.statistics.Sleekit`: running layer statistics and layer quantization on the GPU.

Interface of the reference class (sleekit/statistics.py:12-199, "API compatible with GPTQ"):
`Sleekit(layer).add_batch(inp)`, `.quantize(nbits, ...)`, the three presets, `.export(path)`, `.free()`.

* `add_batch` (statistics.py:76-87) runs on the matrix cores through `slk_hessian_accumulate` (float32-grade
  products of bfloat16 pieces when the feature count is a multiple of 128, float32 MFMA otherwise); the mean
  and the Hessian live on the layer's device and never visit the host.  Conv1d / Conv2d inputs are
  unfolded with `torch.nn.functional.unfold` exactly as the reference does (statistics.py:44-69) --
  data movement, not arithmetic -- and then take the same kernel.
* `quantize` (statistics.py:146-190) = scale selection + the device pipeline + bias correction, with
  no `.numpy()` round trip (the reference moves everything to the host at statistics.py:162-166).
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


class Sleekit:
    """Statistics of a layer, with the GPTQ-compatible interface of the reference."""

    def __init__(self, layer):
        self.layer = layer
        if not isinstance(self.layer, (nn.Linear, nn.Conv1d, nn.Conv2d)):
            raise ValueError(f"Unsupported layer type {type(self.layer)}")
        if not layer.weight.is_cuda:
            raise RuntimeError("sleekit_amd.Sleekit accumulates on the GPU: move the layer to the device first")
        weight = layer.weight
        if isinstance(self.layer, (nn.Conv1d, nn.Conv2d)):
            weight = weight.flatten(1)
        n = weight.shape[1]
        self.mean = torch.zeros(n, dtype=torch.float32, device=self.device)
        self.hessian = torch.zeros((n, n), dtype=torch.float32, device=self.device)
        self.count = 0

    @property
    def device(self):
        return self.layer.weight.device

    def _prepare_input(self, inp):
        """Activations -> (T, n) float32 samples, one per ROW (statistics.py:37-74).

        The reference produces (n, T); the kernel reads samples as rows, so the transposes differ
        but the statistics are the same.
        """
        inp = inp.to(self.device)
        if isinstance(self.layer, nn.Linear):
            inp = inp.reshape((-1, inp.shape[-1]))
        elif isinstance(self.layer, nn.Conv2d):
            if inp.ndim == 3:
                inp = torch.unsqueeze(inp, 0)
            inp = F.unfold(inp, self.layer.kernel_size, self.layer.dilation, self.layer.padding, self.layer.stride)
            inp = inp.permute([0, 2, 1]).flatten(0, 1)  # (batch * positions, channels * k * k)
        else:  # Conv1d: unfold as a (k, 1) 2-D convolution, like the reference
            if inp.ndim == 2:
                inp = torch.unsqueeze(inp, 0)
            inp = torch.unsqueeze(inp, -1)
            inp = F.unfold(
                inp, (self.layer.kernel_size[0], 1), (self.layer.dilation[0], 1), (self.layer.padding[0], 0),
                (self.layer.stride[0], 1),
            )
            inp = inp.permute([0, 2, 1]).flatten(0, 1)
        assert inp.ndim == 2
        return inp.float().contiguous()

    def add_batch(self, inp, out=None):
        """Fold a batch into the running mean and Hessian (statistics.py:76-87)."""
        X = self._prepare_input(inp)
        T, n = X.shape
        assert n == self.mean.shape[0]
        ws, ws_bytes = dev.workspace(0, n)
        _lib.check(
            _lib.lib.slk_hessian_accumulate(
                dev.ptr(self.hessian), dev.ptr(self.mean), dev.ptr(X), n, T, int(self.count), dev.ptr(ws), ws_bytes,
                dev.stream_handle(),
            )
        )
        self.count += T

    def export(self, path, npy_format=False):
        """Dump bias / weight / mean / hessian as .pt or .npy, the files the experiments read (statistics.py:89-105)."""
        os.makedirs(path, exist_ok=True)
        items = dict(bias=self.layer.bias, weight=self.layer.weight, mean=self.mean, hessian=self.hessian)
        for name, t in items.items():
            if npy_format:
                import numpy as np

                np.save(os.path.join(path, name + ".npy"), t.detach().cpu().numpy())
            else:
                torch.save(t.detach().cpu(), os.path.join(path, name + ".pt"))

    def quantize_basic(self, nbits):
        """A typical quantization method, without the improvements (statistics.py:107-118)."""
        return self.quantize(nbits, scaling_mode="mse", order_mode="diag", bias_correction=False, damp=0.01, nb_ls_moves=0)

    def quantize_sleekit_light(self, nbits):
        """Sleekit "light" (statistics.py:120-131)."""
        return self.quantize(nbits, scaling_mode="diag", order_mode="sqerr", bias_correction=True, damp=0.03, nb_ls_moves=0)

    def quantize_sleekit_heavy(self, nbits):
        """Sleekit "heavy" (statistics.py:133-144)."""
        return self.quantize(nbits, scaling_mode="hessian", order_mode="sqerr", bias_correction=True, damp=0.03, nb_ls_moves=100)

    def quantize(self, nbits, scaling_mode="mse", order_mode="diag", bias_correction=False, damp=0.01, nb_ls_moves=0,
                 grid_size=100, min_factor=0.05, max_factor=1.0, scale=None):
        """Quantize the layer in place to `nbits` (statistics.py:146-190).

        `scale` (optional, (out,) float32): skip the scale search and use this per-row scale.
        """
        cb = UniformCodebook(2**nbits, -1, 1)
        H = self.hessian
        if bias_correction:
            Hc = torch.empty_like(H)
            _lib.check(
                _lib.lib.slk_hessian_strip_mean(dev.ptr(H), dev.ptr(self.mean), H.shape[0], dev.ptr(Hc), dev.stream_handle())
            )
            H = Hc
        weight = self.layer.weight.data.flatten(1).float().contiguous()
        if scale is None:
            scale = compute_scaling(weight, cb, H=H, mode=scaling_mode, grid_size=grid_size, min_factor=min_factor,
                                    max_factor=max_factor)
        res = engine.quantize_layer(weight, H, cb, dev.to_device(scale), order_mode, damp, nb_ls_moves)
        self.layer.weight.data = res.Q.reshape(self.layer.weight.shape).to(self.layer.weight.dtype)
        if bias_correction:
            delta = ((weight - res.Q) * self.mean).sum(dim=1)
            self.layer.bias.data += delta.to(self.layer.bias.dtype)
        return res

    def free(self):
        self.layer = None
        self.mean = None
        self.hessian = None
        self.count = 0
