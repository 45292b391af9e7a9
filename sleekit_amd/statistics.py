"""Drop-in for `sleekit.statistics.Sleekit`: running layer statistics on the GPU.

`add_batch` (sleekit/statistics.py:76-87) for nn.Linear runs on float32 MFMA through
`slk_hessian_accumulate`; mean and Hessian live on the layer's device.  `quantize`
(statistics.py:146-190) drives the device pipeline with a caller-supplied scale; the
scale searches and the Conv1d/Conv2d unfold are "next" rows of SURVEY.md 8(f).
"""

import torch
import torch.nn as nn

from . import _device as dev
from . import _lib
from . import engine
from .codebook import UniformCodebook


class Sleekit:
    """Statistics of a layer, with the GPTQ-compatible interface of the reference."""

    def __init__(self, layer):
        self.layer = layer
        if not isinstance(self.layer, (nn.Linear, nn.Conv1d, nn.Conv2d)):
            raise ValueError(f"Unsupported layer type {type(self.layer)}")
        if not isinstance(self.layer, nn.Linear):
            raise NotImplementedError("Conv1d/Conv2d unfolding is a 'next' row (SURVEY.md 8f); nn.Linear is on the path")
        if not layer.weight.is_cuda:
            raise RuntimeError("sleekit_amd.Sleekit accumulates on the GPU: move the layer to the device first")
        n = layer.weight.shape[1]
        self.mean = torch.zeros(n, dtype=torch.float32, device=self.device)
        self.hessian = torch.zeros((n, n), dtype=torch.float32, device=self.device)
        self.count = 0

    @property
    def device(self):
        return self.layer.weight.device

    def _prepare_input(self, inp):
        """(…, in) activations -> (T, in) float32 tokens, row-major (statistics.py:41-43, 73).

        The reference transposes to (in, T); the kernel reads tokens as rows instead.
        """
        inp = inp.reshape((-1, inp.shape[-1]))
        return inp.to(device=self.device, dtype=torch.float32).contiguous()

    def add_batch(self, inp, out=None):
        X = self._prepare_input(inp)
        T, n = X.shape
        assert n == self.mean.shape[0]
        _lib.check(
            _lib.lib.slk_hessian_accumulate(
                dev.ptr(self.hessian), dev.ptr(self.mean), dev.ptr(X), n, T, int(self.count), dev.stream_handle()
            )
        )
        self.count += T

    def quantize(self, nbits, scale, order_mode="diag", bias_correction=False, damp=0.01, nb_ls_moves=0):
        """Quantize the layer in place with a given per-row scale (statistics.py:146-190).

        `scale`: (out,) float32 tensor.  The reference derives it from `compute_scaling`
        (a 'next' row); everything after that point is the accelerated path.
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
