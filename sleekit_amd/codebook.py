"""Drop-in for `sleekit.codebook` (hot-path slice: the uniform codebook).

`UniformCodebook` keeps the reference's interface (sleekit/codebook.py:4-95) and runs
every map on the GPU through `slk_codebook_apply`, in float32 with the reference's exact
rounding (true divide, round-half-to-even, separate multiply and add).

The general `Codebook` / `lloyd_max` (sleekit/codebook.py:98-367) are outside the
accelerated path (no experiment or BASELINE config uses them); the names exist and
raise, rather than silently running something else.
"""

import numpy as np  # noqa: F401  (the reference's star-importers rely on `np` leaking from here)
import torch

from . import _device as dev
from . import _lib


class UniformCodebook:
    """Evenly spaced codebook on [min_val, max_val]; the only quantizer on the accelerated path."""

    def __init__(self, codebook_size, min_val, max_val):
        self.codebook_size = int(codebook_size)
        self.min_val = min_val
        self.max_val = max_val
        assert self.min_val < self.max_val
        assert self.codebook_size >= 2

    def __len__(self):
        return self.codebook_size

    @property
    def values(self):
        return np.linspace(self.min_val, self.max_val, self.codebook_size)

    def min(self):
        return self.min_val

    def max(self):
        return self.max_val

    @property
    def scale(self):
        return (self.max_val - self.min_val) / (self.codebook_size - 1)

    @property
    def zero(self):
        return self.min_val

    # (levels, lo, hi) as the C ABI wants them
    def _abi(self):
        return self.codebook_size, float(self.min_val), float(self.max_val)

    def _apply(self, data, what):
        x = dev.to_device(data, torch.float32)
        if what == _lib.CB_INDEX:
            if self.codebook_size > 256:
                raise NotImplementedError("device indices are uint8: codebooks above 256 entries are not on the path")
            out = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
        else:
            out = torch.empty_like(x)
        levels, lo, hi = self._abi()
        _lib.check(
            _lib.lib.slk_codebook_apply(dev.ptr(x), x.numel(), levels, lo, hi, what, dev.ptr(out), dev.stream_handle())
        )
        return dev.like_input(out, data)

    def quantize_index(self, data):
        """Index of the nearest codebook value (uint8)."""
        return self._apply(data, _lib.CB_INDEX)

    def quantize_value(self, data):
        """Nearest codebook value."""
        return self._apply(data, _lib.CB_VALUE)

    def quantize_up(self, data):
        """The codebook value one step above the nearest, saturating at the top."""
        return self._apply(data, _lib.CB_UP)

    def quantize_down(self, data):
        """The codebook value one step below the nearest, saturating at the bottom."""
        return self._apply(data, _lib.CB_DOWN)

    def __call__(self, data):
        return self.quantize_value(data)


class Codebook:
    """Non-uniform codebook (sleekit/codebook.py:98-335): not on the accelerated path."""

    def __init__(self, *args, **kwargs):
        raise NotImplementedError(
            "sleekit_amd accelerates the UniformCodebook path only; non-uniform codebooks are out of scope (DESIGN.md)"
        )

    @staticmethod
    def uniform(codebook_size, min_val, max_val):
        return UniformCodebook(codebook_size, min_val, max_val)


def lloyd_max(*args, **kwargs):
    raise NotImplementedError("lloyd_max (sleekit/codebook.py:338-367) is outside the accelerated path (DESIGN.md)")
