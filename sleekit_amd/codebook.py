"""Drop-in for `sleekit.codebook`: the quantizers of the hot path.

`UniformCodebook` keeps the reference's interface (sleekit/codebook.py:4-95) and runs
every map on the GPU through `slk_codebook_apply`, in float32 with the reference's exact
rounding (true divide, round-half-to-even, separate multiply and add).

`Codebook` (sleekit/codebook.py:98-190, 277-320) is the general, table-driven codebook --
sorted values and the limits between their bins, `np.digitize` semantics -- with the same four
maps on the GPU; every kernel of the path takes either kind.  Its TRAINING helpers (`improve`,
`centroids`, `equiprobable`, `random`, ... and `lloyd_max`, sleekit/codebook.py:192-275, 322-367)
fit a codebook to data on the host in the reference; they are not part of the quantization path
and raise NotImplementedError here rather than silently running something else.
"""

import numpy as np  # noqa: F401  (the reference's star-importers rely on `np` leaking from here)
import torch

from . import _device as dev
from . import _lib


class DeviceCodebook:
    """What every quantizer of the path provides: `_abi()` -> (levels, lo, hi, device table or None),
    and the four maps on the GPU."""

    def _apply(self, data, what):
        x = dev.to_device(data, torch.float32)
        if what == _lib.CB_INDEX:
            # the narrowest unsigned type that holds every index, like the reference (codebook.py:50-54); torch has no
            # arithmetic on uint16 / uint32, so a device tensor gets int16 / int32 storage viewed as the unsigned type
            if len(self) <= 2**8:
                out = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
            elif len(self) <= 2**16:
                what, out = _lib.CB_INDEX16, torch.empty(x.shape, dtype=torch.uint16, device=x.device)
            else:
                what, out = _lib.CB_INDEX32, torch.empty(x.shape, dtype=torch.uint32, device=x.device)
        else:
            out = torch.empty_like(x)
        levels, lo, hi, table = self._abi()
        _lib.check(
            _lib.lib.slk_codebook_apply(
                dev.ptr(x), x.numel(), levels, lo, hi, dev.ptr(table), what, dev.ptr(out), dev.stream_handle()
            )
        )
        return dev.like_input(out, data)

    def quantize_index(self, data):
        """Index of the nearest codebook value (uint8, or uint16 / uint32 for codebooks above 256 / 65536 entries)."""
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


class UniformCodebook(DeviceCodebook):
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

    # (levels, lo, hi, table) as the C ABI wants them
    def _abi(self):
        return self.codebook_size, float(self.min_val), float(self.max_val), None


class Codebook(DeviceCodebook):
    """General codebook: sorted values and the limits between their bins (sleekit/codebook.py:98-190).

    The four maps run on the GPU like the uniform ones (`DeviceCodebook._apply`); the table (values,
    then limits) lives on the device, at most 256 entries.
    """

    def __init__(self, values, limits=None):
        self.values = np.array(values, dtype=np.float32)
        if limits is not None:
            self.thresholds = np.array(limits, dtype=np.float32)
        else:
            self.values.sort()
            self.thresholds = (self.values[:-1] + self.values[1:]) / 2
        self.check()
        self._table = {}

    def clone(self):
        return Codebook(self.values.copy(), self.thresholds.copy())

    def check(self):
        """Consistency checks of the reference (sleekit/codebook.py:119-132)."""
        assert self.values.ndim == 1
        assert self.values.size > 0
        assert np.isfinite(self.values).all()
        assert (self.values[1:] > self.values[:-1]).all()
        assert self.thresholds.ndim == 1
        assert self.thresholds.size == self.values.size - 1
        assert np.isfinite(self.thresholds).all()
        assert (self.thresholds[1:] > self.thresholds[:-1]).all()
        assert (self.thresholds >= self.values[:-1]).all()
        assert (self.thresholds <= self.values[1:]).all()

    def __len__(self):
        return len(self.values)

    def min(self):
        return self.values[0]

    def max(self):
        return self.values[-1]

    def _abi(self):
        if len(self) < 2 or len(self) > 256:
            raise NotImplementedError("general codebooks on the device hold 2 to 256 entries")
        device = dev.require_gpu()
        key = str(device)
        if key not in self._table:
            both = np.concatenate([self.values, self.thresholds]).astype(np.float32)
            self._table[key] = torch.as_tensor(both, device=device)
        return len(self), float(self.values[0]), float(self.values[-1]), self._table[key]

    @staticmethod
    def uniform(codebook_size, min_val, max_val):
        """A uniform codebook in table form (sleekit/codebook.py:288-294)."""
        assert min_val <= max_val
        return Codebook(np.linspace(min_val, max_val, codebook_size))

    @staticmethod
    def nf4():
        """The NormalFloat4 datatype (sleekit/codebook.py:296-320)."""
        return Codebook(
            [
                -1.0, -0.6961928009986877, -0.5250730514526367, -0.39491748809814453, -0.28444138169288635,
                -0.18477343022823334, -0.09105003625154495, 0.0, 0.07958029955625534, 0.16093020141124725,
                0.24611230194568634, 0.33791524171829224, 0.44070982933044434, 0.5626170039176941,
                0.7229568362236023, 1.0,
            ]
        )

    # -- codebook training (host-side in the reference; not part of the quantization path) --
    def _training(self, name):
        raise NotImplementedError(
            f"Codebook.{name} fits a codebook to data on the host in the reference (sleekit/codebook.py:192-275); "
            "sleekit_amd runs the quantization path only and has no CPU fallback"
        )

    def probabilities(self, data):
        self._training("probabilities")

    def entropy(self, data):
        self._training("entropy")

    def mse(self, data):
        self._training("mse")

    def centroids(self, data):
        self._training("centroids")

    def remove_unused(self, data):
        self._training("remove_unused")

    def improve(self, data, lagrange_mult=0.0):
        self._training("improve")

    @staticmethod
    def random(data, codebook_size):
        raise NotImplementedError("Codebook.random (sleekit/codebook.py:277-286) is codebook training: not on the path")

    @staticmethod
    def equiprobable(data, codebook_size):
        raise NotImplementedError("Codebook.equiprobable (sleekit/codebook.py:322-335) is codebook training: not on the path")


def lloyd_max(*args, **kwargs):
    raise NotImplementedError("lloyd_max (sleekit/codebook.py:338-367) is codebook training: outside the quantization path (DESIGN.md)")
