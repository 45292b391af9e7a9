"""Oracle: evenly spaced codebook ("uniform grid").

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates `UniformCodebook` of the reference (sleekit/codebook.py:4-95).
Every method performs the same NumPy element-wise steps in the same order so
that float32 inputs are processed entirely in float32 (NEP 50: the Python
scalars `zero`/`scale` are weak and adopt the array dtype):

    t = (x - zero) / scale          true IEEE divide
    t = rint(t + shift)             round-half-to-even
    t = clip(t, lo, hi)
    value = t * scale + zero        separate multiply and add (no FMA)
"""

import numpy as np


class UniformGrid:
    """Evenly spaced grid of `levels` points on [lo, hi] (codebook.py:9-14)."""

    def __init__(self, levels, lo, hi):
        self.levels = int(levels)
        self.lo = lo
        self.hi = hi
        assert self.lo < self.hi
        assert self.levels >= 2

    # -- attributes mirrored from the reference class (codebook.py:16-41) --
    def __len__(self):
        return self.levels

    def min(self):
        return self.lo

    def max(self):
        return self.hi

    @property
    def step(self):
        return (self.hi - self.lo) / (self.levels - 1)

    @property
    def values(self):
        return np.linspace(self.lo, self.hi, self.levels)

    # -- element-wise maps --------------------------------------------------
    def _position(self, x, shift, first, last):
        """Grid coordinate of x after an integer shift, rounded and clipped."""
        t = x - self.lo
        t /= self.step
        if shift:
            t += shift
        return t.round().clip(first, last)

    def _to_value(self, t):
        t *= self.step
        t += self.lo
        return t

    def index(self, x):
        """codebook.py:43-54 -- grid index with the narrowest unsigned dtype."""
        t = self._position(x, 0, 0, self.levels - 1)
        if self.levels <= 2**8:
            return t.astype(np.uint8)
        if self.levels <= 2**16:
            return t.astype(np.uint16)
        return t.astype(np.uint32)

    def value(self, x):
        """codebook.py:56-65 -- nearest grid value."""
        return self._to_value(self._position(x, 0, 0, self.levels - 1))

    def up(self, x):
        """codebook.py:67-77 -- the grid value one step above, saturating."""
        return self._to_value(self._position(x, 1, 1, self.levels - 1))

    def down(self, x):
        """codebook.py:79-89 -- the grid value one step below, saturating."""
        return self._to_value(self._position(x, -1, 0, self.levels - 2))

    __call__ = value

    # The local search of the reference calls these names (obq.py:257-258).
    quantize_up = up
    quantize_down = down
    quantize_value = value
    quantize_index = index


class TableGrid:
    """General codebook: sorted values and the limits between their bins (codebook.py:98-190).

    Restated on `np.searchsorted(limits, x, side="right")`, which is what `np.digitize` computes for
    increasing bins: the number of limits <= x.
    """

    def __init__(self, values, limits=None):
        self.values = np.array(values, dtype=np.float32)
        if limits is not None:
            self.limits = np.array(limits, dtype=np.float32)
        else:
            self.values.sort()
            self.limits = (self.values[:-1] + self.values[1:]) / 2
        assert self.values.ndim == 1 and self.values.size > 0
        assert (self.values[1:] > self.values[:-1]).all()
        assert self.limits.shape == (self.values.size - 1,)
        assert (self.limits >= self.values[:-1]).all() and (self.limits <= self.values[1:]).all()

    @staticmethod
    def nf4():
        """The NormalFloat4 values (codebook.py:297-320)."""
        return TableGrid([-1.0, -0.6961928009986877, -0.5250730514526367, -0.39491748809814453, -0.28444138169288635,
                          -0.18477343022823334, -0.09105003625154495, 0.0, 0.07958029955625534, 0.16093020141124725,
                          0.24611230194568634, 0.33791524171829224, 0.44070982933044434, 0.5626170039176941,
                          0.7229568362236023, 1.0])

    def __len__(self):
        return len(self.values)

    def min(self):
        return self.values[0]

    def max(self):
        return self.values[-1]

    def _bin(self, x):
        return np.searchsorted(self.limits, x, side="right")

    def index(self, x):
        b = self._bin(x)
        if len(self) <= 2**8:
            return b.astype(np.uint8)
        if len(self) <= 2**16:
            return b.astype(np.uint16)
        return b.astype(np.uint32)

    def value(self, x):
        return self.values[self._bin(x)]

    def up(self, x):
        return self.values[np.minimum(self._bin(x) + 1, len(self) - 1)]

    def down(self, x):
        return self.values[np.maximum(self._bin(x) - 1, 0)]

    __call__ = value
    quantize_up = up
    quantize_down = down
    quantize_value = value
    quantize_index = index
