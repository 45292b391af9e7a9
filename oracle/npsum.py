"""Oracle: model of NumPy's float32 pairwise summation order.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The reference takes `H.diagonal().mean()` in float32 (obq.py:32, 198); the
value feeds the damping term, so the device kernel reproduces NumPy's
summation ORDER rather than just its value.  This module states that order in
plain Python so tests can check (a) the model against `np.add.reduce` itself
and (b) the device kernel against the model.

Order (numpy/_core/src/umath/loops_utils.h.src, `*_pairwise_sum`, block 128):
    n < 8      : left-to-right from 0.0 (the reduction starts at the identity)
    n <= 128   : eight interleaved accumulators r[k] = a[k]; r[k] += a[8m+k];
                 combined ((r0+r1)+(r2+r3)) + ((r4+r5)+(r6+r7)); then the
                 n % 8 tail left-to-right
    n > 128    : split at h = (n/2) rounded down to a multiple of 8; sum(left) + sum(right)
The reduction's running value starts at 0.0 and adds the pairwise sum of each
inner-loop chunk; the ufunc machinery feeds chunks of 8192 elements
(checked against np.add.reduce for contiguous and strided float32 vectors).
"""

import numpy as np

_BLOCK = 128


def _pairwise(a, lo, n):
    f = np.float32
    if n < 8:
        # NumPy seeds this branch with -0.0 so that a sum of negative zeros
        # keeps its sign (numpy >= 2.0); otherwise identical to a 0.0 seed.
        r = f(-0.0)
        for i in range(n):
            r = f(r + a[lo + i])
        return r
    if n <= _BLOCK:
        r = [a[lo + k] for k in range(8)]
        i = 8
        while i < n - (n % 8):
            for k in range(8):
                r[k] = f(r[k] + a[lo + i + k])
            i += 8
        res = f(f(f(r[0] + r[1]) + f(r[2] + r[3])) + f(f(r[4] + r[5]) + f(r[6] + r[7])))
        while i < n:
            res = f(res + a[lo + i])
            i += 1
        return res
    h = n // 2
    h -= h % 8
    return np.float32(_pairwise(a, lo, h) + _pairwise(a, lo + h, n - h))


def pairwise_sum_f32(a, chunk=8192):
    """Sum of a 1-D float32 array in NumPy's order; `chunk` = inner-loop length (None: whole array)."""
    a = np.asarray(a, dtype=np.float32)
    n = a.shape[0]
    step = n if chunk is None else chunk
    total = np.float32(0.0)
    pos = 0
    while pos < n:
        m = min(step, n - pos)
        total = np.float32(total + _pairwise(a, pos, m))
        pos += m
    return total


def mean_f32(a, chunk=8192):
    """np.mean of a float32 vector: pairwise sum, then one float32 divide by the length."""
    a = np.asarray(a, dtype=np.float32)
    return np.float32(pairwise_sum_f32(a, chunk) / np.float32(a.shape[0]))
