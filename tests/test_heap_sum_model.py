"""Host-side model of the workgroup row sum of the local-search kernel (sleekit_amd/csrc/npsum.h: HeapSum).

The kernel derives NumPy's pairwise summation tree in heap numbering -- every thread walks the bits of its heap
index from the root -- sums each leaf with 8 interleaved accumulators and combines the levels bottom-up.  This is
that algorithm in Python, statement for statement, held against np.sum itself: a wrong split, depth bound or
combination order shows up here, without a GPU.
"""

import numpy as np
import pytest

CHUNK, BLOCK = 8192, 128
f32 = np.float32


def heap_plan(n):
    lo, ln = np.zeros(512, dtype=np.int64), np.zeros(512, dtype=np.int64)
    leaves = []
    for c in range(2):
        for t in range(1, 256):
            a, m = c * CHUNK, min(CHUNK, n - c * CHUNK)
            ok = m > 0
            if ok:
                depth = t.bit_length() - 1
                for b in range(depth - 1, -1, -1):
                    if m <= BLOCK:
                        ok = False
                        break
                    h = (m // 2) & ~7
                    if (t >> b) & 1:
                        a, m = a + h, m - h
                    else:
                        m = h
            lo[256 * c + t], ln[256 * c + t] = a, (m if ok else 0)
            if ok and m <= BLOCK:
                leaves.append(256 * c + t)
    return lo, ln, leaves


def heap_sum(x):
    n = len(x)
    lo, ln, leaves = heap_plan(n)
    assert sum(ln[s] for s in leaves) == n, "the leaves must tile the row"
    val = np.zeros(512, dtype=f32)
    for s in leaves:
        a, m = lo[s], ln[s]
        if m < 8:
            acc = f32(-0.0)
            for i in range(m):
                acc = f32(acc + x[a + i])
        else:
            body = m - (m & 7)
            r = [x[a + k] for k in range(8)]
            for i in range(8, body, 8):
                for k in range(8):
                    r[k] = f32(r[k] + x[a + i + k])
            for step in (1, 2, 4):  # the xor-shuffle combination
                r = [f32(r[k] + r[k ^ step]) for k in range(8)]
            acc = r[0]
            for i in range(body, m):
                acc = f32(acc + x[a + i])
        val[s] = acc
    for d in range(6, -1, -1):
        for k in range(2 << d):
            base, h = 256 * (k >> d), (1 << d) + (k & ((1 << d) - 1))
            if ln[base + h] > BLOCK:
                assert ln[base + 2 * h] > 0 and ln[base + 2 * h + 1] > 0
                val[base + h] = f32(val[base + 2 * h] + val[base + 2 * h + 1])
    total = f32(f32(0.0) + val[1])
    if n > CHUNK:
        total = f32(total + val[257])
    return total


@pytest.mark.parametrize("n", [1, 2, 7, 8, 9, 15, 16, 17, 96, 100, 127, 128, 129, 130, 172, 257, 768, 1000, 1024, 1100, 3072,
                               4096, 8191, 8192, 8193, 8200, 11008, 16383, 16384])
def test_heap_sum_is_numpy_sum(n):
    rng = np.random.default_rng(n)
    for trial in range(3):
        x = (rng.standard_normal(n) * np.exp(rng.uniform(-3, 3, n))).astype(f32)
        assert heap_sum(x) == x.sum() == x.reshape(1, n).sum(axis=-1)[0], (n, trial)


def test_depth_bound():
    """No chunk length needs more than 7 levels below the root (heap indices stay below 256)."""
    for m in list(range(1, 600)) + list(range(7800, CHUNK + 1)):
        lo, ln, leaves = heap_plan(m)
        assert sum(ln[s] for s in leaves) == m, m
