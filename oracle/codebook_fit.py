"""Oracle: fitting a general codebook to data (Lloyd-Max).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates the training half of `Codebook` and `lloyd_max` of the reference on top of `grid.TableGrid`:

    bin_counts / bin_shares   codebook.py:190-195  (probabilities)
    code_entropy              codebook.py:197-203  (entropy)
    mean_square_miss          codebook.py:205-210  (mse)
    bin_centres               codebook.py:212-231  (centroids)
    drop_empty_bins           codebook.py:233-246  (remove_unused)
    lloyd_round               codebook.py:248-267  (improve)
    similar                   codebook.py:269-276  (close_to)
    pick_random               codebook.py:278-286  (Codebook.random; NumPy's global generator, as there)
    equal_mass                codebook.py:322-335  (Codebook.equiprobable)
    fit_lloyd_max             codebook.py:338-367  (lloyd_max)

NumPy decides the dtypes exactly as in the reference: float32 data keeps everything in float32 (bin means are
float32 pairwise sums), float64 data makes the values float64 until the next `copy()` casts them back.
"""

import numpy as np

from .grid import TableGrid


def check(g):
    """codebook.py:119-132."""
    assert g.values.ndim == 1
    assert g.values.size > 0
    assert np.isfinite(g.values).all()
    assert (g.values[1:] > g.values[:-1]).all()
    assert g.limits.ndim == 1
    assert g.limits.size == g.values.size - 1
    assert np.isfinite(g.limits).all()
    assert (g.limits[1:] > g.limits[:-1]).all()
    assert (g.limits >= g.values[:-1]).all()
    assert (g.limits <= g.values[1:]).all()


def copy(g):
    """codebook.py:113-117 (clone): through the constructor, i.e. back to float32."""
    return TableGrid(g.values.copy(), g.limits.copy())


def bin_counts(g, data):
    return np.bincount(g.index(data), minlength=len(g.values))


def bin_shares(g, data):
    return bin_counts(g, data) / len(data)


def code_entropy(g, data):
    p = bin_shares(g, data)
    p = p[p > 0]
    return -(p * np.log2(p)).sum()


def mean_square_miss(g, data):
    return np.square(data - g.value(data)).mean()


def bin_centres(g, data):
    labels = g.index(data)
    out = []
    last = len(g.values) - 1
    for k in range(last + 1):
        members = data[labels == k]
        if len(members) != 0:
            out.append(members.mean())
        elif k == 0:
            out.append(g.limits[0] - 1.0e-6)
        elif k == last:
            out.append(g.limits[-1] + 1.0e-6)
        else:
            out.append((g.limits[k - 1] + g.limits[k]) / 2)
    return np.array(out)


def drop_empty_bins(g, data):
    counts = bin_counts(g, data)
    if (counts == 0).any():
        g.values = g.values[counts != 0]
        g.limits = g.limits[counts[:-1] != 0]  # the limit to the right of a dropped bin goes with it
        if counts[-1] == 0:
            g.limits = g.limits[:-1]
        check(g)


def lloyd_round(g, data, lagrange_mult=0.0):
    if lagrange_mult != 0.0:
        drop_empty_bins(g, data)
        v = g.values
        bits = -np.log2(bin_shares(g, data))
        slope = (bits[1:] - bits[:-1]) / (v[1:] - v[:-1])
        g.limits = (v[:-1] + v[1:]) / 2 + lagrange_mult * slope / 2
        g.limits.sort()
    else:
        v = g.values
        g.limits = (v[:-1] + v[1:]) / 2
    g.values = bin_centres(g, data)
    check(g)


def similar(a, b, tol=1.0e-6):
    if len(a) != len(b):
        return False
    span = max(a.values.max() - a.values.min(), 1.0e-10)
    return np.allclose(a.values, b.values, atol=tol * span)


def pick_random(data, size):
    distinct = np.unique(data)
    return TableGrid(np.random.choice(distinct, min(size, distinct.size), replace=False))


def equal_mass(data, size):
    parts = [p for p in np.array_split(np.sort(data), size) if len(p) > 0]
    limits = [(parts[k][-1] + parts[k + 1][0]) / 2 for k in range(len(parts) - 1)]
    g = TableGrid([p.mean() for p in parts], limits)
    check(g)
    g.values = bin_centres(g, data)
    return g


def fit_lloyd_max(data, size, lagrange_mult=0.0, max_iter=100, tol=1e-6, random_init=False, sample_count=None, rounds=None):
    """`rounds`: optional list that receives the number of improvement rounds run (test evidence, not in the reference)."""
    data = data.reshape((-1,))
    if sample_count is not None:
        wanted = size * sample_count
        if wanted < len(data):
            data = np.random.choice(data, wanted, replace=False)
    data = np.sort(data)
    g = pick_random(data, size) if random_init else equal_mass(data, size)
    done = 0
    for _ in range(max_iter):
        nxt = copy(g)
        lloyd_round(nxt, data, lagrange_mult)
        done += 1
        if similar(nxt, g, tol):
            break
        g = nxt
    if rounds is not None:
        rounds.append(done)
    return g
