"""Synthetic layer statistics: weights W, calibration activations X, Hessian H.

Data generation for tests and benchmarks only (host NumPy); nothing here is on
the quantization path.  The reference ships no inputs (its `data/` directory is
git-ignored), so parity and throughput are measured on layers made here.

The bytes must be identical in the build container and on the GPU box, whatever
BLAS or libm is installed, so the generator uses integer hashing and exact
arithmetic only:

  * u(i, j)   = splitmix64(key ^ (i << 32 | j)),  key = splitmix64(seed ^ stream)
  * z(i, j)   = (sum of the four 16-bit fields of u - 131070) / 37837.2  ~ N(0, 1)
                (Irwin-Hall, |z| <= 3.47; IEEE +,-,*,/ only)
  * W[i, j]   = float32(0.02 * z)
  * X[t, j]   = round(256 * (g_j * z[t, j] + sum_k a[j, k] * f[t, k] + b_j)) / 256
                with per-channel gain g_j in [0.5, 2.5] (x8 on 20 outlier
                channels), 8 shared latent factors f and an offset b_j: a
                correlated, badly scaled input like a transformer block sees.
  * H         = X^T X / T computed on the INTEGERS 256*X in float64.  Every
                partial sum is an integer below 2**53, so the product is exact
                for any summation order and H is bit-identical everywhere.
  * mean      = column mean of X, exact the same way.

`T` defaults to 2n tokens (full-rank H).
"""

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)
_GOLD = np.uint64(0x9E3779B97F4A7C15)
_MUL1 = np.uint64(0xBF58476D1CE4E5B9)
_MUL2 = np.uint64(0x94D049BB133111EB)
_FIX = 256.0  # activations live on a 1/256 grid


def splitmix64(x):
    """Vectorised splitmix64 finaliser on uint64 arrays (wrapping arithmetic)."""
    with np.errstate(over="ignore"):
        z = (x + _GOLD) & _M64
        z = ((z ^ (z >> np.uint64(30))) * _MUL1) & _M64
        z = ((z ^ (z >> np.uint64(27))) * _MUL2) & _M64
        return z ^ (z >> np.uint64(31))


def _key(seed, stream):
    return splitmix64(np.array([np.uint64(seed) ^ (np.uint64(stream) << np.uint64(48))], dtype=np.uint64))[0]


def hash_grid(seed, stream, rows, cols, row0=0):
    """uint64 hash of every (row0 + i, j) cell."""
    i = (np.arange(row0, row0 + rows, dtype=np.uint64) << np.uint64(32))[:, None]
    j = np.arange(cols, dtype=np.uint64)[None, :]
    return splitmix64(_key(seed, stream) ^ (i | j))


def normal_grid(seed, stream, rows, cols, row0=0):
    """Approximately standard-normal float64 from four 16-bit fields per hash."""
    u = hash_grid(seed, stream, rows, cols, row0)
    m = np.uint64(0xFFFF)
    s = (u & m) + ((u >> np.uint64(16)) & m) + ((u >> np.uint64(32)) & m) + (u >> np.uint64(48))
    return (s.astype(np.float64) - 131070.0) / 37837.2


def uniform_grid(seed, stream, rows, cols):
    """Uniform float64 in [0, 1) from the top 53 bits."""
    u = hash_grid(seed, stream, rows, cols)
    return (u >> np.uint64(11)).astype(np.float64) / float(1 << 53)


def make_weights(R, n, seed):
    """W (R, n) float32, ~N(0, 0.02^2)."""
    return (0.02 * normal_grid(seed, 1, R, n)).astype(np.float32)


def _channel_params(n, seed):
    gain = 0.5 + 2.0 * uniform_grid(seed, 2, 1, n)[0]
    hot = (hash_grid(seed, 3, 1, 20)[0] % np.uint64(n)).astype(np.int64)
    gain[hot] *= 8.0
    load = 0.4 * normal_grid(seed, 4, n, 8)
    offset = 0.5 * normal_grid(seed, 5, 1, n)[0]
    return gain, load, offset


def make_activations(T, n, seed, t0=0):
    """X (T, n) float64 on the 1/256 grid; rows t0 .. t0+T-1 of the layer's token stream."""
    gain, load, offset = _channel_params(n, seed)
    z = normal_grid(seed, 6, T, n, row0=t0)
    f = normal_grid(seed, 7, T, 8, row0=t0)
    x = z * gain[None, :]
    for k in range(8):  # fixed order, element-wise only: no BLAS, no FMA
        x += f[:, k : k + 1] * load[None, :, k]
    x += offset[None, :]
    return np.rint(x * _FIX) / _FIX


def make_hessian(n, seed, T=None, chunk=2048, dead=(), device=None):
    """(H float32 (n, n), mean float32 (n,), T).  Exact integer accumulation in float64.

    `dead`: column indices whose activations are forced to zero (inputs that never fire).
    `device`: a torch device to run the float64 products on (same bytes: the sums are exact
    integers whatever the order), used by bench.py to set up large layers quickly.
    """
    T = 2 * n if T is None else T
    col = np.zeros(n, dtype=np.float64)
    if device is not None:
        import torch

        acc_t = torch.zeros((n, n), dtype=torch.float64, device=device)
    else:
        acc = np.zeros((n, n), dtype=np.float64)
    for t0 in range(0, T, chunk):
        xi = make_activations(min(chunk, T - t0), n, seed, t0) * _FIX  # integers
        if len(dead):
            xi[:, list(dead)] = 0.0
        if device is not None:
            xt = torch.from_numpy(xi).to(device)
            acc_t += xt.T @ xt
        else:
            acc += xi.T @ xi
        col += xi.sum(axis=0)
    if device is not None:
        acc = acc_t.cpu().numpy()
    assert np.abs(acc).max() < 2.0**53
    H = acc / (_FIX * _FIX * T)
    mean = col / (_FIX * T)
    return H.astype(np.float32), mean.astype(np.float32), T


def make_scale(W, levels_hi=1.0, factor=0.6):
    """Per-row scale = factor * max|row| / levels_hi (float32), floored at 1e-16.

    Stands in for the reference's grid search (scaling.py:98-134), which is a
    pre-step outside the timed path; 0.6 is where that search typically lands
    for 3-bit grids.
    """
    s = (np.abs(W).max(axis=1) / np.float32(levels_hi)).astype(np.float32)
    s = np.maximum(s * np.float32(factor), np.float32(1.0e-16))
    return s.astype(np.float32)


def make_layer(R, n, seed, T=None, dead=(), device=None):
    """All inputs of one layer: dict(W, H, mean, scale, T)."""
    W = make_weights(R, n, seed)
    H, mean, T = make_hessian(n, seed, T=T, dead=dead, device=device)
    return dict(W=W, H=H, mean=mean, scale=make_scale(W), T=T, seed=seed)
