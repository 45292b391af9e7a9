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


def make_samples(count, seed, dtype=np.float32):
    """1-D data to fit a codebook to: ~N(0, 1) with every 97th sample stretched by 4 (a heavy tail)."""
    z = normal_grid(seed, 8, 1, count)[0]
    z[::97] *= 4.0
    return z.astype(dtype)


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


# ---------------------------------------------------------------------------------------------------
# The same generator on the GPU (torch): identical bytes -- integer hashing in wrapping int64 arithmetic,
# IEEE float64 +, -, *, / element by element (no fused multiply-add: every product is a tensor of its own),
# exact integer float64 matrix products -- so that benchmarks can set up whole models in seconds
# (tests/test_gpu_parity.py::test_device_generator_makes_the_same_bytes holds it to the host generator).
def _t_splitmix64(x):
    import torch  # noqa: F401

    def lsr(v, k):  # logical shift right of an int64 tensor
        return (v >> k) & ((1 << (64 - k)) - 1)

    z = x + (-7046029254386353131)            # 0x9E3779B97F4A7C15 as int64
    z = (z ^ lsr(z, 30)) * (-4658895280553007687)  # 0xBF58476D1CE4E5B9
    z = (z ^ lsr(z, 27)) * (-7723592293110705685)  # 0x94D049BB133111EB
    return z ^ lsr(z, 31)


def _t_hash_grid(seed, stream, rows, cols, device, row0=0):
    import torch

    key = int(np.array([_key(seed, stream)], dtype=np.uint64).view(np.int64)[0])
    i = (torch.arange(row0, row0 + rows, dtype=torch.int64, device=device) << 32)[:, None]
    j = torch.arange(cols, dtype=torch.int64, device=device)[None, :]
    return _t_splitmix64((i | j) ^ key)


def _t_normal_grid(seed, stream, rows, cols, device, row0=0):
    import torch

    u = _t_hash_grid(seed, stream, rows, cols, device, row0)
    s = (u & 0xFFFF) + ((u >> 16) & 0xFFFF) + ((u >> 32) & 0xFFFF) + ((u >> 48) & 0xFFFF)
    return (s.to(torch.float64) - 131070.0) / 37837.2


def make_layer_device(R, n, seed, device, T=None, chunk=4096, keep=("W", "H", "mean", "scale")):
    """make_layer on the GPU: dict of device tensors W, H, mean, scale (+ T, seed), bit-identical to make_layer."""
    import torch

    T = 2 * n if T is None else T
    W = (0.02 * _t_normal_grid(seed, 1, R, n, device)).to(torch.float32)
    gain, load, offset = (torch.from_numpy(np.ascontiguousarray(a)).to(device) for a in _channel_params(n, seed))
    acc = torch.zeros((n, n), dtype=torch.float64, device=device)
    col = torch.zeros(n, dtype=torch.float64, device=device)
    for t0 in range(0, T, chunk):
        rows = min(chunk, T - t0)
        x = _t_normal_grid(seed, 6, rows, n, device, row0=t0) * gain[None, :]
        f = _t_normal_grid(seed, 7, rows, 8, device, row0=t0)
        for k in range(8):
            x = x + f[:, k:k + 1] * load[None, :, k]
        x = x + offset[None, :]
        xi = torch.round(x * _FIX)  # half-to-even, like np.rint; integers
        acc += xi.T @ xi
        col += xi.sum(dim=0)
    assert float(acc.abs().max()) < 2.0**53
    H = (acc / (_FIX * _FIX * T)).to(torch.float32)
    mean = (col / (_FIX * T)).to(torch.float32)
    scale = torch.clamp(W.abs().amax(dim=1) / np.float32(1.0) * np.float32(0.6), min=1.0e-16).to(torch.float32)
    out = dict(W=W, H=H, mean=mean, scale=scale)
    out = {k: v for k, v in out.items() if k in keep}
    out.update(T=T, seed=seed)
    return out
