"""Oracle: running input statistics of a layer (Hessian accumulation).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates the Linear / Conv branches of `Sleekit._prepare_input` and
`Sleekit.add_batch` (sleekit/statistics.py:37-87) on NumPy arrays:

    X      = inp.reshape(-1, in).T            (in, T)   float32
    factor = count / (count + T)
    mean   = mean * factor + X.sum(1) / count'
    H      = H * factor + X @ X.T / count'

The matrix product is a float32 GEMM whose summation order is the BLAS's own,
so comparisons against it are by tolerance (rtol 1e-5), never bit-wise.
"""

import numpy as np


class RunningStats:
    def __init__(self, n):
        self.mean = np.zeros(n, dtype=np.float32)
        self.hessian = np.zeros((n, n), dtype=np.float32)
        self.count = 0

    def add_tokens(self, tokens):
        """`tokens`: (..., n) activations of an nn.Linear input (statistics.py:41-43, 76-87)."""
        X = np.ascontiguousarray(tokens, dtype=np.float32).reshape(-1, tokens.shape[-1]).T
        self.add_columns(X)

    def add_columns(self, X):
        """`X`: (n, T) already unfolded samples, one per column (statistics.py:76-87)."""
        X = X.astype(np.float32, copy=False)
        added = X.shape[1]
        factor = self.count / (self.count + added)
        self.count += added
        self.mean *= np.float32(factor)
        self.hessian *= np.float32(factor)
        self.mean += X.sum(axis=1) / np.float32(self.count)
        self.hessian += X @ X.T / np.float32(self.count)


def bias_correction(W, Qw, mean):
    """statistics.py:187-190: the bias shift that compensates E[x] (W - Qw)."""
    return ((W - Qw) * mean).sum(axis=1)
