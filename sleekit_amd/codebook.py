"""Drop-in for `sleekit.codebook`: the quantizers of the hot path.

`UniformCodebook` keeps the reference's interface (sleekit/codebook.py:4-95) and runs
every map on the GPU through `slk_codebook_apply`, in float32 with the reference's exact
rounding (true divide, round-half-to-even, separate multiply and add).

`Codebook` (sleekit/codebook.py:98-190, 277-320) is the general, table-driven codebook --
sorted values and the limits between their bins, `np.digitize` semantics -- with the same four
maps on the GPU; every kernel of the path takes either kind.  Its TRAINING helpers (`probabilities`,
`entropy`, `mse`, `centroids`, `remove_unused`, `improve`, `equiprobable`, `random` and `lloyd_max`,
sleekit/codebook.py:190-286, 322-367) keep the data on the GPU: a radix sort and one statistics pass per
round (`slk_sort_f32`, `slk_unique_f32`, `slk_codebook_stats`), the arithmetic on the codebook's few entries on
the host like the reference.
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

    def _device_table(self, least=2):
        if len(self) < least or len(self) > 256:
            raise NotImplementedError("general codebooks on the device hold 2 to 256 entries")
        device = dev.require_gpu()
        both = np.concatenate([self.values, self.thresholds]).astype(np.float32)
        key = (str(device), both.tobytes())  # training moves values and limits: the cached table follows them
        if key not in self._table:
            self._table.clear()
            self._table[key] = torch.as_tensor(both, device=device)
        return self._table[key]

    def _abi(self):
        table = self._device_table()
        return len(self), float(self.values[0]), float(self.values[-1]), table

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

    # -- codebook training (sleekit/codebook.py:190-286, 322-335) -----------------------------------------------
    # One pass of `slk_codebook_stats` over the data on the GPU gives the per-bin counts, the per-bin sums and the
    # squared miss; the arithmetic on the (at most 256) codebook entries stays on the host, written like the
    # reference's.  Data is processed in float32 on the device (float64 input is rounded once); results come back in
    # the input's dtype like NumPy's would.  Means are float64-accurate sums, not the reference's float32 pairwise
    # sums: equal to float32 rounding (tests/test_gpu_codebook_fit.py states the tolerance).
    def _stats(self, data):
        """(counts int64, sums float64, squared miss, element count, result dtype) of `data` under this codebook."""
        x, dt = _flat(data)
        return (*_bin_stats(x, len(self), self._device_table(least=1)), dt)

    def probabilities(self, data):
        """Share of the data in each bin (sleekit/codebook.py:190-195)."""
        if getattr(data, "ndim", 1) != 1:
            raise ValueError("object too deep for desired array")  # what np.bincount says there
        counts = self._stats(data)[0]
        return counts / len(data)

    def entropy(self, data):
        """Entropy in bits of the quantized data (sleekit/codebook.py:197-203)."""
        probs = self.probabilities(data)
        probs = probs[probs > 0]
        return -(probs * np.log2(probs)).sum()

    def mse(self, data):
        """Mean squared quantization error (sleekit/codebook.py:205-210)."""
        _, _, miss, count, dt = self._stats(data)
        return dt(miss / count)

    def centroids(self, data):
        """Mean of the data in each bin under the current limits; an empty bin gets a value just outside / in the
        middle of its limits (sleekit/codebook.py:212-231)."""
        counts, sums, _, _, dt = self._stats(data)
        ret = []
        last = len(self.values) - 1
        for k in range(last + 1):
            if counts[k] != 0:
                ret.append(dt(sums[k] / counts[k]))
            elif k == 0:
                ret.append(self.thresholds[0] - 1.0e-6)
            elif k == last:
                ret.append(self.thresholds[-1] + 1.0e-6)
            else:
                ret.append((self.thresholds[k - 1] + self.thresholds[k]) / 2)
        return np.array(ret)

    def _drop_empty(self, counts):
        if (counts == 0).any():
            self.values = self.values[counts != 0]
            self.thresholds = self.thresholds[counts[:-1] != 0]  # the limit to the right of a dropped bin goes with it
            if counts[-1] == 0:
                self.thresholds = self.thresholds[:-1]
            self.check()

    def remove_unused(self, data):
        """Drop the codewords no data point maps to; the remaining limits stay (sleekit/codebook.py:233-246)."""
        self._drop_empty(self._stats(data)[0])

    def improve(self, data, lagrange_mult=0.0):
        """One Lloyd-Max round: limits from the values (plus the entropy penalty), values from the centroids
        (sleekit/codebook.py:248-267)."""
        if lagrange_mult != 0.0:
            counts, _, _, count, _ = self._stats(data)
            self._drop_empty(counts)
            # dropping empty bins merges each with a neighbour and moves no data point: the shares of the
            # remaining bins are the non-zero counts (codebook.py:255 runs a second pass for them)
            v = self.values
            bits = -np.log2(counts[counts != 0] / count)
            slope = (bits[1:] - bits[:-1]) / (v[1:] - v[:-1])
            self.thresholds = (v[:-1] + v[1:]) / 2 + lagrange_mult * slope / 2
            self.thresholds.sort()  # the penalty can throw the order away
        else:
            v = self.values
            self.thresholds = (v[:-1] + v[1:]) / 2
        self.values = self.centroids(data)
        self.check()

    def close_to(self, other, tol=1.0e-6):
        """Whether two codebooks are the same up to `tol` of the value range (sleekit/codebook.py:269-276)."""
        if len(self) != len(other):
            return False
        data_range = max(self.values.max() - self.values.min(), 1.0e-10)
        return np.allclose(self.values, other.values, atol=tol * data_range)

    @staticmethod
    def random(data, codebook_size):
        """Codebook of distinct data values drawn with NumPy's global generator (sleekit/codebook.py:277-286): the
        draw consumes the generator exactly like the reference's np.random.choice(values, k, replace=False)."""
        x, dt = _flat(data)
        distinct = _distinct(x if getattr(data, "is_sorted", False) else _sorted(x))
        pick = np.random.choice(distinct.numel(), min(codebook_size, distinct.numel()), replace=False)
        chosen = distinct[torch.from_numpy(np.asarray(pick, dtype=np.int64)).to(x.device)]
        return Codebook(chosen.cpu().numpy().astype(dt))

    @staticmethod
    def equiprobable(data, codebook_size):
        """Codebook whose bins hold equal shares of the data (sleekit/codebook.py:322-335)."""
        x, dt = _flat(data)
        return _equiprobable_sorted(_sorted(x), codebook_size, dt)


def _flat(data):
    """(flat float32 device tensor, NumPy scalar type results are reported in)."""
    if isinstance(data, _Tagged):
        return data.x, data.dt
    if isinstance(data, torch.Tensor):
        dt = np.float64 if data.dtype == torch.float64 else np.float32
    else:
        data = np.asarray(data)
        dt = np.float32 if data.dtype == np.float32 else np.float64  # Python lists and integers: float64, like NumPy
    return dev.to_device(data, torch.float32).reshape(-1), dt


class _Tagged:
    """A flat float32 device tensor that remembers the dtype its data came in (lloyd_max keeps one across rounds)."""

    ndim = 1

    def __init__(self, x, dt, is_sorted=False):
        self.x, self.dt, self.is_sorted = x, dt, is_sorted

    def __len__(self):
        return self.x.numel()


def _bin_stats(x, levels, table, by_position=False):
    """slk_codebook_stats on a flat float32 device tensor: (counts int64, sums float64, squared miss, element count).
    `by_position`: the bins are the `levels` pieces of np.array_split by position, no codebook."""
    if levels > 256:
        raise NotImplementedError("codebook statistics on the device take at most 256 bins")
    out = torch.empty(2 * levels + 1, dtype=torch.int64, device=x.device)  # counts | sums (float64 bits) | squared miss: one copy back
    ws_bytes = int(_lib.lib.slk_codebook_stats_workspace_bytes())
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device)
    base = out.data_ptr()
    _lib.check(
        _lib.lib.slk_codebook_stats(
            dev.ptr(x), x.numel(), levels, 0.0, 0.0, dev.ptr(table), 1 if by_position else 0, base, base + 8 * levels,
            base + 16 * levels, dev.ptr(ws), ws_bytes, dev.stream_handle(),
        )
    )
    host = out.cpu().numpy()
    sums = host[levels:].view(np.float64)
    return host[:levels], sums[:levels], float(sums[levels]), x.numel()


def _sort_workspace(x):
    ws_bytes = int(_lib.lib.slk_sort_workspace_bytes(x.numel()))
    return torch.empty(max(ws_bytes, 256), dtype=torch.uint8, device=x.device), ws_bytes


def _sorted(x):
    out = torch.empty_like(x)
    ws, ws_bytes = _sort_workspace(x)
    _lib.check(_lib.lib.slk_sort_f32(dev.ptr(x), x.numel(), dev.ptr(out), dev.ptr(ws), ws_bytes, dev.stream_handle()))
    return out


def _distinct(x_sorted):
    out = torch.empty_like(x_sorted)
    n_out = torch.zeros(1, dtype=torch.int32, device=x_sorted.device)
    ws, ws_bytes = _sort_workspace(x_sorted)
    _lib.check(_lib.lib.slk_unique_f32(dev.ptr(x_sorted), x_sorted.numel(), dev.ptr(out), dev.ptr(n_out), dev.ptr(ws), ws_bytes,
                                       dev.stream_handle()))
    return out[: int(n_out.item())]


def _equiprobable_sorted(x, codebook_size, dt):
    n = x.numel()
    q, r = divmod(n, codebook_size)  # np.array_split: the first r parts hold q + 1 elements, the others q
    sizes = [q + 1] * r + [q] * (codebook_size - r)
    sizes = [m for m in sizes if m > 0]
    starts = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    # the elements either side of each cut
    edge = np.stack([starts[1:-1] - 1, starts[1:-1]], axis=1).reshape(-1)
    ends = x[torch.from_numpy(edge).to(x.device)].cpu().numpy().astype(dt).reshape(-1, 2)
    limits = (ends[:, 0] + ends[:, 1]) / 2
    counts, sums, _, _ = _bin_stats(x, codebook_size, None, by_position=True)
    values = [dt(sums[k] / counts[k]) for k in range(codebook_size) if counts[k] != 0]
    cb = Codebook(values, limits)
    cb.values = cb.centroids(_Tagged(x, dt, is_sorted=True))
    return cb


def lloyd_max(data, codebook_size, lagrange_mult=0.0, max_iter=100, tol=1e-6, random_init=False, sample_count=None):
    """Lloyd-Max scalar quantizer design (sleekit/codebook.py:338-367): a codebook that minimises the mse plus
    `lagrange_mult` times the entropy.  The data lives on the GPU for the whole fit: one sort, then one or two passes
    of `slk_codebook_stats` per round; the optional subsample is drawn with NumPy's global generator like the reference's."""
    x, dt = _flat(data)
    if sample_count is not None:
        nsamples = codebook_size * sample_count
        if nsamples < x.numel():
            keep = np.random.choice(x.numel(), nsamples, replace=False)
            x = x[torch.from_numpy(np.asarray(keep, dtype=np.int64)).to(x.device)].contiguous()
    x = _sorted(x)
    tagged = _Tagged(x, dt, is_sorted=True)
    if random_init:
        codebook = Codebook.random(tagged, codebook_size)
    else:
        codebook = _equiprobable_sorted(x, codebook_size, dt)
    for _ in range(max_iter):
        new_codebook = codebook.clone()
        new_codebook.improve(tagged, lagrange_mult)
        if new_codebook.close_to(codebook, tol):
            break
        codebook = new_codebook
    return codebook
