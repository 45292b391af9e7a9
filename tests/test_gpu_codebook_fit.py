"""Codebook TRAINING on the GPU (sleekit_amd/codebook.py over slk_codebook_stats / slk_sort_f32 / slk_unique_f32)
against the reference's outputs (tests/golden/codebook_fit.npz) and the CPU oracle (oracle/codebook_fit.py).

What is exact and what is not:
  * counts, shares, entropy, dropped bins, limits read off the sorted data, drawn starts: BIT-EQUAL (integers, or
    float32 arithmetic on the same elements);
  * bin means: the reference takes float32 pairwise sums (np.mean of a float32 array), the device a float64-accurate
    sum: equal to a float32 rounding of the mean.  A rounding in a value moves a limit by a rounding, which in turn
    may move one data point across it (one point of 20000 changes a mean by ~1e-4 of a bin width), so a Lloyd-Max
    trajectory is compared within FIT_TOL of the value range per round, and the converged codebook -- the
    reference stops when a round moves the values by less than 1e-6 of the range, not at the fixed point --
    within FINAL_TOL.  Measured on the fixtures' seven cases: <= 2.4e-8 of the range per round, <= 4.8e-8 at the end
    (no data point changed bins); the tolerances are ten times that.
"""

import ctypes
import json

import numpy as np
import pytest
import torch

from oracle import codebook_fit as fit
from oracle import grid
from sleekit_amd import synth

pytestmark = pytest.mark.gpu

MEAN_TOL = 2.0e-7   # of max|x|: a bin mean against the float64 mean of the same points (float32 rounding of the result)
FIT_TOL = 2.0e-7    # of the value range: values / limits after 1-3 rounds against the reference's
FINAL_TOL = 5.0e-7  # of the value range: converged codebooks


@pytest.fixture(scope="module")
def cbm():
    assert torch.cuda.is_available(), "these tests need the GPU"
    from sleekit_amd import codebook

    return codebook


def span(v):
    return float(np.max(v) - np.min(v))


def close(got, want, tol, what):
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    worst = float(np.abs(got - want).max()) if got.size else 0.0
    assert worst <= tol, (what, worst, tol)
    return worst


def raw_stats(x, table_values=None, table_limits=None, uniform=None, by_position=0, levels=None):
    """slk_codebook_stats through the C ABI on a device tensor."""
    from sleekit_amd import _device as dev
    from sleekit_amd import _lib

    if table_values is not None:
        table = torch.as_tensor(np.concatenate([table_values, table_limits]).astype(np.float32), device=x.device)
        levels, lo, hi = len(table_values), 0.0, 0.0
    elif uniform is not None:
        table, (levels, lo, hi) = None, uniform
    else:
        table, lo, hi = None, 0.0, 0.0
    counts = torch.empty(levels, dtype=torch.int64, device=x.device)
    sums = torch.empty(levels, dtype=torch.float64, device=x.device)
    miss = torch.empty(1, dtype=torch.float64, device=x.device)
    nb = int(_lib.lib.slk_codebook_stats_workspace_bytes())
    ws = torch.empty(nb, dtype=torch.uint8, device=x.device)
    rc = _lib.lib.slk_codebook_stats(dev.ptr(x), x.numel(), levels, lo, hi, dev.ptr(table), by_position, dev.ptr(counts), dev.ptr(sums),
                                     dev.ptr(miss), dev.ptr(ws), nb, dev.stream_handle())
    assert rc == 0, _lib.lib.slk_last_error()
    return counts.cpu().numpy(), sums.cpu().numpy(), float(miss.cpu()[0])


# ------------------------------------------------------------------------------------------ the statistics pass
@pytest.mark.parametrize("count", [1, 3, 4, 1023, 1024, 1025, 100003, 1 << 20])
def test_stats_against_numpy(cbm, count):
    data = synth.make_samples(count + 1, 40 + count % 7)
    for tag, g in (("nf4", grid.TableGrid.nf4()), ("odd", grid.TableGrid([-3.0, -0.2, 0.0, 0.1, 5.0])),
                   ("256", grid.TableGrid(np.linspace(-4, 4, 256))), ("one", None), ("uniform", grid.UniformGrid(8, -2, 2))):
        for offset in (0, 1):  # 1: a view that is not 16-byte aligned
            x = data[offset:offset + count]
            xd = torch.as_tensor(data, device="cuda")[offset:offset + count]
            if tag == "uniform":
                counts, sums, miss = raw_stats(xd, uniform=(8, -2.0, 2.0))
                idx, val = g.index(x.copy()), g.value(x.copy())
            elif tag == "one":
                counts, sums, miss = raw_stats(xd, table_values=np.float32([0.5]), table_limits=np.float32([]))
                idx, val = np.zeros(count, dtype=np.int64), np.full(count, 0.5, dtype=np.float32)
            else:
                counts, sums, miss = raw_stats(xd, table_values=g.values, table_limits=g.limits)
                idx, val = g.index(x), g.value(x)
            levels = len(counts)
            assert np.array_equal(counts, np.bincount(idx, minlength=levels)), (tag, count)
            want = np.bincount(idx, weights=x.astype(np.float64), minlength=levels)
            bound = float(np.abs(x).max()) * 2.0 ** -(61 - int(np.ceil(np.log2(max(count, 1))))) * max(counts.max(), 1)
            assert np.abs(sums - want).max() <= bound + 1e-300, (tag, count, np.abs(sums - want).max(), bound)
            d = (x - val).astype(np.float64)
            assert abs(miss - float((d * d).sum())) <= 1e-12 * max(float((d * d).sum()), 1e-300), (tag, count)


def test_stats_do_not_depend_on_the_run(cbm):
    """Counts, sums and squared miss are bit-equal from run to run (integer accumulation and fixed trees: no floating-point
    atomics anywhere); for another ORDER of the same data the counts are equal and the sums agree to the fixed-point
    resolution (a workgroup accumulates at the scale of the largest element it has seen, so the roundings differ)."""
    x = synth.make_samples(300000, 51)
    g = grid.TableGrid.nf4()
    a = raw_stats(torch.as_tensor(x, device="cuda"), table_values=g.values, table_limits=g.limits)
    for _ in range(3):
        b = raw_stats(torch.as_tensor(x, device="cuda"), table_values=g.values, table_limits=g.limits)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2]
    perm = np.random.RandomState(3).permutation(x.size)
    c = raw_stats(torch.as_tensor(x[perm], device="cuda"), table_values=g.values, table_limits=g.limits)
    s = raw_stats(torch.as_tensor(np.sort(x), device="cuda"), table_values=g.values, table_limits=g.limits)
    resolution = float(np.abs(x).max()) * 2.0 ** -(61 - 19)  # 300000 < 2^19 elements
    for other in (c, s):
        assert np.array_equal(a[0], other[0])
        assert np.abs(a[1] - other[1]).max() <= resolution * x.size
        assert abs(a[2] - other[2]) <= 1e-12 * a[2]


def test_stats_by_position_are_array_split_means(cbm):
    for count, parts in ((8, 4), (10, 4), (3, 8), (100003, 7), (4096, 256)):
        x = np.sort(synth.make_samples(count, 60))
        counts, sums, miss = raw_stats(torch.as_tensor(x, device="cuda"), by_position=1, levels=parts)
        pieces = np.array_split(x, parts)
        assert [len(p) for p in pieces] == list(counts) and miss == 0.0
        for k, p in enumerate(pieces):
            if len(p):
                assert abs(sums[k] - p.astype(np.float64).sum()) <= 1e-9 * np.abs(x).max() * len(p)


def test_sort_and_distinct(cbm):
    from sleekit_amd.codebook import _distinct, _sorted

    x = synth.make_samples(200001, 52)
    x[5], x[6], x[7], x[8] = np.inf, -np.inf, 0.0, -0.0
    x[100:200] = x[50]  # repeats
    got = _sorted(torch.as_tensor(x, device="cuda"))
    assert np.array_equal(got.cpu().numpy(), np.sort(x))
    assert np.array_equal(_distinct(got).cpu().numpy(), np.unique(x))
    empty = torch.empty(0, dtype=torch.float32, device="cuda")
    assert _sorted(empty).numel() == 0 and _distinct(empty).numel() == 0


# ------------------------------------------------------------------------------------------ the reference's methods
def test_shares_entropy_mse_centroids(cbm, codebook_fit):
    z = codebook_fit
    data = synth.make_samples(20000, 34, np.float32)
    nf4 = cbm.Codebook.nf4()
    x = data / 4
    assert np.array_equal(nf4.probabilities(x), z["nf4/shares"])          # integers / len
    assert nf4.entropy(x) == z["nf4/entropy"]                             # the same NumPy expression on the same shares
    mse = nf4.mse(x)
    assert mse.dtype == z["nf4/mse"].dtype and abs(float(mse) - float(z["nf4/mse"])) <= 1e-6 * float(z["nf4/mse"])
    c = nf4.centroids(x)
    assert c.dtype == z["nf4/centroids"].dtype
    close(c, z["nf4/centroids"], MEAN_TOL * float(np.abs(x).max()), "nf4 centroids")
    # device tensors in, the same numbers out
    xd = torch.as_tensor(x, device="cuda")
    assert np.array_equal(nf4.probabilities(xd), z["nf4/shares"]) and np.array_equal(nf4.centroids(xd), c)
    # float64 data: results in float64 like NumPy's (the device rounds the data to float32 once)
    c64 = nf4.centroids(x.astype(np.float64))
    assert c64.dtype == np.float64 and nf4.mse(x.astype(np.float64)).dtype == np.float64
    close(c64, fit.bin_centres(grid.TableGrid.nf4(), x.astype(np.float64)), MEAN_TOL * float(np.abs(x).max()), "float64 centroids")
    with pytest.raises(ValueError):
        nf4.probabilities(x.reshape(100, 200))


def test_empty_bins(cbm, codebook_fit):
    """codebook.py:224-230, 233-246: the three fall-backs of an empty bin (float32 arithmetic on the limits: exact) and
    the bins remove_unused drops."""
    z = codebook_fit
    data = synth.make_samples(20000, 34, np.float32)
    cb = cbm.Codebook([-50.0, -40.0, -0.5, 0.0, 0.25, 0.5, 30.0, 40.0, 50.0])
    c = cb.centroids(data)
    want = z["empty/centroids"]
    assert c.dtype == want.dtype
    for k in (0, 1, 6, 7, 8):
        assert c[k] == want[k], k
    close(c, want, MEAN_TOL * float(np.abs(data).max()), "centroids beside empty bins")
    cb.remove_unused(data)
    assert np.array_equal(cb.values, z["empty/kept_values"]) and np.array_equal(cb.thresholds, z["empty/kept_limits"])
    # the quantizing maps follow the new table
    g = grid.TableGrid(z["empty/kept_values"], z["empty/kept_limits"])
    assert np.array_equal(cb.quantize_index(data), g.index(data))


def test_equiprobable_like_the_reference_test(cbm):
    """tests/test_codebook.py:60-64 of the reference."""
    values = np.array([1, 2, 4, 5, 7, 8, 10, 11], dtype=np.float32)
    cb = cbm.Codebook.equiprobable(values, 4)
    assert np.allclose(cb.values, [1.5, 4.5, 7.5, 10.5])
    assert np.allclose(cb.thresholds, [3, 6, 9])
    # fewer points than codewords: the empty parts are dropped (codebook.py:328)
    cb = cbm.Codebook.equiprobable(np.float32([3.0, 1.0, 2.0]), 8)
    want = fit.equal_mass(np.float32([3.0, 1.0, 2.0]), 8)
    assert np.array_equal(cb.values, want.values) and np.array_equal(cb.thresholds, want.limits)


def test_lloyd_like_the_reference_tests(cbm):
    """tests/test_codebook.py:67-86 of the reference."""
    data = np.random.RandomState(0).randn(1000)
    for kw in (dict(), dict(lagrange_mult=0.01), dict(random_init=True)):
        cb = cbm.lloyd_max(data, 8, **kw)
        assert len(cb.values) == 8 and len(cb.thresholds) == 7
        cb.check()


def test_fit_against_the_reference(cbm, codebook_fit):
    z = codebook_fit
    worst = {}
    for name, count, seed, dtype, size, lam in json.loads(str(z["cases"])):
        data = synth.make_samples(count, seed, np.dtype(dtype).type)
        rng = span(data)
        start = cbm.Codebook.equiprobable(data, size)
        if dtype == "float32":
            assert np.array_equal(start.thresholds, z[f"{name}/start_limits"]), name  # float32 arithmetic on data elements
        close(start.thresholds, z[f"{name}/start_limits"], FIT_TOL * rng, name + " start limits")
        close(start.values, z[f"{name}/start_values"], FIT_TOL * rng, name + " start values")
        for k in (1, 2, 3):
            cb = cbm.lloyd_max(data, size, lam, max_iter=k)
            assert cb.values.dtype == z[f"{name}/round{k}_values"].dtype and cb.thresholds.dtype == z[f"{name}/round{k}_limits"].dtype
            a = close(cb.values, z[f"{name}/round{k}_values"], FIT_TOL * rng, f"{name} round {k} values")
            b = close(cb.thresholds, z[f"{name}/round{k}_limits"], FIT_TOL * rng, f"{name} round {k} limits")
            worst[name] = max(worst.get(name, 0.0), a / rng, b / rng)
        cb = cbm.lloyd_max(data, size, lam)
        assert len(cb) == len(z[f"{name}/final_values"]), name
        a = close(cb.values, z[f"{name}/final_values"], FINAL_TOL * rng, name + " final values")
        b = close(cb.thresholds, z[f"{name}/final_limits"], FINAL_TOL * rng, name + " final limits")
        close(cb.probabilities(data), z[f"{name}/final_shares"], 5.0 / count, name + " shares")
        assert abs(cb.entropy(data) - float(z[f"{name}/final_entropy"])) <= 1e-3
        assert abs(float(cb.mse(data)) - float(z[f"{name}/final_mse"])) <= 1e-4 * float(z[f"{name}/final_mse"])
        worst[name + " final"] = max(a, b) / rng
    print("worst deviation in units of the value range:", {k: f"{v:.1e}" for k, v in worst.items()})


def test_drawn_starts_use_numpys_generator_like_the_reference(cbm, codebook_fit):
    z = codebook_fit
    data = synth.make_samples(20000, 34, np.float32)
    np.random.seed(11)
    cb = cbm.Codebook.random(data, 8)
    after = np.random.rand()
    assert np.array_equal(cb.values, z["random/values"]) and np.array_equal(cb.thresholds, z["random/limits"])  # data values: exact
    np.random.seed(11)
    fit.pick_random(data, 8)
    assert after == np.random.rand()  # the generator was advanced exactly as the reference advances it
    np.random.seed(12)
    cb = cbm.lloyd_max(data, 8, random_init=True, sample_count=500)
    close(cb.values, z["random_fit/values"], FINAL_TOL * span(data), "fit from a drawn start")
    close(cb.thresholds, z["random_fit/limits"], FINAL_TOL * span(data), "fit from a drawn start")


def test_fit_at_full_size_is_a_lloyd_fixed_point(cbm):
    """16.7M weights (a 4096 x 4096 layer's worth): properties that do not need the CPU -- every value is the mean of
    its bin, every limit the midpoint of its values, the mse never rises from round to round, the shares sum to 1."""
    torch.manual_seed(5)
    x = torch.randn(4096 * 4096, device="cuda")
    x[::97] *= 4
    import time

    prev = None
    for k in (1, 2, 4, 8):
        cb = cbm.lloyd_max(x, 16, max_iter=k)
        mse = float(cb.mse(x))
        assert prev is None or mse <= prev * (1 + 1e-6), (k, mse, prev)
        prev = mse
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    cb = cbm.lloyd_max(x, 16)
    torch.cuda.synchronize()
    seconds = time.perf_counter() - t0
    assert float(cb.mse(x)) <= prev * (1 + 1e-6)
    shares = cb.probabilities(x)
    assert abs(shares.sum() - 1.0) < 1e-12 and (shares > 0).all()
    limits = torch.as_tensor(cb.thresholds.astype(np.float32), device="cuda")
    which = torch.bucketize(x, limits, right=True)
    counts = torch.bincount(which, minlength=16)
    assert np.array_equal(counts.cpu().numpy(), np.rint(shares * x.numel()).astype(np.int64))
    sums = torch.zeros(16, dtype=torch.float64, device="cuda").index_add_(0, which, x.double())
    means = (sums / counts).cpu().numpy()
    rng = span(cb.values)
    nxt = cb.clone()
    nxt.improve(x)
    assert float(nxt.mse(x)) <= float(cb.mse(x)) * (1 + 1e-6), "one more round cannot raise the mse"
    assert np.array_equal(cb.values, cb.centroids(x)), "the statistics pass is deterministic: the same means again"
    close(cb.values, means, MEAN_TOL * float(x.abs().max()), "values are bin means")
    close(nxt.thresholds, (cb.values[:-1] + cb.values[1:]) / 2, 1e-6 * rng, "limits are midpoints")
    print(f"lloyd_max on {x.numel()} samples, 16 levels: {seconds:.3f} s")


def test_argument_errors(cbm):
    from sleekit_amd import _device as dev
    from sleekit_amd import _lib

    x = torch.zeros(1024, device="cuda")
    counts = torch.empty(300, dtype=torch.int64, device="cuda")
    sums = torch.empty(300, dtype=torch.float64, device="cuda")
    ws = torch.empty(int(_lib.lib.slk_codebook_stats_workspace_bytes()), dtype=torch.uint8, device="cuda")
    call = lambda levels, lo, hi, nb: _lib.lib.slk_codebook_stats(dev.ptr(x), 1024, levels, lo, hi, None, 0, dev.ptr(counts), dev.ptr(sums), None,
                                                                  dev.ptr(ws), nb, dev.stream_handle())
    assert call(257, -1.0, 1.0, ws.numel()) == -1   # SLK_E_ARG
    assert call(8, 1.0, -1.0, ws.numel()) == -1
    assert call(8, -1.0, 1.0, 16) == -4             # SLK_E_WS
    assert call(8, -1.0, 1.0, ws.numel()) == 0      # sqerr may be NULL
    out = torch.empty_like(x)
    assert _lib.lib.slk_sort_f32(dev.ptr(x), 1024, dev.ptr(x), dev.ptr(ws), ws.numel(), dev.stream_handle()) == -1   # aliased
    assert _lib.lib.slk_sort_f32(dev.ptr(x), 1024, dev.ptr(out), dev.ptr(ws), 8, dev.stream_handle()) == -4
    with pytest.raises(NotImplementedError):
        cbm.Codebook(np.linspace(-1, 1, 300)).centroids(x)
    torch.cuda.synchronize()
