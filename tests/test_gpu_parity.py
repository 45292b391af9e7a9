"""Parity of the HIP path (through the C ABI) with the CPU oracle and the golden fixtures.

Run on the MI355X box:  python -m pytest tests -m gpu -x -q
Integer/index results are compared bit-for-bit; float64 factors to 1e-9 relative (they
need not match LAPACK bit-wise: SURVEY.md 8a6); layer errors to 1e-5 relative (north_star).
"""

import hashlib
import os
import sys

import numpy as np
import pytest
import torch

from conftest import parse_case
from ls_evidence import explain_rows
from oracle import grid, npsum, obq_ref, scaling_ref, stats_ref
from sleekit_amd import synth

pytestmark = pytest.mark.gpu

_layers = {}


def layer(R, n, seed, **kw):
    key = (R, n, seed, tuple(sorted(kw.items())))
    if key not in _layers:
        _layers[key] = synth.make_layer(R, n, seed, **kw)
    return _layers[key]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def amd():
    assert torch.cuda.is_available(), "these tests need the GPU"
    from sleekit_amd import _lib, codebook, engine, obq, scaling, statistics

    class NS:
        pass

    ns = NS()
    ns.lib, ns.codebook, ns.engine, ns.obq, ns.scaling, ns.statistics = _lib, codebook, engine, obq, scaling, statistics
    return ns


def run_product(amd, L, c):
    cb = amd.codebook.UniformCodebook(c["levels"], -1, 1)
    H = amd.obq.remove_input_bias(L["H"], L["mean"]) if c["strip"] else L["H"]
    out = amd.scaling.quantize_with_scaling(L["W"], L["scale"], cb, H, c["order"], c["damp"], c["moves"])
    idx = cb.quantize_index(amd.scaling.apply_scaling(out, L["scale"], 0))
    return out, idx, amd.obq.channelwise_error(L["W"], out, H), amd.obq.quantization_error(L["W"], out, H)


# --------------------------------------------------------------------------- element-wise stages
@pytest.mark.parametrize("levels", [2, 3, 4, 8, 16, 256, "asym"])
def test_codebook_maps_bit_exact(amd, pieces, levels):
    x = pieces["cb/x"]
    cb = amd.codebook.UniformCodebook(5, -0.75, 1.25) if levels == "asym" else amd.codebook.UniformCodebook(levels, -1, 1)
    tag = "cb/asym" if levels == "asym" else f"cb/N{levels}"
    for name in ("value", "index", "up", "down"):
        got = getattr(cb, "quantize_" + name)(x)
        want = pieces[f"{tag}/{name}"]
        assert got.dtype == want.dtype and np.array_equal(got, want), (levels, name)


@pytest.mark.parametrize("levels", [257, 1024, 65536, 70000])
def test_codebooks_above_256_entries(amd, pieces, levels):
    """codebook.py:50-54: indices widen to uint16 / uint32; the maps and a whole GPTQ layer (values; indices from the map)
    against the oracle, bit for bit."""
    x = pieces["cb/x"]
    cb, g = amd.codebook.UniformCodebook(levels, -1, 1), grid.UniformGrid(levels, -1, 1)
    for name, fn in (("value", g.value), ("index", g.index), ("up", g.up), ("down", g.down)):
        got, want = getattr(cb, "quantize_" + name)(x), fn(x.copy())
        assert got.dtype == want.dtype and np.array_equal(got, want), (levels, name)
    if levels <= 1024:
        L = layer(64, 96, 2001)
        want = scaling_ref.quantize_scaled(L["W"], L["scale"], g, L["H"], "diag", 0.01, 3 if levels == 1024 else 0)
        got = amd.scaling.quantize_with_scaling(L["W"], L["scale"], cb, L["H"], nb_ls_moves=3 if levels == 1024 else 0)
        assert np.array_equal(got, want)
        assert np.array_equal(cb.quantize_index(amd.scaling.apply_scaling(got, L["scale"], 0)),
                              g.index(scaling_ref.divide_rows(want, L["scale"], 0)))


def test_float32_divide_is_ieee(amd):
    """apply_scaling must be a correctly rounded divide, incl. subnormal results (scaling.py:21-25, 80)."""
    rng = np.random.default_rng(3)
    W = (rng.standard_normal((257, 1031)) * np.exp(rng.uniform(-30, 30, (257, 1031)))).astype(np.float32)
    W[0, :8] = [0.0, -0.0, 1e-38, -1e-38, 1e-45, 3.4e38, 1.0, -1.0]
    sc = np.exp(rng.uniform(-20, 20, 257)).astype(np.float32)
    got = amd.scaling.apply_scaling(W, sc, 0)
    assert np.array_equal(got, scaling_ref.divide_rows(W, sc, 0))
    back = amd.engine.rows_divide(torch.from_numpy(got).cuda(), torch.from_numpy(sc).cuda(), invert=True).cpu().numpy()
    assert np.array_equal(back, scaling_ref.divide_rows(got, 1 / sc, 0))


def test_scale_helpers_and_rtn(amd, pieces):
    L = layer(64, 96, 2001)
    cb = amd.codebook.UniformCodebook(8, -1, 1)
    assert np.array_equal(amd.scaling.apply_scaling(L["W"], L["scale"], 0), pieces["scale/apply"])
    assert np.array_equal(amd.scaling.quantize_with_scaling(L["W"], L["scale"], cb), pieces["scale/rtn"])
    W2 = L["W"].copy()
    amd.scaling.apply_scaling_in_place(W2, L["scale"], 0)
    assert np.array_equal(W2, pieces["scale/apply"])


def test_scaling_along_any_axis(amd):
    """The reference's own known-answer tests for the axis argument (tests/test_scaling.py:16-72), plus an
    apply / in-place round trip on a 4-D array against NumPy's broadcasting."""
    sc_mod = amd.scaling
    data = np.array([[0.0, 10.0], [5.0, 5.0]], dtype=np.float32)
    sc = sc_mod.compute_norm_scaling(data, 0)
    assert np.allclose(sc, [10.0 / np.sqrt(2), 5.0])
    scaled = sc_mod.apply_scaling(data, sc, 0)
    assert np.allclose(scaled, [[0.0, np.sqrt(2)], [1.0, 1.0]]) and np.allclose(sc_mod.apply_scaling(scaled, 1 / sc, 0), data)
    sc = sc_mod.compute_norm_scaling(data, 1)
    assert np.allclose(sc, [5.0 / np.sqrt(2), np.sqrt(125 / 2)])
    scaled = sc_mod.apply_scaling(data, sc, 1)
    assert np.allclose(scaled, [[0.0, 10.0 / np.sqrt(125 / 2)], [np.sqrt(2), 5.0 / np.sqrt(125 / 2)]])
    assert np.allclose(sc_mod.apply_scaling(scaled, 1 / sc, 1), data)
    rng = np.random.default_rng(8)
    big = rng.standard_normal((10, 20, 30, 40)).astype(np.float32)
    for axis in range(4):
        sc = sc_mod.compute_norm_scaling(big, axis)
        rest = tuple(i for i in range(4) if i != axis)
        assert len(sc) == big.shape[axis] and np.allclose(sc, np.sqrt(np.square(big).mean(axis=rest)), rtol=2e-5)
        shape = [1, 1, 1, 1]
        shape[axis] = -1
        assert np.array_equal(sc_mod.apply_scaling(big, sc, axis), big / sc.reshape(shape))
        copy = big.copy()
        sc_mod.apply_scaling_in_place(copy, sc, axis)
        assert np.array_equal(copy, big / sc.reshape(shape))
    table = np.array([[0.0, 10.0, -20.0, 15.0], [5.0, 5.0, 10.0, -10.0], [1.0, 2.0, -4.0, 3.0], [0.0, 0.0, 0.0, 0.0],
                      [1.0, 10.0, 100.0, 1000.0], [-1.0, 10.0, 100.0, 1000.0]], dtype=np.float32)
    cb = amd.codebook.Codebook([-1.0, 0.0, 10.0, 20.0])
    assert np.allclose(sc_mod.compute_non_saturating_scaling(table, cb, 0), [20, 10, 4, 1e-16, 50, 50])
    assert np.allclose(sc_mod.compute_non_saturating_scaling(table, cb, 1), [1, 0.5, 20, 50])
    # quantization along axis 1 == along axis 0 of the transpose
    ucb = amd.codebook.UniformCodebook(8, -1, 1)
    W = (rng.standard_normal((24, 40)) * 0.1).astype(np.float32)
    assert np.array_equal(sc_mod.compute_min_mse_scaling(W, ucb, axis=1, grid_size=20),
                          sc_mod.compute_min_mse_scaling(np.ascontiguousarray(W.T), ucb, axis=0, grid_size=20))


def test_scale_selection(amd, pieces):
    """closed-form scales and the grid searches, against the reference's outputs."""
    L = layer(64, 96, 2001)
    cb = amd.codebook.UniformCodebook(8, -1, 1)
    sc = amd.scaling
    assert np.array_equal(sc.compute_non_saturating_scaling(L["W"], cb, 0), pieces["scale/noclip"])
    assert np.array_equal(sc.compute_norm_scaling(L["W"], 0), pieces["scale/norm"])
    assert np.array_equal(sc.compute_scaling(L["W"], cb, L["H"], mode="max"), pieces["scale/noclip"])
    # H = None and diagonal H: float32 row sums in NumPy's order -> the same grid point, bit for bit
    for mode in ("mse", "diag", "diag3"):
        got = sc.compute_scaling(L["W"], cb, L["H"], mode=mode, grid_size=20)
        assert np.array_equal(got, pieces[f"scale/search_{mode}"]), mode
    # full Hessian / OBQ-aware: the row errors come out of a GEMM (the BLAS's summation order there, ours here), so a row
    # may take another grid point ONLY where the reference's own errors of the two points agree to within that rounding:
    # every differing row is checked against the reference's recorded per-point errors (pieces: scale/search_*_errors)
    for mode, gs in (("hessian", 20), ("hessian1", 20), ("obq", 10)):
        got = sc.compute_scaling(L["W"], cb, L["H"], mode=mode, grid_size=gs)
        want = pieces[f"scale/search_{mode}"]
        errs, factors, base = (pieces[f"scale/search_{mode}_{k}"] for k in ("errors", "factors", "base"))
        rows = np.flatnonzero(got != want)
        for r in rows:
            picked = np.flatnonzero(base[r] * factors == got[r])
            assert len(picked) == 1, (mode, r, "not a grid point")
            e_dev, e_ref = float(errs[picked[0], r]), float(errs[:, r].min())
            assert e_dev - e_ref <= 4e-6 * e_ref, (mode, int(r), e_dev, e_ref)  # a few float32 roundings of a sum of n^2 terms
        print(mode, "rows on another grid point:", len(rows), "of", len(want))
    with pytest.raises(RuntimeError):
        sc.compute_scaling(L["W"], cb, L["H"], mode="bogus")
    with pytest.raises(RuntimeError):
        sc.compute_non_saturating_scaling(L["W"], amd.codebook.UniformCodebook(4, 0.0, 1.0), 0)


@pytest.mark.parametrize("R,n", [(33, 1100), (16, 8200), (8, 11008)])
def test_scale_search_row_sums_follow_numpy(amd, R, n):
    """Rows longer than one pairwise block / one 8192-element chunk: same sums, same choices as NumPy."""
    g = grid.UniformGrid(4, -1, 1)
    cb = amd.codebook.UniformCodebook(4, -1, 1)
    W = synth.make_weights(R, n, 4000 + n)
    hd = (np.abs(synth.normal_grid(4000 + n, 8, 1, n)[0]) * 3 + 0.1).astype(np.float32)
    assert np.array_equal(amd.scaling.compute_norm_scaling(W, 0), scaling_ref.norm_scale(W, 0))
    assert np.array_equal(amd.scaling.compute_non_saturating_scaling(W, cb, 0), scaling_ref.no_clip_scale(W, g, 0))
    assert np.array_equal(amd.scaling.compute_min_mse_scaling(W, cb, grid_size=30), scaling_ref.best_grid_scale(W, g, grid_size=30))
    assert np.array_equal(amd.scaling.compute_min_mse_scaling(W, cb, H=hd, grid_size=30),
                          scaling_ref.best_grid_scale(W, g, H=hd, grid_size=30))


def test_dead_columns_and_mean_removal(amd, pieces):
    L = layer(32, 64, 2020, dead=(3, 17, 40))
    H, W = L["H"].copy(), L["W"].copy()
    amd.obq.remove_dead_values(H, W)
    assert np.array_equal(H, pieces["dead/H"]) and np.array_equal(W, pieces["dead/W"])
    assert np.array_equal(amd.obq.remove_input_bias(L["H"], L["mean"]), pieces["strip/H"])
    cb = amd.codebook.UniformCodebook(8, -1, 1)
    out = amd.scaling.quantize_with_scaling(W, L["scale"], cb, H)
    assert np.array_equal(cb.quantize_index(amd.scaling.apply_scaling(out, L["scale"], 0)), pieces["dead/idx"])
    np.testing.assert_allclose(amd.obq.quantization_error(W, out, H), pieces["dead/err"], rtol=1e-5)


@pytest.mark.parametrize("n", [1, 7, 8, 9, 127, 128, 129, 768, 1100, 4096, 8192, 8200, 11008])
def test_diag_mean_follows_numpy_order(amd, n):
    rng = np.random.default_rng(n)
    d = (np.square(rng.standard_normal(n)) * 3).astype(np.float32)
    H = torch.zeros((n, n), dtype=torch.float32, device="cuda")
    H.diagonal().copy_(torch.from_numpy(d))
    out = torch.empty(1, dtype=torch.float32, device="cuda")
    amd.lib.check(amd.lib.lib.slk_diag_mean(H.data_ptr(), n, out.data_ptr(), None, 0, None))
    assert np.float32(out.item()) == npsum.mean_f32(d) == d.mean()


def test_device_generator_makes_the_same_bytes():
    """bench.py sets its layers up with synth.make_layer_device (integer hashing and exact float64 arithmetic in
    torch on the GPU): the bytes must be those of the host generator the fixtures were made from."""
    for R, n, seed in ((64, 96, 2001), (96, 172, 2003), (256, 768, 2010), (40, 1100, 77)):
        host = layer(R, n, seed)
        got = synth.make_layer_device(R, n, seed, torch.device("cuda"), chunk=512)
        for k in ("W", "H", "mean", "scale"):
            assert np.array_equal(got[k].cpu().numpy(), host[k]), (R, n, k)


# --------------------------------------------------------------------------- order + factor
@pytest.mark.parametrize("R,n,seed", [(8, 16, 2000), (64, 96, 2001), (96, 172, 2003), (256, 768, 2010), (8, 1100, 2040)])
def test_order_and_factor(amd, R, n, seed):
    L = layer(R, n, seed)
    H = L["H"]
    Hd = H + 0.01 * H.diagonal().mean() * np.eye(n)
    want_order = obq_ref.column_order(L["W"], Hd, None, "diag")
    order, U, info = amd.engine.factorize(torch.from_numpy(H).cuda(), n, 0.01, amd.lib.ORDER_DIAG)
    assert int(info.item()) == 0
    assert np.array_equal(order.cpu().numpy(), want_order)
    want_U = obq_ref.inverse_factor_upper(Hd[want_order][:, want_order])
    got_U = U.cpu().numpy()
    assert np.array_equal(got_U, np.triu(got_U))
    scale = np.abs(want_U).max()
    assert np.abs(got_U - np.triu(want_U)).max() <= 1e-9 * scale
    # defining property, independent of LAPACK: U^T U = Hd[order][:, order]^-1
    P = Hd[want_order][:, want_order]
    assert np.abs(got_U.T @ got_U @ P - np.eye(n)).max() < 1e-8


def test_factor_lookahead_changes_nothing(amd):
    """The single-layer API lets the factorisation look ahead (the bulk of every outer update on a helper stream, beside the
    next block's panels): every tile still receives the same updates in the same order, so U is the same bit for bit."""
    n = 2560  # 40 tile rows: outer blocks with 36, 32, 28 and 24 trailing rows fork
    L = synth.make_layer_device(8, n, 4100, torch.device("cuda"))
    o0, U0, i0 = amd.engine.factorize(L["H"], n, 0.01, amd.lib.ORDER_DIAG)
    o1, U1, i1 = amd.engine.factorize(L["H"], n, 0.01, amd.lib.ORDER_DIAG, lookahead=True)
    torch.cuda.synchronize()
    assert amd.lib.lib.slk_get_option(b"lookahead") == 0  # (an argument of the call: no process-wide switch is touched)
    assert torch.equal(o0, o1) and torch.equal(U0, U1) and int(i0.item()) == int(i1.item()) == 0
    # the helper stream and its events go when asked to, and come back at the next use
    from sleekit_amd import _device as sdev

    sdev.release_workspaces()
    o2, U2, i2 = amd.engine.factorize(L["H"], n, 0.01, amd.lib.ORDER_DIAG, lookahead=True)
    torch.cuda.synchronize()
    assert torch.equal(U0, U2)


@pytest.mark.parametrize("n", [64, 100, 128, 320, 768, 1100, 2048, 4096, 4700])
def test_factor_panel_step_in_two_launches_is_the_one_launch_form(amd, n):
    """Three forms of the factorisation's panel step, one arithmetic: the CHAIN (round 4, the default: an outer block's panels
    in one launch of workgroups that hand the panels on through flags, then one wide launch for the rows below), the panel
    kernel in two launches (diagonal tile by one workgroup, then the tiles below it) and in one (option panel_split = 3 | 1 |
    2): same blocks, same products, same order -- the same U, order and status bit for bit, one matrix or a batch, positive
    definite or not.  (4096 and 4700 columns: outer blocks of 512 columns = chains of 8 workgroups, a short last block.)"""
    dev_ = torch.device("cuda")
    Hs = [synth.make_layer_device(8, n, 4200 + b, dev_)["H"] for b in range(3)]
    bad = Hs[1].clone()
    bad[n // 2, n // 2] = -1.0  # not positive definite: the status word names the same pivot either way
    forms = {}
    for form in (1, 2, 3, 0):
        with amd.lib.option("panel_split", form):
            forms[form] = ([amd.engine.factorize(H, n, 0.01, amd.lib.ORDER_DIAG) for H in Hs + [bad]],
                           amd.engine.factorize_batch(Hs, n, 0.01, amd.lib.ORDER_DIAG),
                           amd.engine.factorize(Hs[0], n, 0.01, amd.lib.ORDER_DIAG, lookahead=True))
    torch.cuda.synchronize()
    two, two_b, _ = forms[1]
    for form in (2, 3, 0):
        one, one_b, ahead = forms[form]
        for (o2, U2, i2), (o1, U1, i1) in zip(two, one):
            assert torch.equal(o1, o2) and int(i1.item()) == int(i2.item()), form
            if int(i1.item()) == 0:
                assert torch.equal(U1, U2), form
        for x, y in zip(two_b, one_b):
            assert torch.equal(x, y), form
        assert torch.equal(ahead[1], two[0][1]) and int(ahead[2].item()) == 0, form
    assert int(two[3][2].item()) != 0
    for b in range(3):
        assert torch.equal(two_b[1][b], two[b][1])
    # the rows below a diagonal block: 64 rows per workgroup (k_chol_chain<true>, the default beside other layers' kernels)
    # and 16 (k_chol_rows_below, the default of a factorisation that looks ahead) -- forced either way, same U
    for wide in (1, 2):
        with amd.lib.option("rows_below_wide", wide):
            o, U, i = amd.engine.factorize(Hs[2], n, 0.01, amd.lib.ORDER_DIAG)
            ob, Ub, ib = amd.engine.factorize_batch(Hs, n, 0.01, amd.lib.ORDER_DIAG)
        assert torch.equal(U, two[2][1]) and int(i.item()) == 0 and torch.equal(Ub, two_b[1]), wide


def test_chain_handoffs_under_uneven_load(amd):
    """The chain's hand-offs (flags in memory, write-through stores, sc1 loads) with the chip busy and UNEVENLY so -- six
    streams factor matrices of four widths over and over while two more run layer-error products and loops: every U must be
    the one the same call gives alone (the CDNA4 guide: idle chips and uniform load hide stale reads)."""
    dev_ = torch.device("cuda")
    widths = [768, 1024, 1100, 2048, 3072, 4096]
    Hs = [synth.make_layer_device(8, n, 4300 + i, dev_)["H"] for i, n in enumerate(widths)]
    alone = []
    for H, n in zip(Hs, widths):
        alone.append(amd.engine.factorize(H, n, 0.01, amd.lib.ORDER_DIAG))
        torch.cuda.synchronize()
    batch_alone = amd.engine.factorize_batch([Hs[0]] * 5, 768, 0.01, amd.lib.ORDER_DIAG)
    torch.cuda.synchronize()
    L = synth.make_layer_device(2048, 2048, 4310, dev_)
    cb = amd.codebook.UniformCodebook(8, -1, 1)
    streams = [torch.cuda.Stream() for _ in range(8)]
    got = [[] for _ in widths]
    got_b = []
    for rep in range(6):
        for i, (H, n) in enumerate(zip(Hs, widths)):
            with torch.cuda.stream(streams[(i + rep) % 6]):
                got[i].append(amd.engine.factorize(H, n, 0.01, amd.lib.ORDER_DIAG))
        with torch.cuda.stream(streams[rep % 6]):
            got_b.append(amd.engine.factorize_batch([Hs[0]] * 5, 768, 0.01, amd.lib.ORDER_DIAG))
        with torch.cuda.stream(streams[6]):
            res = amd.engine.quantize_layer(L["W"], L["H"], cb, L["scale"], lookahead=False)
        with torch.cuda.stream(streams[7]):
            amd.engine.row_errors(L["W"], L["W"] * 0.99, L["H"])
    torch.cuda.synchronize()
    for i in range(len(widths)):
        for o, U, info in got[i]:
            assert int(info.item()) == 0 and torch.equal(o, alone[i][0]) and torch.equal(U, alone[i][1]), widths[i]
    for o, U, info in got_b:
        assert torch.equal(U, batch_alone[1]) and int(info.abs().sum().item()) == 0
    assert res.Q is not None


def test_factor_of_plain_matrix_and_not_pd(amd):
    rng = np.random.default_rng(11)
    A = rng.standard_normal((200, 150))
    M = A.T @ A + 0.5 * np.eye(150)  # asymmetric-looking input is fine: only one triangle is read
    U = amd.obq.compute_hessian_chol(M)
    want = obq_ref.inverse_factor_upper(M)
    assert np.abs(U - np.triu(want)).max() <= 1e-10 * np.abs(want).max()
    M[70, 70] = -1.0
    with pytest.raises(np.linalg.LinAlgError):
        amd.obq.compute_hessian_chol(M)
    with pytest.raises(np.linalg.LinAlgError):
        obq_ref.inverse_factor_upper(M)


@pytest.mark.parametrize("n", [96, 1024])
def test_layer_with_an_indefinite_hessian_raises_like_the_reference(amd, n):
    """sleekit/obq.py:49-50: np.linalg.cholesky raises LinAlgError for a Hessian that is not positive definite.  The single-layer
    API reads the factorisation's status word back AFTER the loop (and the search) are enqueued -- the round trip used to hold
    up the loop's first launch -- so the loop runs on a void factor before the exception comes: still the reference's
    exception, nothing returned, and the next layer is unaffected."""
    L = layer(40, n, 2070 + n)
    cb = amd.codebook.UniformCodebook(8, -1, 1)
    good = amd.scaling.quantize_with_scaling(L["W"], L["scale"], cb, L["H"], act_order="diag", damp=0.01, nb_ls_moves=3)
    H = L["H"].copy()
    k = n // 2
    H[k, :] = 0
    H[:, k] = 0
    H[k, k] = -10.0 * np.abs(L["H"]).max()  # stays negative under 1 % damping
    with pytest.raises(np.linalg.LinAlgError):
        amd.scaling.quantize_with_scaling(L["W"], L["scale"], cb, H, act_order="diag", damp=0.01, nb_ls_moves=3)
    with pytest.raises(np.linalg.LinAlgError):
        amd.obq.quantize_opt(L["W"], H, cb)
    again = amd.scaling.quantize_with_scaling(L["W"], L["scale"], cb, L["H"], act_order="diag", damp=0.01, nb_ls_moves=3)
    assert np.array_equal(good, again)


def test_orders_err_sqerr(amd):
    L = layer(64, 96, 2001)
    g = grid.UniformGrid(8, -1, 1)
    cb = amd.codebook.UniformCodebook(8, -1, 1)
    Ws = scaling_ref.divide_rows(L["W"], L["scale"], 0)
    for mode, squared in (("err", False), ("sqerr", True)):
        miss = amd.engine.column_miss(torch.from_numpy(Ws).cuda(), cb._abi(), squared).cpu().numpy()
        d = g(Ws) - Ws
        want = (np.square(d) if squared else np.abs(d)).sum(axis=0)
        assert np.array_equal(miss, want), mode


# --------------------------------------------------------------------------- the loop
def test_loop_trace_bit_exact(amd, pieces):
    """Q and E after the blocked loop with the reference's own factor: isolates the loop kernels."""
    L = layer(8, 16, 2000)
    cb = amd.codebook.UniformCodebook(8, -1, 1)
    order = pieces["trace/order"]
    Ws = scaling_ref.divide_rows(L["W"], L["scale"], 0)
    Q = Ws[:, order].copy()
    E = np.zeros_like(Q)
    amd.obq._quantize_opt_block(Q, E, pieces["trace/U"], cb, 4, 2)
    assert np.array_equal(Q, pieces["trace/Q"])
    assert np.array_equal(E, pieces["trace/E"])


@pytest.mark.parametrize("min_block,num_blocks", [(1, 2), (3, 2), (4, 4), (7, 2), (8, 4), (63, 2), (64, 4), (32, 8), (1000, 8)])
def test_loop_blockings_match_oracle(amd, min_block, num_blocks):
    """Every recursion shape the reference's own test sweeps (tests/test_obq.py:57-70), bit for bit."""
    L = layer(96, 172, 2003)
    g = grid.UniformGrid(8, -1, 1)
    cb = amd.codebook.UniformCodebook(8, -1, 1)
    Ws = scaling_ref.divide_rows(L["W"], L["scale"], 0)
    Hd = L["H"] + 0.01 * L["H"].diagonal().mean() * np.eye(172)
    U = obq_ref.inverse_factor_upper(Hd)
    Q0, E0 = Ws.copy(), np.zeros_like(Ws)
    obq_ref.run_schedule(Q0, E0, U, g, obq_ref.block_schedule(172, min_block, num_blocks))
    Q1, E1 = Ws.copy(), np.zeros_like(Ws)
    amd.obq._quantize_opt_block(Q1, E1, U, cb, min_block, num_blocks)
    assert np.array_equal(Q1, Q0)
    np.testing.assert_allclose(E1, E0, rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("n", [96, 768, 1024, 1376, 3072, 4096])
@pytest.mark.parametrize("odd_diagonal", [False, True])
def test_window_kernels_agree(amd, n, odd_diagonal, slkopt):
    """The standard-schedule window kernel (row-independent chain waves, 4x4x4 MFMA local updates,
    helper waves) against the general one (16-row MFMA, barriers), bit for bit on Q and E: every
    BASELINE width (one- and two-leaf periods, 32 + 16 and 5 x 32 + 12 leaves), a ragged last row
    tile, and a diagonal entry with an all-ones significand (exact-division exception: true divides)."""
    rng = np.random.default_rng(n)
    R = 40 if n != 1024 else 109  # ragged last tiles of 16 and of 32 rows
    W = (rng.standard_normal((R, n)) * 0.6).astype(np.float32)
    U = np.triu(rng.standard_normal((n, n)) * (0.3 / np.sqrt(n))) + np.diag(1.0 + rng.random(n))
    if odd_diagonal:
        U[n // 3, n // 3] = np.nextafter(2.0, 0.0)  # 1.111...1b: the fma division shortcut is not exact for it
    cb = amd.codebook.UniformCodebook(8, -1, 1)
    out = []
    for variant in ("rows16", "rows32", "general"):
        # rows16: four rows per chain wave (16 lanes x 2 columns); rows32: EIGHT rows per chain wave (8 lanes x 4 columns,
        # one quantizer stream for them) and single U buffers; general: 16-row MFMA, barrier per op
        if variant == "general":
            slkopt.setenv("SLK_NO_WINDOW2", "1")
        else:
            slkopt.delenv("SLK_NO_WINDOW2", raising=False)
            slkopt.setenv("SLK_WINDOW_ROWS", variant[4:])
        Q, E = W.copy(), np.zeros_like(W)
        amd.obq._quantize_opt_block(Q, E, U, cb, 32, 8)
        out.append((Q, E))
    slkopt.delenv("SLK_WINDOW_ROWS")
    for other in out[1:]:
        assert np.array_equal(out[0][0], other[0])
        assert np.array_equal(out[0][1], other[1])
    # and both against the oracle on the narrow cases (seconds on the CPU)
    if n <= 1024:
        Q0, E0 = W.copy(), np.zeros_like(W)
        obq_ref.run_schedule(Q0, E0, U, grid.UniformGrid(8, -1, 1), obq_ref.block_schedule(n, 32, 8))
        assert np.array_equal(out[0][0], Q0)
        np.testing.assert_allclose(out[0][1], E0, rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("levels,lo,hi", [(2, -1, 1), (3, -1, 1), (4, -1, 1), (8, -1, 1), (16, -1, 1), (256, -1, 1), (5, -0.75, 1.25), (7, 0.0, 3.0)])
def test_fast_quantizer_matches_true_divide(amd, levels, lo, hi):
    """The leaf kernel replaces the float32 divide by an exact-division fma sequence; with an
    identity factor the loop degenerates to q(W), which must equal the true-divide codebook
    kernel bit for bit -- on random values and on values a few ulps around every rounding tie."""
    cb = amd.codebook.UniformCodebook(levels, lo, hi)
    step = np.float32((hi - lo) / (levels - 1))
    rng = np.random.default_rng(levels)
    ties = (np.float32(lo) + (np.arange(-2, levels + 2, dtype=np.float32) + np.float32(0.5)) * step).astype(np.float32)
    near, up, down = [ties], ties, ties
    for _ in range(6):  # up to six ulps on either side of every tie (finite values only)
        up, down = np.nextafter(up, np.float32(np.inf)), np.nextafter(down, np.float32(-np.inf))
        near += [up.astype(np.float32), down.astype(np.float32)]
    near = np.concatenate(near)
    body = (rng.standard_normal(512 * 2048) * 0.7 * (hi - lo) + (hi + lo) / 2).astype(np.float32)
    W = np.concatenate([near, body])[: 512 * 2048].reshape(512, 2048).copy()
    W.reshape(-1)[: len(near)] = near[: W.size]
    Q, E = W.copy(), np.zeros_like(W)
    amd.obq._quantize_opt_block(Q, E, np.eye(2048), cb, 32, 8)
    want = cb.quantize_value(W)
    assert np.array_equal(Q, want)
    assert np.array_equal(E, W - want)


def test_small_cases_bit_exact(amd, small_cases):
    names = [str(x) for x in small_cases["names"]]
    bad = []
    for name in names:
        c = parse_case(name)
        L = layer(c["R"], c["n"], c["seed"])
        out, idx, rows, err = run_product(amd, L, c)
        if not np.array_equal(idx, small_cases[name + "/idx"]):
            bad.append((name, int((idx != small_cases[name + "/idx"]).sum())))
            continue
        want = float(small_cases[name + "/err"])
        assert abs(float(err) - want) <= 1e-5 * abs(want), (name, float(err), want)
        np.testing.assert_allclose(rows, small_cases[name + "/row_err"], rtol=2e-4, atol=1e-7, err_msg=name)
    assert not bad, bad


def test_quantize_opt_direct_and_device_tensors(amd):
    """quantize_opt on already scaled weights; device tensors in -> device tensors out, same bits."""
    L = layer(128, 256, 2002)
    g = grid.UniformGrid(4, -1, 1)
    cb = amd.codebook.UniformCodebook(4, -1, 1)
    Ws = scaling_ref.divide_rows(L["W"], L["scale"], 0)
    want = obq_ref.quantize_layer(Ws, L["H"], g, "diag", 0.01, 0)
    got = amd.obq.quantize_opt(Ws, L["H"], cb)
    assert got.dtype == np.float32 and np.array_equal(got, want)
    got_dev = amd.obq.quantize_opt(torch.from_numpy(Ws).cuda(), torch.from_numpy(L["H"]).cuda(), cb)
    assert got_dev.is_cuda and np.array_equal(got_dev.cpu().numpy(), want)
    with pytest.raises(RuntimeError):
        amd.obq.quantize_opt(Ws, L["H"], cb, act_order="bogus")
    with pytest.raises(AssertionError):
        amd.obq.quantize_opt(Ws, L["H"][:-1], cb)
    with pytest.raises(NotImplementedError):
        amd.obq.quantize_opt(Ws, L["H"], lambda x: np.round(x))


@pytest.mark.parametrize("order", ["inv_diag", "combined_diag"])
def test_orders_from_the_inverse_diagonal(amd, order, orders):
    """obq.py:70-75: orders that need diag(Hd^-1); here from a first factorisation in the original order.  Against the
    REFERENCE's order and output (tests/golden/orders.npz) and the oracle's."""
    for R, n, seed in ((64, 96, 2001), (96, 172, 2003)):
        L = layer(R, n, seed)
        g = grid.UniformGrid(8, -1, 1)
        cb = amd.codebook.UniformCodebook(8, -1, 1)
        want = scaling_ref.quantize_scaled(L["W"], L["scale"], g, L["H"], order, 0.01, 0)
        got = amd.scaling.quantize_with_scaling(L["W"], L["scale"], cb, L["H"], act_order=order, damp=0.01)
        assert np.array_equal(got, want), (order, n)
        tag = f"{order}/r{R}_n{n}_s{seed}"
        assert np.array_equal(got, orders[tag + "/out"]), tag


def test_experiment_scripts_replayed_over_the_dropin_package(experiments, tmp_path):
    """north_star: "drops in under experiments/*.py".  The scripts do `from sleekit.codebook import *` (+ .obq, .scaling;
    experiments/compare.py:1-3) and use `np` without importing it.  With dropin/ on the path those lines resolve to this
    build; the main loops of local_search / correction / ordering / dampening / bits / scaling / compare
    (tests/experiment_replays.py) over the star-imported names must print the layer errors the reference's own scripts
    printed for the same dumps (tests/golden/experiments.json)."""
    import experiment_replays

    dropin = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dropin")
    sys.path.insert(0, dropin)
    try:
        ns = {}
        exec("from sleekit.codebook import *\nfrom sleekit.obq import *\nfrom sleekit.scaling import *\n", ns)
        import sleekit

        assert sleekit.__file__.startswith(dropin) and ns["UniformCodebook"].__module__ == "sleekit_amd.codebook"
        experiment_replays.check_against_fixture(experiments, ns, str(tmp_path))
    finally:
        sys.path.remove(dropin)


@pytest.mark.parametrize("staged", [False, True])
def test_hessian_accumulate_bf16_path(amd, slkopt, staged):
    """statistics.py:76-87 with a feature count that is a multiple of 128: X^T X on the bfloat16 MFMA (three
    pieces per operand).  A ragged token count, two batches (running-mean factor), and a workspace so small
    that the tokens go in chunks of 32 -- each against float64, to float32 GEMM tolerance.  Both ways of bringing
    the operands to LDS: global_load_lds of swizzled planes (default) and staging through registers."""
    import torch
    if staged:
        slkopt.setenv("SLK_NO_BF16_DMA", "1")
    from sleekit_amd import _lib, _device as dev

    rng = np.random.default_rng(21)
    n = 256
    X1 = (rng.standard_normal((200, n)) * (0.5 + rng.random(n))).astype(np.float32)
    X2 = (rng.standard_normal((75, n)) + 0.3).astype(np.float32)
    want_H = (X1.astype(np.float64).T @ X1 + X2.astype(np.float64).T @ X2) / 275.0
    want_m = (X1.astype(np.float64).sum(0) + X2.astype(np.float64).sum(0)) / 275.0
    for ws_bytes in (None, 4096 + 6 * n * 40):  # full workspace; room for 32 tokens at a time
        H = torch.zeros((n, n), dtype=torch.float32, device="cuda")
        m = torch.zeros(n, dtype=torch.float32, device="cuda")
        ws, full = dev.workspace(0, n)
        count = 0
        for X in (X1, X2):
            Xd = torch.as_tensor(X, device="cuda")
            _lib.check(_lib.lib.slk_hessian_accumulate(H.data_ptr(), m.data_ptr(), Xd.data_ptr(), n, X.shape[0], count,
                                                       ws.data_ptr(), full if ws_bytes is None else ws_bytes, None))
            count += X.shape[0]
        torch.cuda.synchronize()
        Hh = H.cpu().numpy()
        assert np.array_equal(Hh, Hh.T)
        np.testing.assert_allclose(Hh, want_H, rtol=2e-5, atol=2e-6 * np.abs(want_H).max())
        np.testing.assert_allclose(m.cpu().numpy(), want_m, rtol=1e-5, atol=1e-6)


def test_integration_stub_runs(amd):
    """The ctypes binding printed in INTEGRATION.md is executed as it stands (library path aside) and must
    give what sleekit_amd.obq.quantize_opt gives."""
    import os
    import re
    import torch

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    code = next(b for b in re.findall(r"```python\n(.*?)```", text, re.S) if "quantize_opt_mi355x" in b)
    code = "\n".join(line[3:] if line.startswith("   ") else line for line in code.splitlines())
    code = code.replace('ctypes.CDLL("libsleekit_amd.so")', f'ctypes.CDLL("{os.path.join(root, "sleekit_amd", "libsleekit_amd.so")}")')
    ns = {"numpy": np}
    exec(compile(code, "INTEGRATION.md", "exec"), ns)
    L = layer(64, 96, 2001)
    cb = amd.codebook.UniformCodebook(8, -1, 1)
    Ws = scaling_ref.divide_rows(L["W"], L["scale"], 0)
    got = ns["quantize_opt_mi355x"](torch.from_numpy(Ws).cuda(), torch.from_numpy(L["H"]).cuda(), cb)
    assert np.array_equal(got.cpu().numpy(), amd.obq.quantize_opt(Ws, L["H"], cb))


def test_layer_error_on_tall_tiles_is_the_square_tile_kernel(amd, slkopt):
    """Whole layers (>= 2048 rows) CAN take 256 x 128 tiles on the bfloat16 MFMA (option tall_error; k_error_tiles_bf16_tall: two
    images of a round in LDS, half the workgroups); the default, smaller ones and K-chunked shards are 128 x 128.  Same slabs, same six products per element,
    same rounds: the row errors and the product G = (W - Q) H are the square-tile kernel's BIT FOR BIT -- symmetric H (half
    the products), an H that is not symmetric (averaged planes), the full product for the local search, and a stack of layers."""
    rng = np.random.default_rng(23)
    R, n = 2304, 1024  # nine row tiles of 256: the XCD-aware order leaves a last partial group
    W = torch.from_numpy(rng.standard_normal((R, n)).astype(np.float32)).cuda()
    Q = W + 0.2 * torch.from_numpy(rng.standard_normal((R, n)).astype(np.float32)).cuda()
    X = rng.standard_normal((2 * n, n)).astype(np.float32)
    H = (X.T @ X / (2 * n)).astype(np.float32)
    H = ((H + H.T) * np.float32(0.5)).astype(np.float32)
    Ha = (H + np.float32(0.05) * np.triu(rng.standard_normal((n, n)).astype(np.float32), 1)).astype(np.float32)
    Hs, Has = torch.from_numpy(H).cuda(), torch.from_numpy(Ha).cuda()

    def both(fn):
        square = fn()
        slkopt.setenv("SLK_TALL_ERROR", "1")
        tall = fn()
        slkopt.delenv("SLK_TALL_ERROR")
        return tall, square

    for Hx in (Hs, Has):
        tall, square = both(lambda: amd.engine.row_errors(W, Q, Hx))
        assert torch.equal(tall, square)
        (e1, G1), (e2, G2) = both(lambda: amd.engine.row_errors(W, Q, Hx, want_G=True))
        assert torch.equal(e1, e2) and torch.equal(G1, G2)
    D = (W - Q).double().cpu().numpy()
    want = ((D @ H.astype(np.float64)) * D).sum(axis=1)
    np.testing.assert_allclose(amd.engine.row_errors(W, Q, Hs).cpu().numpy(), want, rtol=1e-5)
    # a stack of three layers of 768 rows (a multiple of 256) with Hessians of their own
    H2 = torch.from_numpy(((H * np.float32(1.25)) + np.float32(0.01) * np.eye(n, dtype=np.float32))).cuda()
    Wb, Qb = W.view(3, 768, n).contiguous(), Q.view(3, 768, n).contiguous()
    tall, square = both(lambda: amd.engine.row_errors_batch(Wb, Qb, [Hs, H2, Has]))
    assert torch.equal(tall, square)
    one = amd.engine.row_errors(Wb[1].contiguous(), Qb[1].contiguous(), H2)
    np.testing.assert_allclose(tall[1].cpu().numpy(), one.cpu().numpy(), rtol=2e-6)


def test_layer_error_on_256_tiles(amd, slkopt):
    """Whole layers CAN take 256 x 256 tiles (option tall_error = 2; k_error_tiles_bf16_big: K in steps of 16 through a ring of
    three LDS images filled by global_load_lds, planes in the K16 layout).  Same six products per element and 16-k step in
    the same order: the product G = (W - Q) H is the square-tile kernel's BIT FOR BIT; a row's partial sums are added in
    another order (256 columns and two k blocks per slot), so the row errors agree to rounding.  Symmetric H, an H that is
    not symmetric (averaged planes without G, the float32 kernel with it), and a stack of layers."""
    rng = np.random.default_rng(29)
    for R, n in ((2304, 1024), (2048, 2304)):  # nine row tiles / nine column tiles: partial patches and XCDs without a row tile
        W = torch.from_numpy(rng.standard_normal((R, n)).astype(np.float32)).cuda()
        Q = W + 0.2 * torch.from_numpy(rng.standard_normal((R, n)).astype(np.float32)).cuda()
        X = rng.standard_normal((2 * n, n)).astype(np.float32)
        H = (X.T @ X / (2 * n)).astype(np.float32)
        H = ((H + H.T) * np.float32(0.5)).astype(np.float32)
        Ha = (H + np.float32(0.05) * np.triu(rng.standard_normal((n, n)).astype(np.float32), 1)).astype(np.float32)
        Hs, Has = torch.from_numpy(H).cuda(), torch.from_numpy(Ha).cuda()

        def both(fn):
            square = fn()
            slkopt.setenv("SLK_TALL_ERROR", "2")
            big = fn()
            slkopt.delenv("SLK_TALL_ERROR")
            return big, square

        for Hx in (Hs, Has):
            big, square = both(lambda: amd.engine.row_errors(W, Q, Hx))
            np.testing.assert_allclose(big.cpu().numpy(), square.cpu().numpy(), rtol=2e-6)
            (e1, G1), (e2, G2) = both(lambda: amd.engine.row_errors(W, Q, Hx, want_G=True))
            if Hx is Hs:
                assert torch.equal(G1, G2)
            else:  # (not symmetric, G wanted: the float32 kernel here, transposed planes on the square tiles)
                np.testing.assert_allclose(G1.cpu().numpy(), G2.cpu().numpy(), rtol=0, atol=2e-5)
            np.testing.assert_allclose(e1.cpu().numpy(), e2.cpu().numpy(), rtol=2e-6)
        D = (W - Q).double().cpu().numpy()
        want = ((D @ H.astype(np.float64)) * D).sum(axis=1)
        slkopt.setenv("SLK_TALL_ERROR", "2")
        np.testing.assert_allclose(amd.engine.row_errors(W, Q, Hs).cpu().numpy(), want, rtol=1e-5)
        slkopt.delenv("SLK_TALL_ERROR")
        if R % 768 == 0:
            H2 = torch.from_numpy(((H * np.float32(1.25)) + np.float32(0.01) * np.eye(n, dtype=np.float32))).cuda()
            Wb, Qb = W.view(3, 768, n).contiguous(), Q.view(3, 768, n).contiguous()
            big, square = both(lambda: amd.engine.row_errors_batch(Wb, Qb, [Hs, H2, Has]))
            np.testing.assert_allclose(big.cpu().numpy(), square.cpu().numpy(), rtol=2e-6)


def test_layer_error_bf16_path(amd, slkopt):
    """The layer error of a symmetric Hessian runs on the bfloat16 MFMA with three pieces per operand
    (six products): every row within 1e-5 of the float64 value, like the float32 kernel it replaces --
    ragged row count, float32-MFMA path forced for comparison, and an asymmetric H (float32 kernel, all of H)."""
    rng = np.random.default_rng(11)
    R, n = 200, 1024
    W = rng.standard_normal((R, n)).astype(np.float32)
    Q = (W + 0.2 * rng.standard_normal((R, n))).astype(np.float32)
    X = rng.standard_normal((2 * n, n)).astype(np.float32)
    H = (X.T @ X / (2 * n)).astype(np.float32)
    H = ((H + H.T) * np.float32(0.5)).astype(np.float32)
    D = (W - Q).astype(np.float64)
    want = ((D @ H.astype(np.float64)) * D).sum(axis=1)
    got = amd.obq.channelwise_error(W, Q, H)
    np.testing.assert_allclose(got, want, rtol=1e-5)
    slkopt.setenv("SLK_NO_BF16_DMA", "1")  # operands staged through registers instead of global_load_lds
    np.testing.assert_allclose(amd.obq.channelwise_error(W, Q, H), got, rtol=2e-6)
    slkopt.delenv("SLK_NO_BF16_DMA")
    slkopt.setenv("SLK_NO_BF16_ERROR", "1")
    got32 = amd.obq.channelwise_error(W, Q, H)
    slkopt.delenv("SLK_NO_BF16_ERROR")
    np.testing.assert_allclose(got32, want, rtol=1e-5)
    np.testing.assert_allclose(got, got32, rtol=2e-6)
    # few rows: K is cut into chunks so that the tiles fill the chip (a row shard of a multi-GPU run); every chunk
    # size, and the uncut kernel, within rounding of each other
    for chunk in ("2", "4", "8", "16"):
        slkopt.setenv("SLK_ERROR_CB", chunk)
        np.testing.assert_allclose(amd.obq.channelwise_error(W, Q, H), want, rtol=1e-5)
    slkopt.delenv("SLK_ERROR_CB")
    slkopt.setenv("SLK_NO_ERROR_SPLITK", "1")
    whole = amd.obq.channelwise_error(W, Q, H)
    slkopt.delenv("SLK_NO_ERROR_SPLITK")
    np.testing.assert_allclose(whole, want, rtol=1e-5)
    np.testing.assert_allclose(got, whole, rtol=2e-6)
    # not symmetric: the error d H d^T only sees (H + H^T) / 2, which the split forms on the way into the planes, so the
    # half-product route serves it too (default); or every k is multiplied -- on the bfloat16 MFMA from planes of H^T
    # (first switch), on the float32 MFMA (both switches; also what the K-chunked few-row layout did before the average)
    Ha = (H + np.float32(0.05) * np.triu(rng.standard_normal((n, n)).astype(np.float32), 1)).astype(np.float32)
    Ha[3, 7] += np.float32(0.25)
    want_a = ((D @ Ha.astype(np.float64)) * D).sum(axis=1)
    got_a = amd.obq.channelwise_error(W, Q, Ha)
    np.testing.assert_allclose(got_a, want_a, rtol=1e-5)
    slkopt.setenv("SLK_NO_SYM_AVERAGE", "1")
    planes_t = amd.obq.channelwise_error(W, Q, Ha)
    np.testing.assert_allclose(planes_t, want_a, rtol=1e-5)
    np.testing.assert_allclose(planes_t, got_a, rtol=4e-6)
    slkopt.setenv("SLK_NO_BF16_ASYM", "1")
    np.testing.assert_allclose(amd.obq.channelwise_error(W, Q, Ha), got_a, rtol=4e-6)
    slkopt.delenv("SLK_NO_BF16_ASYM")
    slkopt.delenv("SLK_NO_SYM_AVERAGE")
    for chunk in ("2", "8"):  # the K-chunked layout of few-row calls takes the averaged planes like any symmetric H
        slkopt.setenv("SLK_ERROR_CB", chunk)
        np.testing.assert_allclose(amd.obq.channelwise_error(W, Q, Ha), want_a, rtol=1e-5)
    slkopt.delenv("SLK_ERROR_CB")
    # the product itself (what the local search starts from): G = (W - Q) @ Ha, both routes
    Wd, Qd, Hd = (torch.from_numpy(x).cuda() for x in (W, Q, Ha))
    _, G1 = amd.engine.row_errors(Wd, Qd, Hd, want_G=True)
    slkopt.setenv("SLK_NO_BF16_ASYM", "1")
    _, G2 = amd.engine.row_errors(Wd, Qd, Hd, want_G=True)
    slkopt.delenv("SLK_NO_BF16_ASYM")
    Gw = D @ Ha.astype(np.float64)
    unit = 2.0 ** -24 * (np.abs(D) @ np.abs(Ha.astype(np.float64)))  # one rounding of the magnitude of an entry's terms
    e1, e2 = np.abs(G1.cpu().numpy() - Gw) / unit, np.abs(G2.cpu().numpy() - Gw) / unit
    assert e1.max() <= 32 and e2.max() <= 32, (e1.max(), e2.max())  # (K = 1024 terms: the worst case is ~K / 2 roundings)
    print("G error in roundings of its terms: bfloat16 x 3", e1.max(), "float32 MFMA", e2.max())


def test_table_codebook(amd, pieces):
    """codebook.py:98-190: the general codebook on the GPU -- the four maps bit for bit, then whole layers
    (orders that call the quantizer, local search with its up / down candidates) against the reference's outputs."""
    x = pieces["cb/x"]
    for tag in ("nf4", "odd"):
        cb = amd.codebook.Codebook(pieces[f"cbt/{tag}/values"], pieces[f"cbt/{tag}/limits"])
        for k in ("value", "index", "up", "down"):
            got = getattr(cb, "quantize_" + k)(x)
            assert got.dtype == pieces[f"cbt/{tag}/{k}"].dtype and np.array_equal(got, pieces[f"cbt/{tag}/{k}"]), (tag, k)
    nf4 = amd.codebook.Codebook.nf4()
    assert len(nf4) == 16 and nf4.min() == -1.0 and nf4.max() == 1.0
    assert np.array_equal(nf4.values, pieces["cbt/nf4/values"]) and np.array_equal(nf4.thresholds, pieces["cbt/nf4/limits"])
    for R, n, seed in ((64, 96, 2001), (96, 172, 2003)):
        L = layer(R, n, seed)
        for order, moves in (("diag", 0), ("sqerr", 0), ("err", 10), ("diag", 10)):
            want = pieces[f"cbt/nf4/layer_r{R}_n{n}_s{seed}_{order}_ls{moves}"]
            got = amd.scaling.quantize_with_scaling(L["W"], L["scale"], nf4, L["H"], act_order=order, damp=0.01, nb_ls_moves=moves)
            assert np.array_equal(got, want), (n, order, moves)
    # a wide layer: the standard-schedule window kernel with a table in LDS, against the oracle
    L = layer(40, 1024, 2060)
    want = scaling_ref.quantize_scaled(L["W"], L["scale"], grid.TableGrid.nf4(), L["H"], "diag", 0.01, 0)
    got = amd.scaling.quantize_with_scaling(L["W"], L["scale"], nf4, L["H"], act_order="diag", damp=0.01)
    assert np.array_equal(got, want)


def test_pivot_order(amd, pieces):
    """obq.py:140-166: greedy pivoted Cholesky.  The order against the reference's own (fixture), then a
    layer quantized in that order against the oracle."""
    cb = amd.codebook.UniformCodebook(8, -1, 1)
    for tag, (R, n, seed) in (("96", (64, 96, 2001)), ("256", (32, 256, 2050))):
        L = layer(R, n, seed)
        H32 = L["H"].astype(np.float32)
        Hd = H32 + np.float32(0.01 * H32.diagonal().mean()) * np.eye(n)
        got = amd.obq.compute_hessian_order(L["W"], Hd.astype(np.float32), cb, "pivot")
        # the damped float64 matrix rounded to float32 is not the reference's input: the device API damps itself
        import torch
        keys = amd.engine.pivot_keys(torch.as_tensor(H32, device="cuda"), n, 0.01)
        order = torch.argsort(keys).cpu().numpy()
        assert np.array_equal(order, pieces[f"pivot/{tag}/order"]), tag
        assert sorted(got.tolist()) == list(range(n))
        g = grid.UniformGrid(8, -1, 1)
        want = scaling_ref.quantize_scaled(L["W"], L["scale"], g, L["H"], "pivot", 0.01, 0)
        out = amd.scaling.quantize_with_scaling(L["W"], L["scale"], cb, L["H"], act_order="pivot", damp=0.01)
        assert np.array_equal(out, want), tag


def test_local_search_standalone(amd):
    L = layer(64, 96, 2001)
    g = grid.UniformGrid(8, -1, 1)
    cb = amd.codebook.UniformCodebook(8, -1, 1)
    Ws = scaling_ref.divide_rows(L["W"], L["scale"], 0)
    Q0 = g(Ws)
    for moves in (1, 3, 25):
        want = obq_ref.local_search(Ws, Q0, L["H"], g, moves)
        got = amd.obq.quantize_local_search(Ws, Q0, L["H"], cb, moves)
        assert np.array_equal(got, want), moves
    assert amd.obq.quantize_local_search(Ws, Q0, L["H"], cb, 0) is Q0
    # the stateful class (obq.py:234-346): k calls of do_move() == quantize_local_search(k), bit for bit, and the gains it
    # holds after them are the oracle's incrementally updated ones wherever the moves agree (they do on this layer)
    ls = amd.obq.LocalSearchQuantizer(Ws, Q0, L["H"], cb)
    state = obq_ref._SearchState(Ws, Q0, L["H"], g)
    for k in range(1, 8):
        ls.do_move()
        state.move()
        assert np.array_equal(ls.Q, amd.obq.quantize_local_search(Ws, Q0, L["H"], cb, k)), k
        assert np.array_equal(ls.Q, state.Q), k
    for got, want in ((ls.gain_up, state.gain[+1]), (ls.gain_down, state.gain[-1])):
        unit = 2.0 ** -24 * 2.0 * (2.0 / 7.0) * obq_ref.gain_noise_scale(Ws, Q0, L["H"])
        assert (np.abs(got.astype(np.float64) - want) <= 8.0 * unit + 1e-12).all()
    # gains (obq.py:220-231): -D^2 H_jj - 2 (delta @ H)_j D.  The GEMM's summation order is the BLAS's own, so the
    # comparison is per entry against the rounding of ITS terms: 2 |D| (|delta| @ |H|)_j + D^2 H_jj, a few roundings
    for cand in (g.up(Q0), g.down(Q0)):
        got, want = amd.obq.compute_gain(Ws, Q0, L["H"], cand), obq_ref.flip_gains(Ws, Q0, L["H"], cand)
        D = np.abs(cand - Q0).astype(np.float64)
        terms = 2.0 * D * obq_ref.gain_noise_scale(Ws, Q0, L["H"]) + D * D * L["H"].diagonal()
        assert (np.abs(got.astype(np.float64) - want) <= 4.0 * 2.0 ** -24 * terms + 1e-30).all()


@pytest.mark.parametrize("n", [768, 1024, 1536, 2048, 3072, 4096, 6144, 8192])
def test_local_search_wave_kernel_is_the_workgroup_kernel(amd, n):
    """Row lengths whose NumPy summation tree is regular take the chain-per-lane search kernels -- 8 or 16 leaves: a wave
    per row, no LDS, no barrier; 32 or 64 leaves: a workgroup per row, two barriers per move -- with the same moves, the
    same final values and the same carried gains as the general workgroup-per-row kernel, bit for bit; and as the
    oracle on the narrowest one."""
    R = 70  # (not a multiple of the 4 rows per workgroup)
    L = synth.make_layer_device(R, n, 4200 + n, torch.device("cuda"))
    cb = amd.codebook.UniformCodebook(8, -1, 1)
    abi = cb._abi()
    Ws = amd.engine.rows_divide(L["W"], L["scale"])
    Q0 = cb.quantize_value(Ws)
    out = []
    for general in (-1, 1):  # -1: the wave kernel whatever the row count, 1: never
        with amd.lib.option("no_wave_search", general):
            Q = Q0.clone()
            idx = torch.empty((R, n), dtype=torch.uint8, device="cuda")
            gains = torch.empty((R, 2, n), dtype=torch.float32, device="cuda")
            trace = amd.engine.local_search(Ws, Q, L["H"], abi, 12, idx, want_trace=True, gains=gains, gains_mode=1)
            # and carried on: three more single moves from the stored gains
            for _ in range(3):
                amd.engine.local_search(Ws, Q, L["H"], abi, 1, idx, gains=gains, gains_mode=2)
            out.append((Q, idx, trace, gains))
    for a, b in zip(out[0], out[1]):
        assert torch.equal(a, b)
    assert int((out[0][2] >= 0).sum()) > R  # moves were made
    if n == 768:
        g = grid.UniformGrid(8, -1, 1)
        want = obq_ref.local_search(Ws.cpu().numpy(), Q0.cpu().numpy(), L["H"].cpu().numpy(), g, 15)
        bad = np.flatnonzero((out[0][0].cpu().numpy() != want).any(axis=1))
        assert len(bad) <= 1  # (a near-tie of the initial product may fall the other way: proven elsewhere, ls_evidence.py)


@pytest.mark.parametrize("kind", ["nf4", "uniform300"])
def test_local_search_kernels_on_other_codebooks(amd, kind):
    """The search kernels' other instantiations: a general codebook (binary searches; candidates kept as floats or
    recomputed) and a uniform grid of more than 256 levels (no packed levels: the general kernel, recomputing) -- the
    chain-per-lane kernels against the general one bit for bit on regular rows, and both against the oracle."""
    cb = amd.codebook.Codebook.nf4() if kind == "nf4" else amd.codebook.UniformCodebook(300, -1, 1)
    g = grid.TableGrid.nf4() if kind == "nf4" else grid.UniformGrid(300, -1, 1)
    abi = cb._abi()
    for n, R in ((1024, 22), (4096, 6), (1100, 10)):
        L = synth.make_layer_device(R, n, 4400 + n, torch.device("cuda"))
        Ws = amd.engine.rows_divide(L["W"], L["scale"])
        Q0 = torch.from_numpy(g(Ws.cpu().numpy())).cuda()
        out = []
        for general in (-1, 1):
            with amd.lib.option("no_wave_search", general):
                Q = Q0.clone()
                gains = torch.empty((R, 2, n), dtype=torch.float32, device="cuda")
                trace = amd.engine.local_search(Ws, Q, L["H"], abi, 9, None, want_trace=True, gains=gains, gains_mode=1)
                out.append((Q, trace, gains))
        for a, b in zip(out[0], out[1]):
            assert torch.equal(a, b), (kind, n)
        assert int((out[0][1] >= 0).sum()) > R
        want = obq_ref.local_search(Ws.cpu().numpy(), Q0.cpu().numpy(), L["H"].cpu().numpy(), g, 9)
        bad = np.flatnonzero((out[0][0].cpu().numpy() != want).any(axis=1))
        assert len(bad) <= 1, (kind, n, bad)  # (a near-tie of the initial product may fall the other way: ls_evidence.py)


@pytest.mark.parametrize("n", [172, 768, 1024, 4096])
def test_local_search_carries_the_row_errors(amd, n):
    """slk_local_search's `row_err`: every row's error (W - Q) H (W - Q)^T after the moves, carried through the search the way
    the reference's LocalSearchQuantizer carries `err` (obq.py:254, 290: the error before, minus the gain of every move) --
    against the product recomputed from the final Q (what quantization_error does, obq.py:89-103), both kernel families."""
    R = 70
    L = synth.make_layer_device(R, n, 4300 + n, torch.device("cuda"))
    cb = amd.codebook.UniformCodebook(8, -1, 1)
    abi = cb._abi()
    Ws = amd.engine.rows_divide(L["W"], L["scale"])
    for general in (-1, 1):
        with amd.lib.option("no_wave_search", general):
            Q = cb.quantize_value(Ws)
            carried = torch.empty(R, dtype=torch.float32, device="cuda")
            trace = amd.engine.local_search(Ws, Q, L["H"], abi, 15, want_trace=True, row_err=carried)
            assert int((trace >= 0).sum()) > R
            again = amd.engine.row_errors(Ws, Q, L["H"])
            # per row: float32 gains, as exact as the reference's own `ls.err`; the layer's mean: far inside 1e-5
            np.testing.assert_allclose(carried.cpu().numpy(), again.cpu().numpy(), rtol=2e-4)
            assert abs(float(carried.double().mean()) - float(again.double().mean())) <= 1e-6 * float(again.double().mean())
    # end to end through the row-shard backend: a layer that VOUCHES for its Hessian's symmetry gets the carried error,
    # de-scaled; one that does not gets the recomputed product (an asymmetric H would make the carried one wrong)
    from sleekit_amd import dist as sdist

    assert torch.equal(L["H"], L["H"].T)
    be = sdist.HipBackend(cb, "diag", 0.01, 10, with_error=True, overlap=False)
    calls = []
    row_errors_batch = amd.engine.row_errors_batch
    try:
        amd.engine.row_errors_batch = lambda *a, **k: (calls.append(1), row_errors_batch(*a, **k))[1]
        for vouched in (True, False):
            lay = {k: L[k] for k in ("W", "H", "scale")}
            if vouched:
                lay["symmetric"] = True
            del calls[:]
            shard = be.run_rows(lay, 0, R, be.factorize(lay))
            assert len(calls) == (0 if vouched else 1)
            want = amd.engine.row_errors(L["W"], shard["Q"], L["H"])
            np.testing.assert_allclose(shard["row_err"].cpu().numpy(), want.cpu().numpy(), rtol=2e-4 if vouched else 1e-5)
            assert abs(float(shard["row_err"].double().mean()) - float(want.double().mean())) <= 1e-6 * float(want.double().mean())
    finally:
        amd.engine.row_errors_batch = row_errors_batch


def test_hessian_accumulate(amd, pieces):
    X = pieces["stats/X"]
    lin = torch.nn.Linear(48, 10).cuda()
    st = amd.statistics.Sleekit(lin)
    for a, b in ((0, 64), (64, 72), (72, 200)):
        st.add_batch(torch.from_numpy(X[a:b]).reshape(1, b - a, 48).cuda())
    assert st.count == int(pieces["stats/count"])
    np.testing.assert_allclose(st.hessian.cpu().numpy(), pieces["stats/H"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(st.mean.cpu().numpy(), pieces["stats/mean"], rtol=1e-5, atol=1e-7)
    H = st.hessian.cpu().numpy()
    assert np.array_equal(H, H.T)
    # a larger, ragged shape against the float64 truth
    n, T = 300, 1000
    Xb = synth.make_activations(T, n, 77).astype(np.float32)
    st = amd.statistics.Sleekit(torch.nn.Linear(n, 4).cuda())
    st.add_batch(torch.from_numpy(Xb[:333]).cuda())
    st.add_batch(torch.from_numpy(Xb[333:]).cuda())
    truth = Xb.astype(np.float64).T @ Xb.astype(np.float64) / T
    np.testing.assert_allclose(st.hessian.cpu().numpy(), truth, rtol=1e-5, atol=1e-5 * np.abs(truth).max())


def test_adapter_counts_like_reference_tests(amd):
    """Sample counting for Linear / Conv2d / Conv1d: the numbers of the reference's tests/test_statistics.py."""
    nn_ = torch.nn
    st = amd.statistics.Sleekit(nn_.Linear(10, 5).cuda())
    for shape, want in (((10,), 1), ((3, 10), 4), ((3, 3, 10), 13)):
        st.add_batch(torch.randn(*shape))
        assert st.count == want
    st = amd.statistics.Sleekit(nn_.Conv2d(10, 5, 3).cuda())
    for shape, want in (((10, 3, 3), 1), ((5, 10, 3, 3), 6), ((10, 7, 7), 31)):
        st.add_batch(torch.randn(*shape))
        assert st.count == want
    st = amd.statistics.Sleekit(nn_.Conv2d(10, 5, 3, padding=1).cuda())
    for shape, want in (((10, 3, 3), 9), ((5, 10, 3, 3), 54), ((10, 5, 5), 79)):
        st.add_batch(torch.randn(*shape))
        assert st.count == want
    st = amd.statistics.Sleekit(nn_.Conv1d(10, 5, 3).cuda())
    for shape, want in (((10, 3), 1), ((5, 10, 3), 6), ((10, 7), 11)):
        st.add_batch(torch.randn(*shape))
        assert st.count == want
    with pytest.raises(ValueError):
        amd.statistics.Sleekit(nn_.ReLU())


def test_adapter_conv_statistics(amd, pieces):
    nn_ = torch.nn
    for kind, layer in (("conv2d", nn_.Conv2d(6, 5, 3, padding=1, stride=2)), ("conv1d", nn_.Conv1d(6, 5, 3, dilation=2))):
        x = torch.from_numpy(pieces[f"adapter/{kind}/x"])
        st = amd.statistics.Sleekit(layer.cuda())
        st.add_batch(x)
        st.add_batch(x[0] if kind == "conv2d" else x[1])
        assert st.count == int(pieces[f"adapter/{kind}/count"])
        np.testing.assert_allclose(st.hessian.cpu().numpy(), pieces[f"adapter/{kind}/H"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(st.mean.cpu().numpy(), pieces[f"adapter/{kind}/mean"], rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("preset,bits", [("basic", 4), ("sleekit_light", 3), ("sleekit_heavy", 3)])
def test_adapter_presets_end_to_end(amd, pieces, preset, bits):
    """Sleekit(layer).add_batch(...) x2, quantize_<preset>(bits): weights and corrected bias vs the reference.

    The Hessian comes out of a float32 MFMA product here and out of torch's CPU GEMM there, so the
    statistics agree to ~1e-6 and a few weights may sit on the other side of a rounding tie.
    """
    lin = torch.nn.Linear(40, 24)
    with torch.no_grad():
        lin.weight.copy_(torch.from_numpy(synth.make_weights(24, 40, 2032)))
        lin.bias.copy_(torch.from_numpy((0.1 * synth.normal_grid(2032, 11, 1, 24)[0]).astype(np.float32)))
    lin = lin.cuda()
    X = torch.from_numpy(pieces["adapter/X"])
    st = amd.statistics.Sleekit(lin)
    st.add_batch(X[:128])
    st.add_batch(X[128:].reshape(2, 86, 40))
    getattr(st, "quantize_" + preset)(bits)
    got_w, want_w = lin.weight.detach().cpu().numpy(), pieces[f"adapter/{preset}/weight"]
    same_rows = np.isclose(got_w, want_w, rtol=1e-5, atol=1e-7).all(axis=1)
    assert same_rows.mean() >= 0.9, (preset, float(same_rows.mean()))
    np.testing.assert_allclose(lin.bias.detach().cpu().numpy()[same_rows], pieces[f"adapter/{preset}/bias"][same_rows], rtol=1e-3, atol=1e-5)


@pytest.mark.parametrize("preset,bits", [("basic", 4), ("sleekit_light", 3)])
def test_adapter_presets_from_the_reference_statistics(amd, pieces, preset, bits):
    """The same presets with the REFERENCE's accumulated Hessian and mean placed on the device (so that the MFMA
    accumulation's own rounding is out of the picture): `basic` (mse scale search, diag order) and `sleekit_light`
    (diagonal-Hessian scale search, sqerr order, H - m m^T, damp 0.03) involve no choice ranked by a GEMM, so the
    quantized weights must be the reference's bit for bit; the corrected bias is a float32 row sum (torch's order there)."""
    lin = torch.nn.Linear(40, 24)
    with torch.no_grad():
        lin.weight.copy_(torch.from_numpy(synth.make_weights(24, 40, 2032)))
        lin.bias.copy_(torch.from_numpy((0.1 * synth.normal_grid(2032, 11, 1, 24)[0]).astype(np.float32)))
    lin = lin.cuda()
    st = amd.statistics.Sleekit(lin)
    st.hessian.copy_(torch.from_numpy(pieces["adapter/H_ref"]))
    st.mean.copy_(torch.from_numpy(pieces["adapter/mean_ref"]))
    st.count = int(pieces["adapter/count_ref"])
    getattr(st, "quantize_" + preset)(bits)
    assert np.array_equal(lin.weight.detach().cpu().numpy(), pieces[f"adapter/{preset}/weight"]), preset
    np.testing.assert_allclose(lin.bias.detach().cpu().numpy(), pieces[f"adapter/{preset}/bias"], rtol=1e-5, atol=1e-7)


def test_compare_experiment_sequence(amd):
    """The call sequence of the reference's experiments/compare.py:54-131 (its five quantization recipes),
    run through the drop-in modules and through the oracle; the five reported errors must agree."""
    L = layer(96, 172, 2003, )
    Ld = synth.make_layer(96, 172, 2050, dead=(5, 99))

    def recipes(m_obq, m_sc, cb, W, H, mean):
        W, H = W.copy(), H.copy()
        m_obq["dead"](H, W)
        Hc = m_obq["strip"](H, mean)
        out = []
        sc = m_sc["minmse"](W, cb, grid_size=25)
        out.append(m_obq["err"](W, m_sc["qws"](W, sc, cb, H=H, act_order="diag", damp=0.01), H))
        out.append(m_obq["err"](W, m_sc["qws"](W, sc, cb, H=Hc, act_order="diag", damp=0.01), Hc))
        sc = m_sc["minmse"](W, cb, H=H.diagonal().copy(), grid_size=25)
        out.append(m_obq["err"](W, m_sc["qws"](W, sc, cb, H=H, damp=0.01), H))
        sc = m_sc["minmse"](W, cb, H=Hc.diagonal().copy(), grid_size=25)
        out.append(m_obq["err"](W, m_sc["qws"](W, sc, cb, H=Hc, act_order="sqerr", damp=0.03), Hc))
        sc = m_sc["obq"](W, cb, 0, H=Hc, act_order="sqerr", damp=0.03, grid_size=12)
        out.append(m_obq["err"](W, m_sc["qws"](W, sc, cb, H=Hc, act_order="sqerr", damp=0.03, nb_ls_moves=20), Hc))
        return [float(x) for x in out]

    ref_obq = dict(dead=obq_ref.patch_dead_columns, strip=obq_ref.strip_input_mean, err=obq_ref.mean_error)
    ref_sc = dict(minmse=scaling_ref.best_grid_scale,
                  obq=lambda W, cb, axis, H, act_order, damp, grid_size:
                  scaling_ref.best_obq_scale(W, cb, axis, H, damp=damp, order_mode=act_order, grid_size=grid_size),
                  qws=lambda W, sc, cb, H=None, act_order="diag", damp=0.01, nb_ls_moves=0:
                  scaling_ref.quantize_scaled(W, sc, cb, H, act_order, damp, nb_ls_moves))
    amd_obq = dict(dead=amd.obq.remove_dead_values, strip=amd.obq.remove_input_bias, err=amd.obq.quantization_error)
    amd_sc = dict(minmse=amd.scaling.compute_min_mse_scaling, obq=amd.scaling.compute_obq_scaling,
                  qws=amd.scaling.quantize_with_scaling)
    for lay in (L, Ld):
        for levels in (8, 3):
            want = recipes(ref_obq, ref_sc, grid.UniformGrid(levels, -1, 1), lay["W"], lay["H"], lay["mean"])
            got = recipes(amd_obq, amd_sc, amd.codebook.UniformCodebook(levels, -1, 1), lay["W"], lay["H"], lay["mean"])
            # the first four recipes are deterministic end to end; the last one (OBQ-aware grid search +
            # local search) may resolve a near-tie differently, which moves the error by a hair
            np.testing.assert_allclose(got[:4], want[:4], rtol=1e-5)
            np.testing.assert_allclose(got[4], want[4], rtol=2e-3)


# --------------------------------------------------------------------------- BASELINE-sized layers
def row_hashes(idx):
    return np.array([int.from_bytes(hashlib.sha256(np.ascontiguousarray(r).tobytes()).digest()[:8], "little") for r in idx],
                    dtype=np.uint64)


def _large_with_moves(amd, L, c, T):
    """Local search on a BASELINE-sized layer: every row must carry the reference's indices (per-row hashes in
    ls_traces.npz) unless it is a PROVEN near-tie (ls_evidence.explain_rows) -- no allowance by count."""
    name = f"r{c['R']}_n{c['n']}_s{c['seed']}_N{c['levels']}_{c['order']}_ls{c['moves']}"
    assert str(T[name + "/sha_idx"]) == c["sha_idx"], "ls_traces.npz and large_cases.json were made from different runs"
    cb = amd.codebook.UniformCodebook(c["levels"], -1, 1)
    W, H, sc = (torch.from_numpy(L[k]).cuda() for k in ("W", "H", "scale"))
    assert not c["strip_mean"]
    res = amd.engine.quantize_layer(W, H, cb, sc, c["order"], c["damp"], c["moves"], want_ls_trace=True)
    err = float(amd.obq.quantization_error(W, res.Q, H))
    assert abs(err - c["err"]) <= 1e-5 * abs(c["err"]), (err, c["err"])
    idx = res.idx.cpu().numpy()
    if sha(idx) == c["sha_idx"]:
        return "bit-exact"
    bad = np.flatnonzero(row_hashes(idx) != T[name + "/row_hash"])
    near = {k: T[f"{name}/{k}"] for k in ("rows", "choice", "runner", "ratio")}
    worst = explain_rows(bad, res.ls_trace.cpu().numpy(), near)
    return f"{len(bad)} proven near-tie rows of {c['R']} (closest margins: <= {worst:.3g} roundings)"


def _large(amd, c, T=None):
    """A BASELINE-sized layer against the hash the REAL reference produced (tests/golden/large_cases.json).

    Bit-exact is the rule.  Two documented exceptions, both properties of the reference itself
    (DESIGN.md "Parity"):
      * exact ties in the ordering key: NumPy's default argsort is unstable, the device sort is
        stable.  Then the product must equal the oracle run with stable ties, bit for bit,
        and may differ from the golden hash in a handful of rows;
      * local search: a move at which the reference's best candidate leads the runner-up by less than the
        rounding of its initial BLAS product can fall the other way.  Every differing row must be PROVEN to
        be such a case from the reference's recorded moves (_large_with_moves); there is no allowance by count.
    The layer error must match to 1e-5 relative in every case.
    """
    L = layer(c["R"], c["n"], c["seed"])
    assert sha(L["W"]) == c["sha_W"] and sha(L["H"]) == c["sha_H"] and sha(L["scale"]) == c["sha_scale"]
    if c["moves"] > 0:
        return _large_with_moves(amd, L, c, T)
    spec = dict(levels=c["levels"], order=c["order"], damp=c["damp"], moves=c["moves"], strip=c["strip_mean"])
    out, idx, rows, err = run_product(amd, L, spec)
    assert abs(float(err) - c["err"]) <= 1e-5 * abs(c["err"]), (float(err), c["err"])
    if sha(idx) == c["sha_idx"]:
        return "bit-exact"
    g = grid.UniformGrid(c["levels"], -1, 1)
    H = obq_ref.strip_input_mean(L["H"], L["mean"]) if c["strip_mean"] else L["H"]
    Hd_diag = H.diagonal().astype(np.float64) + np.float64(np.float32(c["damp"]) * H.diagonal().mean())
    has_ties = len(np.unique(Hd_diag)) < len(Hd_diag)
    want = scaling_ref.quantize_scaled(L["W"], L["scale"], g, H, c["order"], c["damp"], c["moves"],
                                       ties="stable" if has_ties else "numpy")
    want_idx = g.index(scaling_ref.divide_rows(want, L["scale"], 0))
    bad_rows = int((idx != want_idx).any(axis=1).sum())
    assert has_ties, "no ties and no local search: indices must match the reference bit for bit"
    assert bad_rows == 0, f"{bad_rows} rows differ from the oracle with stable tie-breaking"
    return "bit-exact up to the order of tied keys"


@pytest.mark.parametrize("shape", ["768x768", "3072x768", "768x3072", "1024x1024", "1024x4096", "3072x1024", "4096x1024", "4096x4096"])
def test_large_cases_against_reference_hashes(amd, large_cases, ls_traces, shape):
    todo = [c for c in large_cases if f"{c['R']}x{c['n']}" == shape]
    assert todo
    for c in todo:
        print(shape, c["seed"], _large(amd, c, ls_traces))


def test_small_local_search_cases_follow_the_reference_moves(amd, ls_traces, small_cases):
    """Every small fixture case with moves: final indices bit-exact (test_small_cases_bit_exact) AND the rows that
    came within 64 roundings of a tie take the reference's recorded moves one by one."""
    for name in (str(x) for x in ls_traces["names"]):
        c = parse_case(name)
        if c["R"] * c["n"] > 256 * 768:
            continue
        L = layer(c["R"], c["n"], c["seed"])
        cb = amd.codebook.UniformCodebook(c["levels"], -1, 1)
        W, H, sc = (torch.from_numpy(L[k]).cuda() for k in ("W", "H", "scale"))
        res = amd.engine.quantize_layer(W, H, cb, sc, c["order"], c["damp"], c["moves"], want_ls_trace=True)
        idx, trace = res.idx.cpu().numpy(), res.ls_trace.cpu().numpy()
        bad = np.flatnonzero((idx != small_cases[name + "/idx"]).any(axis=1))
        near = {k: ls_traces[f"{name}/{k}"] for k in ("rows", "choice", "runner", "ratio")}
        explain_rows(bad, trace, near)
        same = [np.array_equal(trace[r], near["choice"][i]) for i, r in enumerate(near["rows"]) if r not in bad]
        assert all(same), name


def test_cfg5_row_shard_11008(amd, large_cases):
    """BASELINE cfg5: one 512-row shard of a 4096 x 11008 layer, 2-bit (leaves 5 x 32 + 12 per 172, K up to 1376)."""
    c = next(c for c in large_cases if (c["R"], c["n"]) == (512, 11008))
    L = synth.make_layer(c["R"], c["n"], c["seed"], device=torch.device("cuda"))
    _layers[(c["R"], c["n"], c["seed"], ())] = L
    print("512x11008", _large(amd, c))


def test_cfg5_full_layer_11008(amd, large_cases):
    """BASELINE cfg5: a WHOLE 4096 x 11008 layer at 2 bit against the hash the real reference produced (its inputs made
    on the device: same bytes as the host generator, test_device_generator_makes_the_same_bytes)."""
    c = next(c for c in large_cases if (c["R"], c["n"]) == (4096, 11008))
    L = synth.make_layer_device(c["R"], c["n"], c["seed"], torch.device("cuda"))
    _layers[(c["R"], c["n"], c["seed"], ())] = {k: L[k].cpu().numpy() for k in ("W", "H", "mean", "scale")}
    del L
    torch.cuda.empty_cache()
    print("4096x11008", _large(amd, c))
    del _layers[(c["R"], c["n"], c["seed"], ())]


def test_headline_properties_4096(amd):
    """Size-independent properties at the headline size: row shards are independent, indices
    decode to the returned values, re-blocking is (almost) a pure re-association."""
    L = layer(4096, 4096, 1007)
    cb = amd.codebook.UniformCodebook(8, -1, 1)
    W, H, sc = (torch.from_numpy(L[k]).cuda() for k in ("W", "H", "scale"))
    full = amd.engine.quantize_layer(W, H, cb, sc)
    # (1) a row shard with the same factor gives the same rows
    part = amd.engine.quantize_layer(W[1024:1536].contiguous(), H, cb, sc[1024:1536].contiguous(),
                                     factor=(full.order, full.U, full.info))
    assert torch.equal(part.idx, full.idx[1024:1536]) and torch.equal(part.Q, full.Q[1024:1536])
    # (2) indices decode to the scaled-domain values
    scaled = amd.engine.quantize_layer(W, H, cb, sc, unscale=False, factor=(full.order, full.U, full.info))
    vals = torch.linspace(-1, 1, 8, device="cuda", dtype=torch.float64)[full.idx.long()].float()
    assert torch.allclose(scaled.Q, vals, rtol=0, atol=1e-6)
    # (3) another blocking moves the float32 rounding points: all but a few near-tie decisions
    #     survive (the reference's own test allows the same: tests/test_obq.py:68-70)
    other = amd.engine.quantize_layer(W, H, cb, sc, min_block_size=32, num_blocks=4, factor=(full.order, full.U, full.info))
    assert (other.idx != full.idx).sum().item() <= 1e-5 * full.idx.numel()
    e_full = float(amd.obq.quantization_error(W, full.Q, H))
    assert abs(float(amd.obq.quantization_error(W, other.Q, H)) - e_full) <= 1e-5 * e_full
    # (4) GPTQ beats round-to-nearest
    rtn = amd.scaling.quantize_with_scaling(W, sc, cb)
    assert e_full < float(amd.obq.quantization_error(W, rtn, H))


def test_exchange_over_rccl_single_rank():
    """The N > 1 exchange (pack, RCCL all-gather on the comm stream, unpack) driven on this one GPU."""
    import subprocess
    import sys

    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rccl_single_rank.py")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, script, "29643"], capture_output=True, text=True, timeout=280, env=env)
    assert out.returncode == 0 and "RCCL_SINGLE_RANK_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]


def test_batched_rounds_two_ranks_one_gpu():
    """Two ranks on this one GPU (gloo): the batched rounds of sleekit_amd.dist against the unsharded result."""
    import subprocess
    import sys

    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dist_round_gpu.py")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29657", script]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=280, env=env)
    assert out.returncode == 0 and out.stdout.count("DIST_ROUND_OK") == 2, out.stdout[-2000:] + out.stderr[-4000:]


@pytest.mark.parametrize("B,R,n,levels", [(3, 128, 320, 8), (2, 256, 1024, 8), (4, 128, 1100, 4), (8, 128, 512, "nf4"), (5, 64, 192, 3)])
def test_batch_entry_points(amd, B, R, n, levels):
    """slk_gptq_quantize_batch / slk_row_errors_batch: B layers stacked by rows == B separate calls."""
    eng = amd.engine
    cb = amd.codebook.Codebook.nf4() if levels == "nf4" else amd.codebook.UniformCodebook(levels, -1, 1)
    abi = eng.require_uniform(cb)
    Ls = [layer(R, n, 5100 + 7 * b + n) for b in range(B)]
    dev_ = torch.device("cuda")
    W = torch.stack([torch.from_numpy(L["W"]) for L in Ls]).to(dev_)
    sc = torch.stack([torch.from_numpy(L["scale"]) for L in Ls]).to(dev_)
    Hs = [torch.from_numpy(L["H"]).to(dev_) for L in Ls]
    Hs[-1] = Hs[-1].clone()
    Hs[-1][1, 3] += 0.25  # the last layer's Hessian is not symmetric: float32 kernel, every k
    facs = [eng.factorize(H, n, 0.01, amd.lib.ORDER_MODES["diag"]) for H in Hs]
    order = torch.stack([f[0] for f in facs])
    U = torch.stack([f[1] for f in facs])
    for scale, unscale in ((sc, True), (sc, False), (None, False)):
        Q, idx = eng.run_loop_batch(W, scale, order, U, abi, 32, 8, unscale=unscale)
        for b in range(B):
            q1, i1, _ = eng.run_loop(W[b], None if scale is None else scale[b], facs[b][0], facs[b][1], abi, 32, 8, unscale=unscale)
            assert torch.equal(Q[b], q1) and torch.equal(idx[b], i1), (b, unscale)
        # the 16-row window workgroups against the default 32-row ones (eight rows per chain wave; a tile of 32 rows never
        # straddles two layers of a stack: rows per layer are a multiple of 64): the same bits, uniform and table codebooks alike
        with amd.lib.option("window_rows", 16):
            Q16, idx16 = eng.run_loop_batch(W, scale, order, U, abi, 32, 8, unscale=unscale)
        assert torch.equal(Q16, Q) and torch.equal(idx16, idx), unscale
    if R % 128:  # 64-row shards: the loop batches (tiles of 64), the error entry needs 128
        with pytest.raises(RuntimeError, match="multiple of 128"):
            eng.row_errors_batch(W, W, Hs)
        return
    # the local search over the stack == a search per layer (the last layer's Hessian is not symmetric): values, indices
    Q0, idx0 = eng.run_loop_batch(W, None, order, U, abi, 32, 8)
    Qb, idxb = Q0.clone(), idx0.clone()
    eng.local_search_batch(W, Qb, Hs, abi, 10, idxb)
    moved = 0
    for b in range(B):
        q1, i1 = Q0[b].clone(), idx0[b].clone()
        eng.local_search(W[b], q1, Hs[b], abi, 10, i1)
        assert torch.equal(Qb[b], q1) and torch.equal(idxb[b], i1), b
        moved += int((q1 != Q0[b]).sum())
    assert moved > 0
    Q, _ = eng.run_loop_batch(W, sc, order, U, abi, 32, 8, unscale=True)
    err = eng.row_errors_batch(W, Q, Hs)
    flags = torch.cat([eng.symmetry_flag(H) for H in Hs])
    assert flags.tolist() == [1] * (B - 1) + [0]
    assert torch.equal(eng.row_errors_batch(W, Q, Hs, flags), err)  # verdicts handed in: same route, same sums
    # the batch goes through the bfloat16 x 3 kernel; the float32 kernel (H as it stands, one launch for the batch), which
    # few-row shards took until the end of round 2 and a switch still selects, must agree
    with amd.lib.option("error_f32_below", 1 << 20):
        np.testing.assert_allclose(eng.row_errors_batch(W, Q, Hs, flags).cpu().numpy(), err.cpu().numpy(), rtol=4e-6)
    for b in range(B):
        want = ((W[b] - Q[b]).double() @ Hs[b].double() * (W[b] - Q[b]).double()).sum(dim=1)
        np.testing.assert_allclose(err[b].cpu().numpy(), want.cpu().numpy(), rtol=1e-5)
        np.testing.assert_allclose(err[b].cpu().numpy(), eng.row_errors(W[b], Q[b], Hs[b]).cpu().numpy(), rtol=4e-6)
    # argument checks of the batch forms
    assert amd.lib.lib.slk_gptq_quantize_batch(W.data_ptr(), None, order.data_ptr(), U.data_ptr(), 2, 100, n, 8, -1.0, 1.0, None, 32, 8, 0,
                                               Q.data_ptr(), None, None, None, 0, None) == amd.lib.E_ARG
    assert b"multiple of 64" in amd.lib.lib.slk_last_error()


@pytest.mark.parametrize("B,rows,padded,cols", [(1, 96, 128, 768), (48, 96, 128, 768), (70, 5, 128, 33), (3, 128, 128, 1024), (4, 0, 128, 64)])
def test_stacked_row_shards(amd, B, rows, padded, cols):
    """slk_stack_rows: the shards of a batch of layers, each its own tensor, padded to whole tiles -- ragged shards, more
    layers than one pointer table holds, whole tiles, no rows at all; weights (2-D) and scales (1-D)."""
    g = torch.Generator().manual_seed(B * 1000 + rows)
    full = [torch.randn((rows + 7, cols), generator=g).cuda() for _ in range(B)]
    parts = [t[3:3 + rows] for t in full]  # row slices of larger tensors, as dist hands them over
    out = amd.engine.stack_rows(parts, padded, 0.0)
    want = torch.zeros((B, padded, cols), device="cuda")
    for b in range(B):
        want[b, :rows] = parts[b]
    assert torch.equal(out, want)
    scales = [t[3:3 + rows, 0].contiguous() for t in full]
    out1 = amd.engine.stack_rows(scales, padded, 1.0)
    want1 = torch.ones((B, padded), device="cuda")
    for b in range(B):
        want1[b, :rows] = scales[b]
    assert torch.equal(out1, want1)
    with pytest.raises(ValueError):
        amd.engine.stack_rows([full[0][:, :5]], padded)  # not contiguous


@pytest.mark.parametrize("n", [1, 5, 63, 64, 65, 130, 516, 1100])
def test_symmetry_verdict_at_the_edges(amd, n):
    """The layer error halves its work when H is bit-wise symmetric: a single asymmetric entry anywhere --
    corners, tile seams, the ragged last tile -- must send it down the general route (obq.py:89-95 takes any H)."""
    rng = np.random.default_rng(n)
    R = 9
    W = rng.standard_normal((R, n)).astype(np.float32)
    Q = (W + 0.3 * rng.standard_normal((R, n))).astype(np.float32)
    X = rng.standard_normal((n + 3, n)).astype(np.float32)
    H = (X.T @ X).astype(np.float32)
    H = ((H + H.T) * np.float32(0.5)).astype(np.float32)
    D = (W - Q).astype(np.float64)
    np.testing.assert_allclose(amd.obq.channelwise_error(W, Q, H), ((D @ H.astype(np.float64)) * D).sum(axis=1), rtol=1e-5)
    spots = {(0, n - 1), (n - 1, 0), (n // 2, n - 1), (min(63, n - 1), min(64, n - 1)), (min(64, n - 1), 0), (n - 1, n // 3)}
    for i, j in sorted(spots):
        if i == j:
            continue
        Ha = H.copy()
        Ha[i, j] += np.float32(n) * (np.float32(1.0) + abs(Ha[i, j]))  # large enough to show in every row's error
        want = ((D @ Ha.astype(np.float64)) * D).sum(axis=1)
        np.testing.assert_allclose(amd.obq.channelwise_error(W, Q, Ha), want, rtol=1e-5, atol=1e-5 * np.abs(want).max(), err_msg=str((i, j)))


@pytest.mark.parametrize("n", [768, 1024, 1536, 2048, 3072, 4096])
def test_scale_search_regular_tree(amd, slkopt, n):
    """Row lengths whose NumPy summation tree is regular (2^k leaves of <= 128 elements) take the chain-per-thread
    search kernel: same scales, bit for bit, as NumPy (the oracle) and as the general kernel."""
    R = 12
    g = grid.UniformGrid(4, -1, 1)
    cb = amd.codebook.UniformCodebook(4, -1, 1)
    W = synth.make_weights(R, n, 7000 + n)
    hd = (np.abs(synth.normal_grid(7000 + n, 8, 1, n)[0]) * 3 + 0.1).astype(np.float32)
    for H in (None, hd):
        got = amd.scaling.compute_min_mse_scaling(W, cb, H=H, grid_size=37)
        assert np.array_equal(got, scaling_ref.best_grid_scale(W, g, H=H, grid_size=37)), (n, H is None)
        slkopt.setenv("SLK_NO_REGULAR_SEARCH", "1")
        assert np.array_equal(got, amd.scaling.compute_min_mse_scaling(W, cb, H=H, grid_size=37))
        slkopt.delenv("SLK_NO_REGULAR_SEARCH")


def test_scale_search_fast_division(amd, slkopt):
    """The regular-tree search replaces its three float32 divisions per element by Markstein's fma sequence where
    that is exact: same choices as with true divides and as NumPy, on rows with a wide dynamic range, zeros,
    denormals, and divisors whose significand is all ones (sent to the true divide)."""
    rng = np.random.default_rng(5)
    R, n = 64, 1024
    W = (rng.standard_normal((R, n)) * np.exp(rng.uniform(-12, 6, (R, 1)))).astype(np.float32)
    W[0, :16] = [0.0, -0.0, 1e-45, -1e-45, 1e-39, 3e-38, 1e-30, -1e-31, 1e30, -3e37, 1.0, -1.0, 0.5, 2.0, 1e-20, 1e20]
    W[1] *= np.float32(1e-25)
    W[2] *= np.float32(1e18)
    hd = (np.abs(rng.standard_normal(n)) + 0.01).astype(np.float32)
    for levels, lo, hi in ((8, -1.0, 1.0), (4, -1.0, 1.0), (3, -1.0, 1.0), (16, -0.75, 1.25)):
        g = grid.UniformGrid(levels, lo, hi)
        cb = amd.codebook.UniformCodebook(levels, lo, hi)
        for H in (None, hd):
            got = amd.scaling.compute_min_mse_scaling(W, cb, H=H, grid_size=150)
            slkopt.setenv("SLK_NO_FAST_SEARCH_DIV", "1")
            slow = amd.scaling.compute_min_mse_scaling(W, cb, H=H, grid_size=150)
            slkopt.delenv("SLK_NO_FAST_SEARCH_DIV")
            assert np.array_equal(got, slow), (levels, H is None)
            with np.errstate(over="ignore"):  # the 1e30 entries square to inf, in NumPy as on the GPU
                want = scaling_ref.best_grid_scale(W, g, H=H, grid_size=150)
            assert np.array_equal(got, want), (levels, H is None)
