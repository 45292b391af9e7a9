"""The CPU oracle against the golden fixtures produced by the real reference.

These tests pin the oracle (tests/golden/make_golden.py ran Coloquinte/sleekit
itself); they run without a GPU.
"""

import hashlib

import numpy as np
import pytest

from conftest import parse_case
from oracle import codebook_fit as fit
from oracle import grid, npsum, obq_ref, scaling_ref, stats_ref
from sleekit_amd import synth

_layers = {}


def layer(R, n, seed, **kw):
    key = (R, n, seed, tuple(sorted(kw.items())))
    if key not in _layers:
        _layers[key] = synth.make_layer(R, n, seed, **kw)
    return _layers[key]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def oracle_run(L, c):
    g = grid.UniformGrid(c["levels"], -1, 1)
    H = obq_ref.strip_input_mean(L["H"], L["mean"]) if c["strip"] else L["H"]
    out = scaling_ref.quantize_scaled(L["W"], L["scale"], g, H, c["order"], c["damp"], c["moves"])
    idx = g.index(scaling_ref.divide_rows(out, L["scale"], 0))
    return out, idx, obq_ref.row_errors(L["W"], out, H), obq_ref.mean_error(L["W"], out, H)


def test_generator_is_stable(small_cases):
    """The integer-hash generator reproduces the inputs the fixtures were made from."""
    for key in small_cases.files:
        if not key.startswith("inputs_"):
            continue
        tag = key.split("/")[0].split("_")
        R, n, seed = int(tag[1][1:]), int(tag[2][1:]), int(tag[3][1:])
        L = layer(R, n, seed)
        assert [sha(L["W"]), sha(L["H"]), sha(L["mean"]), sha(L["scale"])] == list(small_cases[key])


def test_small_cases_bit_exact(small_cases):
    names = [str(x) for x in small_cases["names"]]
    assert len(names) > 90
    for name in names:
        c = parse_case(name)
        L = layer(c["R"], c["n"], c["seed"])
        out, idx, rows, err = oracle_run(L, c)
        assert idx.dtype == np.uint8
        assert np.array_equal(idx, small_cases[name + "/idx"]), name
        assert np.array_equal(rows, small_cases[name + "/row_err"]), name
        assert np.float32(err) == small_cases[name + "/err"], name


def test_local_search_records_match_the_reference(ls_traces):
    """ls_traces.npz holds the REAL reference's sequence of local-search moves (its own LocalSearchQuantizer driven
    move by move) for the rows that came near a tie.  The oracle's search, instrumented the same way, must take
    the same moves with the same margins: this pins the records the GPU parity tests rely on."""
    names = [str(x) for x in ls_traces["names"]]
    assert len(names) >= 50
    limit = int(ls_traces["near_tie_limit"])
    seen = 0
    for name in names:
        c = parse_case(name)
        if c["R"] * c["n"] > 256 * 768:
            continue  # BASELINE-sized: seconds to minutes each on the CPU; the GPU suite holds their hashes
        L = layer(c["R"], c["n"], c["seed"])
        g = grid.UniformGrid(c["levels"], -1, 1)
        Ws = scaling_ref.divide_rows(L["W"], L["scale"], 0)
        Q0 = obq_ref.quantize_layer(Ws, L["H"], g, c["order"], c["damp"], 0)
        records = []
        obq_ref.local_search(Ws.astype(np.float32), Q0, L["H"].astype(np.float32), g, c["moves"], records)
        s = obq_ref.near_tie_summary(records, limit)
        for k in ("rows", "choice", "runner"):
            assert np.array_equal(s[k], ls_traces[f"{name}/{k}"]), (name, k)
        assert np.array_equal(s["ratio"], ls_traces[f"{name}/ratio"]), name
        seen += 1
    assert seen >= 50


@pytest.mark.parametrize("levels", [2, 3, 4, 8, 16, 256, "asym"])
def test_uniform_grid_known_answers(pieces, levels):
    x = pieces["cb/x"]
    g = grid.UniformGrid(5, -0.75, 1.25) if levels == "asym" else grid.UniformGrid(levels, -1, 1)
    tag = "cb/asym" if levels == "asym" else f"cb/N{levels}"
    for name, fn in (("value", g.value), ("index", g.index), ("up", g.up), ("down", g.down)):
        got = fn(x.copy())
        want = pieces[f"{tag}/{name}"]
        assert got.dtype == want.dtype
        assert np.array_equal(got, want), (levels, name)


def test_trace_tiny_layer(pieces):
    """Order, factor, and the full Q/E trace of a 3-level recursion (min_block 4, 2 blocks)."""
    L = layer(8, 16, 2000)
    g = grid.UniformGrid(8, -1, 1)
    Ws = scaling_ref.divide_rows(L["W"], L["scale"], 0)
    Hd = L["H"] + 0.01 * L["H"].diagonal().mean() * np.eye(16)
    assert np.array_equal(Hd.diagonal(), pieces["trace/Hd_diag"])
    order = obq_ref.column_order(Ws, Hd, g, "diag")
    assert np.array_equal(order, pieces["trace/order"])
    U = obq_ref.inverse_factor_upper(Hd[order][:, order])
    assert np.array_equal(U, pieces["trace/U"])
    assert np.allclose(np.linalg.inv(U.T @ U), Hd[order][:, order], rtol=1e-9)
    # LAPACK's general inverse leaves ~1e-17 dust below the diagonal; the loop never reads it
    assert np.abs(np.tril(U, -1)).max() < 1e-12
    Q = Ws[:, order].copy()
    E = np.zeros_like(Q)
    obq_ref.run_schedule(Q, E, U, g, obq_ref.block_schedule(16, 4, 2))
    assert np.array_equal(Q, pieces["trace/Q"])
    assert np.array_equal(E, pieces["trace/E"])


def test_dead_columns_and_mean_removal(pieces):
    L = layer(32, 64, 2020, dead=(3, 17, 40))
    H, W = L["H"].copy(), L["W"].copy()
    assert (H.diagonal() == 0).sum() == 3
    obq_ref.patch_dead_columns(H, W)
    assert np.array_equal(H, pieces["dead/H"]) and np.array_equal(W, pieces["dead/W"])
    assert np.array_equal(obq_ref.strip_input_mean(L["H"], L["mean"]), pieces["strip/H"])
    g = grid.UniformGrid(8, -1, 1)
    out = scaling_ref.quantize_scaled(W, L["scale"], g, H)
    assert np.array_equal(g.index(scaling_ref.divide_rows(out, L["scale"], 0)), pieces["dead/idx"])
    assert np.float32(obq_ref.mean_error(W, out, H)) == pieces["dead/err"]


def test_scale_helpers(pieces):
    L = layer(64, 96, 2001)
    g = grid.UniformGrid(8, -1, 1)
    assert np.array_equal(scaling_ref.no_clip_scale(L["W"], g, 0), pieces["scale/noclip"])
    assert np.array_equal(scaling_ref.norm_scale(L["W"], 0), pieces["scale/norm"])
    assert np.array_equal(scaling_ref.divide_rows(L["W"], L["scale"], 0), pieces["scale/apply"])
    assert np.array_equal(scaling_ref.quantize_scaled(L["W"], L["scale"], g), pieces["scale/rtn"])
    for mode in ("mse", "diag", "hessian", "diag3", "hessian1"):
        got = scaling_ref.pick_scale(L["W"], g, L["H"], mode=mode, grid_size=20)
        assert np.array_equal(got, pieces[f"scale/search_{mode}"]), mode
    got = scaling_ref.pick_scale(L["W"], g, L["H"], mode="obq", grid_size=10)
    assert np.array_equal(got, pieces["scale/search_obq"])


def test_gains(pieces):
    L = layer(64, 96, 2001)
    g = grid.UniformGrid(8, -1, 1)
    Ws = scaling_ref.divide_rows(L["W"], L["scale"], 0)
    Q0 = g(Ws)
    assert np.array_equal(obq_ref.flip_gains(Ws, Q0, L["H"], g.up(Q0)), pieces["gain/up"])
    assert np.array_equal(obq_ref.flip_gains(Ws, Q0, L["H"], g.down(Q0)), pieces["gain/down"])


def test_inverse_diagonal_orders_match_reference(orders):
    """obq.py:70-75, inv_diag / combined_diag: the reference's orders and the layers it quantized in them
    (tests/golden/orders.npz, made by make_golden.py --orders from the imported reference)."""
    g = grid.UniformGrid(8, -1, 1)
    for R, n, seed in ((64, 96, 2001), (96, 172, 2003)):
        L = layer(R, n, seed)
        Hd = L["H"] + 0.01 * L["H"].diagonal().mean() * np.eye(n)
        for mode in ("inv_diag", "combined_diag"):
            tag = f"{mode}/r{R}_n{n}_s{seed}"
            assert np.array_equal(obq_ref.column_order(L["W"], Hd, None, mode), orders[tag + "/order"]), tag
            got = scaling_ref.quantize_scaled(L["W"], L["scale"], g, L["H"], mode, 0.01, 0)
            assert np.array_equal(got, orders[tag + "/out"]), tag


def oracle_namespace():
    """The oracle under the reference's names: what `from sleekit.codebook / obq / scaling import *` gives an experiment."""
    return dict(
        np=np, UniformCodebook=grid.UniformGrid, remove_dead_values=obq_ref.patch_dead_columns, remove_input_bias=obq_ref.strip_input_mean,
        quantization_error=lambda W, Q, H: obq_ref.mean_error(W, Q, H),
        compute_scaling=lambda W, cb, H, mode="mse", axis=0, min_factor=0.05, max_factor=1.0, grid_size=100:
        scaling_ref.pick_scale(W, cb, H, mode=mode, axis=axis, min_factor=min_factor, max_factor=max_factor, grid_size=grid_size),
        compute_min_mse_scaling=scaling_ref.best_grid_scale,
        compute_obq_scaling=lambda W, cb, axis, H, damp=0.01, act_order="diag", min_factor=0.05, max_factor=1.0, grid_size=100:
        scaling_ref.best_obq_scale(W, cb, axis, H, damp=damp, order_mode=act_order, min_factor=min_factor, max_factor=max_factor, grid_size=grid_size),
        quantize_with_scaling=lambda W, sc, cb, H=None, act_order="diag", damp=0.01, nb_ls_moves=0:
        scaling_ref.quantize_scaled(W, sc, cb, H, act_order, damp, nb_ls_moves),
    )


def test_experiment_scripts_replayed_over_the_oracle(experiments, tmp_path):
    """The call sequences of the reference's experiment scripts (tests/experiment_replays.py) over the oracle give the layer
    errors the scripts themselves printed (tests/golden/experiments.json): exactly -- the oracle IS the reference's
    arithmetic on the same BLAS -- so every column is held to 1e-5, the choice-ranked ones too."""
    import experiment_replays

    experiment_replays.check_against_fixture(experiments, oracle_namespace(), str(tmp_path), loose_rtol=1e-5)


def test_table_codebook_matches_reference(pieces):
    """codebook.py:98-190: the general codebook (np.digitize on the bin limits), maps and whole layers."""
    x = pieces["cb/x"]
    for tag in ("nf4", "odd"):
        g = grid.TableGrid(pieces[f"cbt/{tag}/values"], pieces[f"cbt/{tag}/limits"])
        for k in ("value", "index", "up", "down"):
            got = getattr(g, "quantize_" + k)(x)
            assert got.dtype == pieces[f"cbt/{tag}/{k}"].dtype and np.array_equal(got, pieces[f"cbt/{tag}/{k}"]), (tag, k)
    assert np.array_equal(grid.TableGrid.nf4().values, pieces["cbt/nf4/values"])
    assert np.array_equal(grid.TableGrid.nf4().limits, pieces["cbt/nf4/limits"])
    for R, n, seed in ((64, 96, 2001), (96, 172, 2003)):
        L = layer(R, n, seed)
        for order, moves in (("diag", 0), ("sqerr", 0), ("err", 10), ("diag", 10)):
            want = pieces[f"cbt/nf4/layer_r{R}_n{n}_s{seed}_{order}_ls{moves}"]
            got = scaling_ref.quantize_scaled(L["W"], L["scale"], grid.TableGrid.nf4(), L["H"], order, 0.01, moves)
            assert np.array_equal(got, want), (n, order, moves)


def test_pivot_order_matches_reference(pieces):
    """obq.py:140-166: the greedy pivoted-Cholesky order, restated without the trailing matrix."""
    for tag, (R, n, seed) in (("96", (64, 96, 2001)), ("256", (32, 256, 2050))):
        L = layer(R, n, seed)
        Hd = L["H"].astype(np.float32) + np.float32(0.01 * L["H"].astype(np.float32).diagonal().mean()) * np.eye(n)
        assert np.array_equal(obq_ref.pivot_order(Hd), pieces[f"pivot/{tag}/order"])
        assert np.array_equal(obq_ref.column_order(L["W"], Hd, None, "pivot"), pieces[f"pivot/{tag}/order"])


def test_running_stats(pieces):
    X = pieces["stats/X"]
    st = stats_ref.RunningStats(48)
    for a, b in ((0, 64), (64, 72), (72, 200)):
        st.add_tokens(X[a:b].reshape(1, b - a, 48))
    assert st.count == int(pieces["stats/count"])
    np.testing.assert_allclose(st.hessian, pieces["stats/H"], rtol=2e-6, atol=1e-6)
    np.testing.assert_allclose(st.mean, pieces["stats/mean"], rtol=2e-6, atol=1e-7)


def test_numpy_sum_model():
    """oracle.npsum states NumPy's float32 summation order; the device kernel follows the model."""
    rng = np.random.default_rng(5)
    for n in (1, 7, 8, 9, 127, 128, 129, 768, 1100, 4096, 8192, 8193, 11008):
        d = np.square(rng.standard_normal(n).astype(np.float32)) * 3
        assert npsum.pairwise_sum_f32(d) == np.add.reduce(d)
        assert npsum.mean_f32(d) == d.mean()
        M = np.zeros((n, n), np.float32) if n <= 4096 else None
        if M is not None:
            np.fill_diagonal(M, d)
            assert npsum.mean_f32(d) == M.diagonal().mean()


def test_schedule_shapes():
    """Leaf/update structure the survey measured for the BASELINE shapes."""
    def summary(n):
        ops = obq_ref.block_schedule(n)
        leaves = [b - a for k, a, b, _ in ops if k == obq_ref.LEAF]
        ks = sorted({b - a for k, a, b, _ in ops if k == obq_ref.UPDATE})
        return leaves, ks, sum(1 for o in ops if o[0] == obq_ref.UPDATE)
    leaves, ks, _ = summary(768)
    assert leaves == [32] * 24 and ks == [32, 96]
    leaves, ks, nupd = summary(4096)
    assert leaves == [32] * 128 and ks == [32, 64, 512]
    leaves, ks, nupd = summary(11008)
    assert set(leaves) == {32, 12} and ks == [12, 32, 172, 1376] or ks == [32, 172, 1376]
    assert sum(leaves) == 11008
    # every column is quantized exactly once, in order
    for n in (1, 31, 32, 33, 100, 1100, 3072):
        ops = obq_ref.block_schedule(n)
        cols = [c for k, a, b, _ in ops if k == obq_ref.LEAF for c in range(a, b)]
        assert cols == list(range(n))


def test_large_cases_sample(large_cases):
    """One BASELINE-sized layer end to end against the reference's index hash (seconds on CPU)."""
    c = next(c for c in large_cases if (c["R"], c["n"]) == (768, 768))
    L = layer(c["R"], c["n"], c["seed"])
    assert sha(L["W"]) == c["sha_W"] and sha(L["H"]) == c["sha_H"] and sha(L["scale"]) == c["sha_scale"]
    out, idx, rows, err = oracle_run(L, dict(levels=c["levels"], order=c["order"], damp=c["damp"],
                                             moves=c["moves"], strip=c["strip_mean"]))
    assert sha(idx) == c["sha_idx"]
    assert np.float32(err).tobytes().hex() == c["err_f32_hex"]


# ---------------------------------------------------------------------------------------------------
# codebook training (oracle/codebook_fit.py against the reference's own outputs, bit for bit)
def test_codebook_fit_matches_the_reference(codebook_fit):
    import json

    z = codebook_fit
    for name, count, seed, dtype, size, lam in json.loads(str(z["cases"])):
        data = synth.make_samples(count, seed, np.dtype(dtype).type)
        start = fit.equal_mass(data, size)
        assert np.array_equal(start.values, z[f"{name}/start_values"]) and np.array_equal(start.limits, z[f"{name}/start_limits"])
        for k in (1, 2, 3):
            g = fit.fit_lloyd_max(data, size, lam, max_iter=k)
            assert g.values.dtype == z[f"{name}/round{k}_values"].dtype and g.limits.dtype == z[f"{name}/round{k}_limits"].dtype
            assert np.array_equal(g.values, z[f"{name}/round{k}_values"]), (name, k)
            assert np.array_equal(g.limits, z[f"{name}/round{k}_limits"]), (name, k)
        g = fit.fit_lloyd_max(data, size, lam)
        assert np.array_equal(g.values, z[f"{name}/final_values"]) and np.array_equal(g.limits, z[f"{name}/final_limits"]), name
        assert np.array_equal(fit.bin_shares(g, data), z[f"{name}/final_shares"])
        assert fit.code_entropy(g, data) == z[f"{name}/final_entropy"]
        mse = fit.mean_square_miss(g, data)
        assert mse.dtype == z[f"{name}/final_mse"].dtype and mse == z[f"{name}/final_mse"]


def test_codebook_fit_drawn_starts_and_empty_bins(codebook_fit):
    z = codebook_fit
    data = synth.make_samples(20000, 34, np.float32)
    np.random.seed(11)
    g = fit.pick_random(data, 8)
    assert np.array_equal(g.values, z["random/values"]) and np.array_equal(g.limits, z["random/limits"])
    np.random.seed(12)
    g = fit.fit_lloyd_max(data, 8, random_init=True, sample_count=500)
    assert np.array_equal(g.values, z["random_fit/values"]) and np.array_equal(g.limits, z["random_fit/limits"])
    g = grid.TableGrid([-50.0, -40.0, -0.5, 0.0, 0.25, 0.5, 30.0, 40.0, 50.0])
    assert np.array_equal(fit.bin_centres(g, data), z["empty/centroids"])
    fit.drop_empty_bins(g, data)
    assert np.array_equal(g.values, z["empty/kept_values"]) and np.array_equal(g.limits, z["empty/kept_limits"])
    nf4 = grid.TableGrid.nf4()
    assert np.array_equal(fit.bin_shares(nf4, data / 4), z["nf4/shares"])
    assert fit.mean_square_miss(nf4, data / 4) == z["nf4/mse"]
    assert fit.code_entropy(nf4, data / 4) == z["nf4/entropy"]
    assert np.array_equal(fit.bin_centres(nf4, data / 4), z["nf4/centroids"])
