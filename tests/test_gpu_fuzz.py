"""Differential fuzzing of the device path against the oracle: odd shapes, every order, both kinds of
codebook, local search, unusual blockings -- small enough for the NumPy oracle to finish in seconds."""
import numpy as np
import pytest

from ls_evidence import explain_rows
from oracle import grid, obq_ref, scaling_ref

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def amd():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import sleekit_amd
    from sleekit_amd import codebook, obq, scaling

    class NS:
        pass

    ns = NS()
    from sleekit_amd import engine

    ns.codebook, ns.obq, ns.scaling, ns.engine = codebook, obq, scaling, engine
    return ns


def _case(seed):
    rng = np.random.default_rng(seed)
    R = int(rng.choice([1, 2, 3, 15, 16, 17, 31, 33, 48, 70]))
    n = int(rng.choice([1, 2, 5, 16, 31, 32, 33, 47, 64, 65, 96, 100, 130, 172, 200, 257]))
    T = 2 * n + 8
    X = rng.standard_normal((T, n)) * (0.5 + 2.0 * rng.random(n))
    X[:, : min(n, 3)] += rng.standard_normal((T, 1)) * 3.0  # correlated outlier channels
    H = (X.T @ X / T).astype(np.float32)
    H = ((H + H.T) * np.float32(0.5)).astype(np.float32)
    W = (rng.standard_normal((R, n)) * 0.05).astype(np.float32)
    scale = (np.abs(W).max(axis=1) * np.float32(0.55) + np.float32(1e-6)).astype(np.float32)
    order = str(rng.choice(["diag", "none", "err", "sqerr"]))
    moves = int(rng.choice([0, 0, 3]))
    damp = float(rng.choice([0.01, 0.03, 0.1]))
    kind = str(rng.choice(["uniform", "uniform", "table"]))
    levels = int(rng.choice([2, 3, 4, 8, 16]))
    return dict(W=W, H=H, scale=scale, order=order, moves=moves, damp=damp, kind=kind, levels=levels)


@pytest.mark.parametrize("seed", range(40))
def test_layer_against_oracle(amd, seed):
    c = _case(seed)
    if c["kind"] == "uniform":
        g, cb = grid.UniformGrid(c["levels"], -1, 1), amd.codebook.UniformCodebook(c["levels"], -1, 1)
    else:
        vals = np.sort(np.random.default_rng(seed + 1000).uniform(-1, 1, c["levels"])).astype(np.float32)
        vals[0], vals[-1] = -1.0, 1.0
        if (np.diff(vals) <= 0).any():
            vals = np.linspace(-1, 1, c["levels"]).astype(np.float32)
        g, cb = grid.TableGrid(vals), amd.codebook.Codebook(vals)
    records = []
    want = scaling_ref.quantize_scaled(c["W"], c["scale"], g, c["H"], c["order"], c["damp"], c["moves"], ties="stable", ls_records=records)
    if c["moves"] == 0:
        got = amd.scaling.quantize_with_scaling(c["W"], c["scale"], cb, c["H"], act_order=c["order"], damp=c["damp"], nb_ls_moves=0)
        assert np.array_equal(got, want), {k: v for k, v in c.items() if k not in ("W", "H", "scale")}
    else:
        # local search: bit-equal, or -- row by row -- a PROVEN near-tie of the oracle's own decision (ls_evidence.py)
        import torch

        W, H, sc = (torch.from_numpy(c[k]).cuda() for k in ("W", "H", "scale"))
        res = amd.engine.quantize_layer(W, H, cb, sc, c["order"], c["damp"], c["moves"], want_ls_trace=True)
        got = res.Q.cpu().numpy()
        bad = np.flatnonzero((got != want).any(axis=1))
        explain_rows(bad, res.ls_trace.cpu().numpy(), obq_ref.near_tie_summary(records, 64.0))
    e_got = amd.obq.quantization_error(c["W"], got, c["H"])
    e_want = obq_ref.mean_error(c["W"].astype(np.float32), want, c["H"])
    assert abs(float(e_got) - float(e_want)) <= 1e-5 * abs(float(e_want)) + 1e-12 or (c["moves"] > 0 and len(bad))


@pytest.mark.parametrize("seed", range(8))
def test_blockings_against_oracle(amd, seed):
    rng = np.random.default_rng(500 + seed)
    R, n = int(rng.choice([5, 16, 40])), int(rng.choice([33, 64, 100, 172, 192, 300]))
    mb, nb = int(rng.choice([1, 2, 8, 16, 24, 32, 48, 64])), int(rng.choice([2, 3, 4, 8]))
    W = (rng.standard_normal((R, n)) * 0.6).astype(np.float32)
    U = np.triu(rng.standard_normal((n, n)) * (0.3 / np.sqrt(n))) + np.diag(1.0 + rng.random(n))
    cb, g = amd.codebook.UniformCodebook(8, -1, 1), grid.UniformGrid(8, -1, 1)
    Q0, E0 = W.copy(), np.zeros_like(W)
    obq_ref.run_schedule(Q0, E0, U, g, obq_ref.block_schedule(n, mb, nb))
    Q1, E1 = W.copy(), np.zeros_like(W)
    amd.obq._quantize_opt_block(Q1, E1, U, cb, mb, nb)
    assert np.array_equal(Q1, Q0), (R, n, mb, nb)
    np.testing.assert_allclose(E1, E0, rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("seed", range(6))
def test_batched_rounds_against_single_layers(amd, seed):
    """Random rounds of same-shaped small layers through the batched factorisation + stacked loop of one rank
    (sleekit_amd.dist, slk_*_batch) against the same layers one by one: bit-equal values and indices, whatever the shape
    (ragged rows, widths that are not a multiple of the factor's 64-column tiles), the codebook size and the damping."""
    import torch

    from sleekit_amd import dist as sdist

    rng = np.random.default_rng(900 + seed)
    B = int(rng.choice([2, 3, 5, 8]))
    R = int(rng.choice([7, 16, 50, 128, 130]))
    n = int(rng.choice([33, 64, 100, 172, 200, 320]))
    levels = int(rng.choice([2, 3, 4, 8, 16]))
    damp = float(rng.choice([0.01, 0.03, 0.1]))
    moves = int(rng.choice([0, 0, 4]))
    order = str(rng.choice(["diag", "none"]))
    cb = amd.codebook.UniformCodebook(levels, -1, 1)
    layers = []
    for b in range(B):
        T = 2 * n + 8
        X = rng.standard_normal((T, n)) * (0.5 + 2.0 * rng.random(n))
        H = (X.T @ X / T).astype(np.float32)
        H = ((H + H.T) * np.float32(0.5)).astype(np.float32)
        W = (rng.standard_normal((R, n)) * 0.05).astype(np.float32)
        scale = (np.abs(W).max(axis=1) * np.float32(0.55) + np.float32(1e-6)).astype(np.float32)
        layers.append({k: torch.from_numpy(v).cuda() for k, v in (("W", W), ("H", H), ("scale", scale))})
    be = sdist.HipBackend(cb, order, damp, moves, with_error=True)
    shards = sdist.quantize_stream(layers, be)
    torch.cuda.synchronize()
    for lay, sh in zip(layers, shards):
        res = amd.engine.quantize_layer(lay["W"], lay["H"], cb, lay["scale"], order, damp, moves)
        assert torch.equal(sh["Q"], res.Q) and torch.equal(sh["idx"], res.idx), (B, R, n, levels, damp, moves, order)
        assert int(sh["info"].item()) == 0
