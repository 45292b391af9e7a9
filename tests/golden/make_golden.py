#!/usr/bin/env python3
"""Generate the golden fixtures by running the REAL reference (Coloquinte/sleekit).

Run only in the build container, where the reference is mounted read-only:

    python tests/golden/make_golden.py [--large]

It imports `sleekit` from /root/reference, feeds it layers from the build's own
integer-hash generator (sleekit_amd/synth.py) and stores inputs' hashes and the
reference's outputs:

    small_cases.npz   full outputs (u8 indices, order, row errors, ...) of small layers
    pieces.npz        known-answer vectors of the helper functions on the path
    large_cases.json  SHA-256 of indices + float32 errors of BASELINE-sized layers
    codebook_fit.npz  known answers of the codebook training functions (equiprobable start, Lloyd-Max rounds, final
                      codebooks with and without the entropy term, drawn starts, empty bins)
    ls_traces.npz     for every case with local-search moves: the reference's sequence of moves and how close each
                      decision was, for the rows that came near a tie (oracle/obq_ref.py: move_record), plus a
                      per-row hash of the indices of the large cases -- what lets a parity test PROVE that a row
                      which differs after local search differs because a near-tie fell the other way

Only data is written: inputs and expected outputs.  No reference source text is
copied.  The fixtures record the NumPy/BLAS versions because the float64
promotion in the reference (np.eye -> float64 factor) is NumPy >= 2 behaviour.
"""

import argparse
import hashlib
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

import numpy as np  # noqa: E402

from sleekit_amd import synth  # noqa: E402

import sleekit.obq as ref_obq  # noqa: E402
import sleekit.scaling as ref_scaling  # noqa: E402
from sleekit.codebook import Codebook, UniformCodebook  # noqa: E402


def sha(a):
    a = np.ascontiguousarray(a)
    return hashlib.sha256(a.tobytes()).hexdigest()


def run_reference(L, levels, order, damp, moves, strip_mean=False, want_records=False):
    cb = UniformCodebook(levels, -1, 1)
    W, H, sc = L["W"], L["H"], L["scale"]
    if strip_mean:
        H = ref_obq.remove_input_bias(H, L["mean"])
    out = ref_scaling.quantize_with_scaling(W, sc, cb, H, act_order=order, damp=damp, nb_ls_moves=moves)
    idx = cb.quantize_index(ref_scaling.apply_scaling(out, sc, 0))
    rows = ref_obq.channelwise_error(W, out, H)
    res = dict(out=out, idx=idx, row_err=rows, err=ref_obq.quantization_error(W, out, H), H_used=H)
    if want_records and moves > 0:
        res["records"] = local_search_records(W, H, sc, cb, order, damp, moves, out)
    return res


def local_search_records(W, H, sc, cb, order, damp, moves, expected_out):
    """The reference's local search driven move by move (its own LocalSearchQuantizer, obq.py:234-346, exactly as
    quantize_local_search drives it), reading its gains before every move.  The result must be the one-call
    result bit for bit, or the records describe something else."""
    from oracle import obq_ref  # move_record only inspects arrays: no oracle arithmetic enters the fixture's moves

    quant = ref_scaling.apply_scaling(W, sc, 0)
    Q0 = ref_obq.quantize_opt(quant, H, cb, act_order=order, damp=damp, nb_ls_moves=0)
    W32, H32 = quant.astype(np.float32), H.astype(np.float32)  # what quantize_opt hands to the search (obq.py:195-196, 216)
    ls = ref_obq.LocalSearchQuantizer(W32, Q0, H32, cb)
    noise = obq_ref.gain_noise_scale(W32, Q0, H32)
    records = []
    for _ in range(moves):
        records.append(obq_ref.move_record(ls.gain_up, ls.gain_down, ls.Q_up - ls.Q, ls.Q_down - ls.Q, noise))
        ls.do_move()
    assert np.array_equal(ref_scaling.apply_scaling(ls.Q, 1 / sc, 0), expected_out), "instrumented drive != quantize_with_scaling"
    return records


# Rows whose closest decision came within this many roundings (move_record's ratio) are kept with their full records.
# The worst-case bound on the difference of two float32 GEMMs is ~2 n roundings, but at that width nearly every row
# would qualify (typical margins are 1e3 ... 1e5); errors of real GEMMs are a few roundings, and so is the bound the
# parity tests apply (tests/test_gpu_parity.py: LS_NEAR_TIE).
NEAR_TIE_LIMIT = 64


def row_hashes(idx):
    return np.array([int.from_bytes(hashlib.sha256(np.ascontiguousarray(r).tobytes()).digest()[:8], "little") for r in idx],
                    dtype=np.uint64)


def ls_traces():
    """Local-search evidence for every fixture case with moves (small_cases.npz and large_cases.json alike)."""
    from oracle import obq_ref

    T = {}
    names = []

    def add(name, L, levels, order, damp, moves, strip, large):
        r = run_reference(L, levels, order, damp, moves, strip_mean=strip, want_records=True)
        n = L["W"].shape[1]
        s = obq_ref.near_tie_summary(r["records"], NEAR_TIE_LIMIT)
        for k, v in s.items():
            T[f"{name}/{k}"] = v
        T[f"{name}/row_hash"] = row_hashes(r["idx"])
        T[f"{name}/sha_idx"] = np.array(sha(r["idx"]))
        names.append(name)
        allr = np.stack([x["ratio"] for x in r["records"]], axis=1)
        print(f"{name}: {len(s['rows'])} near-tie rows of {L['W'].shape[0]}, smallest ratio {allr.min():.3g} (n = {n})", flush=True)

    shapes = [(8, 16), (64, 96), (128, 256), (96, 172)]
    seed = 2000
    for R, n in shapes:
        L = synth.make_layer(R, n, seed)
        for levels in (2, 3, 4, 8):
            for order in ("diag", "none", "sqerr"):
                add(f"r{R}_n{n}_s{seed}_N{levels}_{order}_ls10", L, levels, order, 0.01, 10, False, False)
        seed += 1
    L = synth.make_layer(256, 768, 2010)
    for levels in (3, 8):
        add(f"r256_n768_s2010_N{levels}_diag_ls10", L, levels, "diag", 0.01, 10, False, False)
    for R, n, seed, levels, order, damp, moves, strip in LARGE:
        if moves > 0:
            add(f"r{R}_n{n}_s{seed}_N{levels}_{order}_ls{moves}", synth.make_layer(R, n, seed), levels, order, damp, moves, strip, True)
    T["names"] = np.array(names)
    T["near_tie_limit"] = np.int64(NEAR_TIE_LIMIT)
    return T


def small_cases():
    store = {}
    names = []
    shapes = [(8, 16), (64, 96), (128, 256), (96, 172)]
    seed = 2000
    for R, n in shapes:
        L = synth.make_layer(R, n, seed)
        for levels in (2, 3, 4, 8):
            for order in ("diag", "none", "sqerr"):
                for moves in (0, 10):
                    name = f"r{R}_n{n}_s{seed}_N{levels}_{order}_ls{moves}"
                    r = run_reference(L, levels, order, 0.01, moves)
                    store[name + "/idx"] = r["idx"]
                    store[name + "/row_err"] = r["row_err"]
                    store[name + "/err"] = np.float32(r["err"])
                    names.append(name)
        store[f"inputs_r{R}_n{n}_s{seed}/sha"] = np.array(
            [sha(L["W"]), sha(L["H"]), sha(L["mean"]), sha(L["scale"])]
        )
        seed += 1
    # one mid-sized layer, the experiment defaults only
    R, n, seed = 256, 768, 2010
    L = synth.make_layer(R, n, seed)
    for levels in (3, 8):
        for moves in (0, 10):
            name = f"r{R}_n{n}_s{seed}_N{levels}_diag_ls{moves}"
            r = run_reference(L, levels, "diag", 0.01, moves)
            store[name + "/idx"] = r["idx"]
            store[name + "/row_err"] = r["row_err"]
            store[name + "/err"] = np.float32(r["err"])
            names.append(name)
    store[f"inputs_r{R}_n{n}_s{seed}/sha"] = np.array([sha(L["W"]), sha(L["H"]), sha(L["mean"]), sha(L["scale"])])
    # bias-corrected Hessian (cfg3 style), damp 0.03, err ordering
    R, n, seed = 64, 96, 2001
    L = synth.make_layer(R, n, seed)
    for levels, order, damp in ((3, "diag", 0.01), (8, "err", 0.03), (4, "sqerr", 0.03)):
        name = f"r{R}_n{n}_s{seed}_N{levels}_{order}_ls0_strip_d{damp}"
        r = run_reference(L, levels, order, damp, 0, strip_mean=True)
        store[name + "/idx"] = r["idx"]
        store[name + "/row_err"] = r["row_err"]
        store[name + "/err"] = np.float32(r["err"])
        names.append(name)
    store["names"] = np.array(names)
    return store


def pieces():
    """Known-answer vectors of the individual functions on the path."""
    P = {}
    # --- uniform codebook (codebook.py:43-95) on crafted float32 inputs ---
    x = np.concatenate(
        [
            np.linspace(-1.5, 1.5, 4001, dtype=np.float32),
            np.array([-1, 1, 0, -0.0, 1 / 7, 3 / 7, 0.5, -0.5, 1e-8, -1e-8, 0.14285713, 0.14285716], dtype=np.float32),
            (synth.normal_grid(7, 9, 1, 4096)[0] * 0.6).astype(np.float32),
        ]
    )
    P["cb/x"] = x
    for levels in (2, 3, 4, 8, 16, 256):
        cb = UniformCodebook(levels, -1, 1)
        P[f"cb/N{levels}/value"] = cb.quantize_value(x)
        P[f"cb/N{levels}/index"] = cb.quantize_index(x)
        P[f"cb/N{levels}/up"] = cb.quantize_up(x)
        P[f"cb/N{levels}/down"] = cb.quantize_down(x)
    cb = UniformCodebook(5, -0.75, 1.25)  # asymmetric grid
    for k in ("value", "index", "up", "down"):
        P[f"cb/asym/{k}"] = getattr(cb, "quantize_" + k)(x)

    # --- general (table) codebooks (codebook.py:98-190): NF4 and an irregular one with explicit limits ---
    tables = {"nf4": Codebook.nf4(), "odd": Codebook([-1.5, -0.4, -0.1, 0.3, 2.0], [-1.0, -0.2, 0.1, 0.5])}
    for tag, cb in tables.items():
        P[f"cbt/{tag}/values"] = cb.values
        P[f"cbt/{tag}/limits"] = cb.thresholds
        for k in ("value", "index", "up", "down"):
            P[f"cbt/{tag}/{k}"] = getattr(cb, "quantize_" + k)(x)
    # whole layers through quantize_with_scaling with NF4: orders that call the quantizer, and local search
    for R_, n_, seed_ in ((64, 96, 2001), (96, 172, 2003)):
        Lt = synth.make_layer(R_, n_, seed_)
        for order, moves in (("diag", 0), ("sqerr", 0), ("err", 10), ("diag", 10)):
            out = ref_scaling.quantize_with_scaling(Lt["W"], Lt["scale"], Codebook.nf4(), H=Lt["H"], act_order=order, damp=0.01, nb_ls_moves=moves)
            P[f"cbt/nf4/layer_r{R_}_n{n_}_s{seed_}_{order}_ls{moves}"] = out

    # --- greedy pivoted-Cholesky order (obq.py:140-166) on damped float64 Hessians ---
    for tag, (R_, n_, seed_) in (("96", (64, 96, 2001)), ("256", (32, 256, 2050))):
        Lp = synth.make_layer(R_, n_, seed_)
        Hd = Lp["H"].astype(np.float32) + np.float32(0.01 * Lp["H"].astype(np.float32).diagonal().mean()) * np.eye(n_)
        P[f"pivot/{tag}/order"] = ref_obq.compute_hessian_order(Lp["W"], Hd, UniformCodebook(8, -1, 1), "pivot").astype(np.int64)

    # --- Hessian preparation + factor + full debug trace on one tiny layer ---
    L = synth.make_layer(8, 16, 2000)
    W, H = L["W"].copy(), L["H"].copy()
    cb = UniformCodebook(8, -1, 1)
    Ws = ref_scaling.apply_scaling(W, L["scale"], 0)
    Hd = H.astype(np.float32) + 0.01 * H.diagonal().mean() * np.eye(16)
    order = ref_obq.compute_hessian_order(Ws, Hd, cb, "diag")
    U = ref_obq.compute_hessian_chol(Hd[order][:, order])
    Q = Ws[:, order].copy()
    E = np.zeros_like(Q)
    ref_obq._quantize_opt_block(Q, E, U, cb, 4, 2)  # small blocks: exercises 3 recursion levels
    P["trace/order"], P["trace/U"], P["trace/Q"], P["trace/E"] = order, U, Q, E
    P["trace/Hd_diag"] = Hd.diagonal().copy()

    # --- dead columns and input-mean removal (obq.py:14-35) ---
    L = synth.make_layer(32, 64, 2020, dead=(3, 17, 40))
    Hx, Wx = L["H"].copy(), L["W"].copy()
    ref_obq.remove_dead_values(Hx, Wx)
    P["dead/H"], P["dead/W"] = Hx, Wx
    P["strip/H"] = ref_obq.remove_input_bias(L["H"], L["mean"])
    cb = UniformCodebook(8, -1, 1)
    out = ref_scaling.quantize_with_scaling(Wx, L["scale"], cb, Hx)
    P["dead/idx"] = cb.quantize_index(ref_scaling.apply_scaling(out, L["scale"], 0))
    P["dead/err"] = np.float32(ref_obq.quantization_error(Wx, out, Hx))

    # --- scale helpers (scaling.py:21-55) ---
    L = synth.make_layer(64, 96, 2001)
    cb = UniformCodebook(8, -1, 1)
    P["scale/noclip"] = ref_scaling.compute_non_saturating_scaling(L["W"], cb, 0)
    P["scale/norm"] = ref_scaling.compute_norm_scaling(L["W"], 0)
    P["scale/apply"] = ref_scaling.apply_scaling(L["W"], L["scale"], 0)
    P["scale/rtn"] = ref_scaling.quantize_with_scaling(L["W"], L["scale"], cb)
    for mode in ("mse", "diag", "hessian", "diag3", "hessian1"):
        P[f"scale/search_{mode}"] = ref_scaling.compute_scaling(L["W"], cb, L["H"], mode=mode, grid_size=20)
    P["scale/search_obq"] = ref_scaling.compute_scaling(L["W"], cb, L["H"], mode="obq", grid_size=10)
    # the same two searches step by step (the reference's own functions in its own order, scaling.py:98-134, 160-190),
    # keeping every grid point's row errors: what lets a test PROVE that a row whose chosen factor differs sits on a
    # near-tie of the reference's own errors (they come out of a BLAS product, whose summation order is not ours)
    for mode, gs in (("hessian", 20), ("hessian1", 20), ("obq", 10)):
        base = ref_scaling.compute_non_saturating_scaling(L["W"], cb, 0)
        factors = np.linspace(0.05, 1.0, gs, dtype=np.float32)
        errs = []
        if mode == "obq":
            H_opt = L["H"] + 0.01 * L["H"].diagonal().mean() * np.eye(L["H"].shape[0])
            order = ref_obq.compute_hessian_order(ref_scaling.apply_scaling(L["W"], base, 0), H_opt, cb, "diag")
            Wp, Hp = L["W"][:, order], L["H"][order][:, order]
            Hinv = ref_obq.compute_hessian_chol(H_opt[order][:, order])
        for f in factors:
            sc = f * base
            if mode == "obq":
                Q = ref_scaling.apply_scaling(Wp, sc, 0)
                ref_obq._quantize_opt_block(Q, np.zeros_like(Wp), Hinv, cb, min_block_size=32, num_blocks=8)
                errs.append(ref_scaling._compute_mse(Hp, ref_scaling.apply_scaling(Q, 1 / sc, 0) - Wp))
            else:
                Hm = L["H"]
                if mode == "hessian1":  # scaling.py:219-222: the damped matrix is float64 (np.eye)
                    Hm = L["H"] + 0.01 * 1.0 * L["H"].diagonal().mean() * np.eye(L["H"].shape[0])
                errs.append(ref_scaling._compute_mse(Hm, ref_scaling.quantize_with_scaling(L["W"], sc, cb) - L["W"]))
        errs = np.stack(errs)
        pick = factors[np.argmin(errs, axis=0)]  # first minimum, like the reference's strict `<`
        assert np.array_equal(base * pick, P[f"scale/search_{mode}"]), mode
        P[f"scale/search_{mode}_errors"], P[f"scale/search_{mode}_factors"], P[f"scale/search_{mode}_base"] = errs, factors, base

    # --- gains (obq.py:220-231) ---
    Q0 = cb(ref_scaling.apply_scaling(L["W"], L["scale"], 0))
    Ws = ref_scaling.apply_scaling(L["W"], L["scale"], 0)
    P["gain/up"] = ref_obq.compute_gain(Ws, Q0, L["H"], cb.quantize_up(Q0))
    P["gain/down"] = ref_obq.compute_gain(Ws, Q0, L["H"], cb.quantize_down(Q0))

    # --- running statistics (statistics.py:76-87) through the real torch adapter ---
    import torch
    from sleekit import Sleekit

    lin = torch.nn.Linear(48, 10)
    st = Sleekit(lin)
    X = synth.make_activations(200, 48, 2030).astype(np.float32)
    for a, b in ((0, 64), (64, 72), (72, 200)):
        st.add_batch(torch.from_numpy(X[a:b]).reshape(1, b - a, 48))
    P["stats/X"], P["stats/H"], P["stats/mean"] = X, st.hessian.numpy().copy(), st.mean.numpy().copy()
    P["stats/count"] = np.int64(st.count)

    # --- the whole adapter: presets on a Linear layer, unfold + statistics on conv layers ---
    def seeded_linear(n_in, n_out, seed):
        lin = torch.nn.Linear(n_in, n_out)
        with torch.no_grad():
            lin.weight.copy_(torch.from_numpy(synth.make_weights(n_out, n_in, seed)))
            lin.bias.copy_(torch.from_numpy((0.1 * synth.normal_grid(seed, 11, 1, n_out)[0]).astype(np.float32)))
        return lin

    Xa = synth.make_activations(300, 40, 2031).astype(np.float32)
    P["adapter/X"] = Xa
    for preset, bits in (("basic", 4), ("sleekit_light", 3), ("sleekit_heavy", 3)):
        lin = seeded_linear(40, 24, 2032)
        st = Sleekit(lin)
        st.add_batch(torch.from_numpy(Xa[:128]))
        st.add_batch(torch.from_numpy(Xa[128:]).reshape(2, 86, 40))
        # the statistics the reference holds before it quantizes (the same for the three presets): with THESE on the
        # device the quantized weights must be the reference's bit for bit where no GEMM-ranked choice is involved
        P["adapter/H_ref"], P["adapter/mean_ref"], P["adapter/count_ref"] = st.hessian.numpy().copy(), st.mean.numpy().copy(), np.int64(st.count)
        getattr(st, "quantize_" + preset)(bits)
        P[f"adapter/{preset}/weight"] = lin.weight.detach().numpy().copy()
        P[f"adapter/{preset}/bias"] = lin.bias.detach().numpy().copy()
    xc2 = torch.from_numpy(synth.make_activations(4 * 6 * 7 * 7, 1, 2033).astype(np.float32).reshape(4, 6, 7, 7))
    c2 = torch.nn.Conv2d(6, 5, 3, padding=1, stride=2)
    st = Sleekit(c2)
    st.add_batch(xc2)
    st.add_batch(xc2[0])
    P["adapter/conv2d/x"], P["adapter/conv2d/H"], P["adapter/conv2d/mean"] = xc2.numpy(), st.hessian.numpy().copy(), st.mean.numpy().copy()
    P["adapter/conv2d/count"] = np.int64(st.count)
    xc1 = torch.from_numpy(synth.make_activations(3 * 6 * 11, 1, 2034).astype(np.float32).reshape(3, 6, 11))
    c1 = torch.nn.Conv1d(6, 5, 3, dilation=2)
    st = Sleekit(c1)
    st.add_batch(xc1)
    st.add_batch(xc1[1])
    P["adapter/conv1d/x"], P["adapter/conv1d/H"], P["adapter/conv1d/mean"] = xc1.numpy(), st.hessian.numpy().copy(), st.mean.numpy().copy()
    P["adapter/conv1d/count"] = np.int64(st.count)
    return P


LARGE = [
    # (R, n, seed, levels, order, damp, moves, strip_mean)
    (768, 768, 1000, 8, "diag", 0.01, 0, False),  # cfg1 / OPT-125M attention
    (3072, 768, 1001, 8, "diag", 0.01, 0, False),  # OPT-125M fc1
    (768, 3072, 1002, 8, "diag", 0.01, 0, False),  # OPT-125M fc2
    (1024, 1024, 1003, 3, "diag", 0.01, 0, True),  # cfg3: 1.5 bit + bias-corrected H
    (1024, 4096, 1004, 3, "diag", 0.01, 0, True),
    (1024, 1024, 1005, 8, "diag", 0.01, 10, False),  # cfg4: 3 bit + 10 moves
    (4096, 1024, 1006, 8, "diag", 0.01, 10, False),
    (4096, 4096, 1007, 8, "diag", 0.01, 0, False),  # headline
    (512, 11008, 1008, 4, "diag", 0.01, 0, False),  # cfg5 row shard (1/8 of 4096 rows)
    (3072, 1024, 1009, 8, "diag", 0.01, 10, False),  # cfg4 (BLOOM-560M qkv) + 10 moves
    (1024, 4096, 1010, 8, "diag", 0.01, 10, False),  # cfg4 (BLOOM-560M 4h->h) + 10 moves
    (4096, 4096, 1011, 8, "diag", 0.01, 10, False),  # headline shape + 10 moves
    (4096, 11008, 1012, 4, "diag", 0.01, 0, False),  # cfg5: a whole Llama-FFN layer, 2 bit (a minute of reference time)
    (4096, 1024, 1013, 3, "diag", 0.01, 0, True),    # cfg3: OPT-350M fc1, 1.5 bit + bias-corrected Hessian
]


def large_cases(selected=None):
    out = []
    for spec in LARGE:
        R, n, seed, levels, order, damp, moves, strip = spec
        if selected and f"{R}x{n}" not in selected and f"{R}x{n}s{seed}" not in selected:
            continue
        t0 = time.time()
        L = synth.make_layer(R, n, seed)
        t1 = time.time()
        r = run_reference(L, levels, order, damp, moves, strip_mean=strip)
        t2 = time.time()
        rec = dict(
            R=R, n=n, seed=seed, levels=levels, order=order, damp=damp, moves=moves, strip_mean=strip,
            sha_W=sha(L["W"]), sha_H=sha(L["H"]), sha_mean=sha(L["mean"]), sha_scale=sha(L["scale"]),
            sha_idx=sha(r["idx"]), err_f32_hex=np.float32(r["err"]).tobytes().hex(), err=float(r["err"]),
            idx_histogram=np.bincount(r["idx"].ravel(), minlength=levels).tolist(),
            gen_seconds=round(t1 - t0, 2), reference_seconds=round(t2 - t1, 2),
        )
        print(rec, flush=True)
        out.append(rec)
    return out


CODEBOOK_FIT_CASES = [  # (name, count, seed, dtype, size, lagrange_mult)
    ("f32_4", 20000, 31, "float32", 4, 0.0),
    ("f32_8", 20000, 31, "float32", 8, 0.0),
    ("f32_16", 20000, 31, "float32", 16, 0.0),
    ("f32_8_entropy", 20000, 31, "float32", 8, 0.05),
    ("f32_16_entropy", 20000, 31, "float32", 16, 0.3),
    ("f64_8", 5000, 32, "float64", 8, 0.0),
    ("f32_200", 50000, 33, "float32", 200, 0.0),
]


def codebook_fit():
    """Known answers of the reference's codebook TRAINING (sleekit/codebook.py:190-367) on synth.make_samples data."""
    from sleekit.codebook import lloyd_max

    out = {"cases": np.array(json.dumps(CODEBOOK_FIT_CASES))}
    for name, count, seed, dtype, size, lam in CODEBOOK_FIT_CASES:
        data = synth.make_samples(count, seed, np.dtype(dtype).type)
        start = Codebook.equiprobable(data, size)
        out[f"{name}/start_values"], out[f"{name}/start_limits"] = start.values, start.thresholds
        for k in (1, 2, 3):
            cb = lloyd_max(data, size, lam, max_iter=k)
            out[f"{name}/round{k}_values"], out[f"{name}/round{k}_limits"] = cb.values, cb.thresholds
        cb = lloyd_max(data, size, lam)
        out[f"{name}/final_values"], out[f"{name}/final_limits"] = cb.values, cb.thresholds
        out[f"{name}/final_shares"] = cb.probabilities(data)
        out[f"{name}/final_entropy"] = np.float64(cb.entropy(data))
        out[f"{name}/final_mse"] = np.array(cb.mse(data))
    # drawn initialisation and subsample: NumPy's global generator, seeded
    data = synth.make_samples(20000, 34, np.float32)
    np.random.seed(11)
    cb = Codebook.random(data, 8)
    out["random/values"], out["random/limits"] = cb.values, cb.thresholds
    np.random.seed(12)
    cb = lloyd_max(data, 8, random_init=True, sample_count=500)
    out["random_fit/values"], out["random_fit/limits"] = cb.values, cb.thresholds
    # empty bins: centroids' three fall-backs and remove_unused
    cb = Codebook([-50.0, -40.0, -0.5, 0.0, 0.25, 0.5, 30.0, 40.0, 50.0])
    out["empty/centroids"] = cb.centroids(data)
    cb.remove_unused(data)
    out["empty/kept_values"], out["empty/kept_limits"] = cb.values, cb.thresholds
    nf4 = Codebook.nf4()
    out["nf4/shares"] = nf4.probabilities(data / 4)
    out["nf4/mse"] = np.array(nf4.mse(data / 4))
    out["nf4/entropy"] = np.float64(nf4.entropy(data / 4))
    out["nf4/centroids"] = nf4.centroids(data / 4)
    return out


EXPERIMENT_LAYERS = (("a", 96, 172, 2003, ()), ("b", 64, 96, 2050, (5, 40)))  # name, rows, columns, seed, dead columns
EXPERIMENT_RUNS = {  # script -> command-line arguments (small search grids: the fixtures pin call sequences, not runtimes)
    "local_search": ["--codebook-size", "8", "--grid-size", "20"],
    "correction": ["--codebook-size", "8", "--grid-size", "20"],
    "ordering": ["--codebook-size", "8", "--grid-size", "20", "--correct-bias"],
    "dampening": ["--codebook-size", "4", "--grid-size", "20"],
    "bits": ["--grid-size", "16"],
    "scaling": ["--codebook-size", "4", "--grid-size", "12", "--run-max", "--run-diag", "--run-diag1", "--run-diag3", "--run-diag10",
                "--run-hessian", "--run-obq-aware"],
    "compare": ["--codebook-size", "8", "--grid-size", "12"],
}


def write_experiment_data(root):
    """The layer-statistics dumps the experiments walk (weight / hessian / mean .npy per directory, compare.py:37-53),
    from the build's own generator.  tests/experiment_replays.py writes the same bytes on the GPU box."""
    for name, R, n, seed, dead in EXPERIMENT_LAYERS:
        L = synth.make_layer(R, n, seed, dead=dead) if dead else synth.make_layer(R, n, seed)
        d = os.path.join(root, name)
        os.makedirs(d, exist_ok=True)
        np.save(os.path.join(d, "weight.npy"), L["W"])
        np.save(os.path.join(d, "hessian.npy"), L["H"])
        np.save(os.path.join(d, "mean.npy"), L["mean"])


def experiments():
    """Run the reference's OWN experiment scripts (experiments/*.py, unmodified, in a subprocess with the reference on the
    path) on two synthetic layer dumps and keep the TSV they print: header + one row of layer errors per layer."""
    import subprocess
    import tempfile

    out = dict(layers=[list(x[:4]) + [list(x[4])] for x in EXPERIMENT_LAYERS], runs={})
    with tempfile.TemporaryDirectory() as tmp:
        write_experiment_data(tmp)
        for script, argv in EXPERIMENT_RUNS.items():
            env = dict(os.environ, PYTHONPATH="/root/reference")
            res = subprocess.run([sys.executable, f"/root/reference/experiments/{script}.py", tmp] + argv, env=env, capture_output=True,
                                 text=True, check=True)
            lines = [ln for ln in res.stdout.splitlines() if "\t" in ln]
            out["runs"][script] = dict(argv=argv, header=lines[0].split("\t"), rows=[ln.split("\t") for ln in lines[1:]])
            print(script, lines)
    return out


def inverse_diag_orders():
    """obq.py:70-75: the orders that need diag(Hd^-1), and whole layers quantized in them."""
    P = {}
    for R_, n_, seed_ in ((64, 96, 2001), (96, 172, 2003)):
        L = synth.make_layer(R_, n_, seed_)
        Hd = L["H"] + 0.01 * L["H"].diagonal().mean() * np.eye(n_)
        for mode in ("inv_diag", "combined_diag"):
            P[f"{mode}/r{R_}_n{n_}_s{seed_}/order"] = ref_obq.compute_hessian_order(L["W"], Hd, UniformCodebook(8, -1, 1), mode).astype(np.int64)
            P[f"{mode}/r{R_}_n{n_}_s{seed_}/out"] = ref_scaling.quantize_with_scaling(L["W"], L["scale"], UniformCodebook(8, -1, 1), H=L["H"],
                                                                                     act_order=mode, damp=0.01)
    return P


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--large", action="store_true", help="also (re)generate large_cases.json (minutes)")
    ap.add_argument("--only", nargs="*", help="large shapes to run, e.g. 768x768 (or 4096x4096s1011 for one seed)")
    ap.add_argument("--skip-small", action="store_true", help="leave small_cases.npz and pieces.npz as they are")
    ap.add_argument("--ls-traces", action="store_true", help="(re)generate ls_traces.npz (a minute: runs the large LS cases too)")
    ap.add_argument("--codebook-fit", action="store_true", help="(re)generate codebook_fit.npz (seconds)")
    ap.add_argument("--experiments", action="store_true", help="(re)generate experiments.json: the reference's experiment scripts' TSV (a minute)")
    ap.add_argument("--orders", action="store_true", help="(re)generate orders.npz: inv_diag / combined_diag (seconds)")
    args = ap.parse_args()

    env = dict(numpy=np.__version__, python=sys.version.split()[0], cpu_count=os.cpu_count())
    try:
        env["blas"] = np.show_config(mode="dicts")["Build Dependencies"]["blas"]["openblas configuration"]
    except Exception:
        env["blas"] = "unknown"

    if not args.skip_small:
        np.savez_compressed(os.path.join(HERE, "small_cases.npz"), **small_cases())
        np.savez_compressed(os.path.join(HERE, "pieces.npz"), **pieces())
        with open(os.path.join(HERE, "environment.json"), "w") as f:
            json.dump(env, f, indent=1)
    if args.experiments:
        with open(os.path.join(HERE, "experiments.json"), "w") as f:
            json.dump(dict(environment=env, **experiments()), f, indent=1)
    if args.orders:
        np.savez_compressed(os.path.join(HERE, "orders.npz"), **inverse_diag_orders())
    if args.codebook_fit:
        np.savez_compressed(os.path.join(HERE, "codebook_fit.npz"), **codebook_fit())
    if args.ls_traces:
        np.savez_compressed(os.path.join(HERE, "ls_traces.npz"), **ls_traces())
    if args.large:
        path = os.path.join(HERE, "large_cases.json")
        recs = large_cases(args.only)
        if args.only and os.path.exists(path):
            old = {(r["R"], r["n"], r["seed"]): r for r in json.load(open(path))["cases"]}
            old.update({(r["R"], r["n"], r["seed"]): r for r in recs})
            recs = list(old.values())
        with open(path, "w") as f:
            json.dump(dict(environment=env, cases=recs), f, indent=1)


if __name__ == "__main__":
    main()
