"""The main loops of the reference's experiment scripts, as call sequences over a namespace of the reference's names.

Each `replay_*` walks a directory of layer-statistics dumps (weight / hessian / mean .npy, one directory per layer) and
makes the calls of the script it is named after, in the script's order, with the script's arguments -- what
`experiments/<name>.py` does between its `for root in it:` and its `it.write(...)`:

    local_search.py:62-86   correction.py:59-81   ordering.py:64-90   dampening.py:69-88
    bits.py:78-114          scaling.py:120-137    compare.py:50-133

`ns` holds the names those scripts get from `from sleekit.codebook import *` / `.obq` / `.scaling` (np included: they
use it without importing it).  tests/golden/experiments.json holds what the REFERENCE's scripts printed for the same
dumps (tests/golden/make_golden.py --experiments ran them unmodified); the tests run these replays
  * over the CPU oracle (tests/test_oracle_golden.py, no GPU), and
  * over `sleekit` -- the drop-in package of dropin/ -- on the GPU (tests/test_gpu_parity.py),
and compare the layer errors column by column.
"""

import os

from sleekit_amd import synth


def write_dumps(root, layers):
    """The same bytes make_golden.py's write_experiment_data fed the reference's scripts."""
    import numpy as np

    for name, R, n, seed, dead in layers:
        L = synth.make_layer(R, n, seed, dead=tuple(dead)) if dead else synth.make_layer(R, n, seed)
        d = os.path.join(root, name)
        os.makedirs(d, exist_ok=True)
        np.save(os.path.join(d, "weight.npy"), L["W"])
        np.save(os.path.join(d, "hessian.npy"), L["H"])
        np.save(os.path.join(d, "mean.npy"), L["mean"])


def _roots(data_dir):
    return sorted(root for root, _, files in sorted(os.walk(data_dir)) if {"weight.npy", "hessian.npy", "mean.npy"} <= set(files))


def _load(ns, root):
    np = ns["np"]
    return (np.load(os.path.join(root, k + ".npy")).astype(np.float32) for k in ("weight", "hessian", "mean"))


def _options(argv):
    """--codebook-size N --grid-size G --damp D --scaling S --correct-bias --run-x ... -> dict (the scripts' argparse defaults)."""
    opt = dict(codebook_size=4, grid_size=100, damp=0.01, scaling="mse", correct_bias=False, min_factor=0.05, max_factor=1.0, run=[])
    it = iter(argv)
    for a in it:
        if a == "--correct-bias":
            opt["correct_bias"] = True
        elif a.startswith("--run-"):
            opt["run"].append(a[6:])
        elif a in ("--codebook-size", "--grid-size"):
            opt[a[2:].replace("-", "_")] = int(next(it))
        elif a in ("--damp", "--min-factor", "--max-factor"):
            opt[a[2:].replace("-", "_")] = float(next(it))
        elif a == "--scaling":
            opt["scaling"] = next(it)
        else:
            raise ValueError(a)
    return opt


def _scale(ns, o, weight, cb, hessian, mode=None):
    return ns["compute_scaling"](weight, cb, H=hessian, mode=mode or o["scaling"], grid_size=o["grid_size"], min_factor=o["min_factor"],
                                 max_factor=o["max_factor"])


def replay_local_search(ns, data_dir, argv):
    o = _options(argv)
    cb = ns["UniformCodebook"](o["codebook_size"], -1, 1)
    rows = []
    for root in _roots(data_dir):
        weight, hessian, mean = _load(ns, root)
        ns["remove_dead_values"](hessian, weight)
        if o["correct_bias"]:
            hessian = ns["remove_input_bias"](hessian, mean)
        sc = _scale(ns, o, weight, cb, hessian)
        errs = []
        for moves in (0, 10, 100):
            kw = dict(nb_ls_moves=moves) if moves else {}
            q = ns["quantize_with_scaling"](weight, sc, cb, H=hessian, damp=o["damp"], **kw)
            errs.append(ns["quantization_error"](weight, q, H=hessian))
        rows.append(errs)
    return rows


def replay_correction(ns, data_dir, argv):
    o = _options(argv)
    cb = ns["UniformCodebook"](o["codebook_size"], -1, 1)
    rows = []
    for root in _roots(data_dir):
        weight, hessian, mean = _load(ns, root)
        ns["remove_dead_values"](hessian, weight)
        corrected = ns["remove_input_bias"](hessian, mean)
        sc = _scale(ns, o, weight, cb, hessian)
        plain = ns["quantize_with_scaling"](weight, sc, cb, H=hessian, damp=o["damp"])
        with_bias = ns["quantize_with_scaling"](weight, sc, cb, H=corrected, damp=o["damp"])
        rows.append([ns["quantization_error"](weight, plain, H=hessian), ns["quantization_error"](weight, plain, H=corrected),
                     ns["quantization_error"](weight, with_bias, H=corrected)])
    return rows


def replay_ordering(ns, data_dir, argv):
    o = _options(argv)
    cb = ns["UniformCodebook"](o["codebook_size"], -1, 1)
    rows = []
    for root in _roots(data_dir):
        weight, hessian, mean = _load(ns, root)
        ns["remove_dead_values"](hessian, weight)
        if o["correct_bias"]:
            hessian = ns["remove_input_bias"](hessian, mean)
        sc = _scale(ns, o, weight, cb, hessian)
        errs = []
        for order in ("diag", "err", "sqerr"):
            q = ns["quantize_with_scaling"](weight, sc, cb, H=hessian, act_order=order, damp=o["damp"])
            errs.append(ns["quantization_error"](weight, q, H=hessian))
        rows.append(errs)
    return rows


def replay_dampening(ns, data_dir, argv):
    o = _options(argv)
    cb = ns["UniformCodebook"](o["codebook_size"], -1, 1)
    rows = []
    for root in _roots(data_dir):
        weight, hessian, mean = _load(ns, root)
        ns["remove_dead_values"](hessian, weight)
        if o["correct_bias"]:
            hessian = ns["remove_input_bias"](hessian, mean)
        sc = _scale(ns, o, weight, cb, hessian)
        errs = []
        for damp in (0.001, 0.003, 0.01, 0.03, 0.1, 0.3, 1.0):
            q = ns["quantize_with_scaling"](weight, sc, cb, H=hessian, damp=damp)
            errs.append(ns["quantization_error"](weight, q, H=hessian))
        rows.append(errs)
    return rows


def replay_bits(ns, data_dir, argv, sizes):
    """`sizes`: the codebook sizes of the script's table, read off the fixture's header count (2 ... 32)."""
    o = _options(argv)
    rows = []
    for root in _roots(data_dir):
        weight, standard, mean = _load(ns, root)
        ns["remove_dead_values"](standard, weight)
        corrected = ns["remove_input_bias"](standard, mean)
        errs = []
        for sz in sizes:
            cb = ns["UniformCodebook"](sz, -1, 1)
            sc = _scale(ns, o, weight, cb, standard, mode="mse")
            q = ns["quantize_with_scaling"](weight, sc, cb, H=standard, act_order="diag", damp=0.01)
            errs.append(ns["quantization_error"](weight, q, H=standard))
        for sz in sizes:
            cb = ns["UniformCodebook"](sz, -1, 1)
            sc = _scale(ns, o, weight, cb, corrected, mode="diag")
            q = ns["quantize_with_scaling"](weight, sc, cb, H=corrected, act_order="sqerr", damp=0.03)
            errs.append(ns["quantization_error"](weight, q, H=corrected))
        rows.append(errs)
    return rows


def replay_scaling(ns, data_dir, argv):
    o = _options(argv)
    cb = ns["UniformCodebook"](o["codebook_size"], -1, 1)
    modes = ["mse"] + [{"obq-aware": "obq"}.get(m, m) for m in ("max", "diag", "diag1", "diag3", "diag10", "hessian", "obq-aware") if m in o["run"]]
    rows = []
    for root in _roots(data_dir):
        weight, hessian, mean = _load(ns, root)
        ns["remove_dead_values"](hessian, weight)
        if o["correct_bias"]:
            hessian = ns["remove_input_bias"](hessian, mean)
        errs = []
        for mode in modes:
            sc = _scale(ns, o, weight, cb, hessian, mode=mode)
            q = ns["quantize_with_scaling"](weight, sc, cb, H=hessian, damp=o["damp"])
            errs.append(ns["quantization_error"](weight, q, H=hessian))
        rows.append(errs)
    return rows


def replay_compare(ns, data_dir, argv):
    o = _options(argv)
    cb = ns["UniformCodebook"](o["codebook_size"], -1, 1)
    grid_kw = dict(grid_size=o["grid_size"], min_factor=o["min_factor"], max_factor=o["max_factor"])
    rows = []
    for root in _roots(data_dir):
        weight, standard, mean = _load(ns, root)
        ns["remove_dead_values"](standard, weight)
        corrected = ns["remove_input_bias"](standard, mean)
        errs = []
        sc = ns["compute_min_mse_scaling"](weight, cb, **grid_kw)
        q = ns["quantize_with_scaling"](weight, sc, cb, H=standard, act_order="diag", damp=0.01)
        errs.append(ns["quantization_error"](weight, q, H=standard))
        q = ns["quantize_with_scaling"](weight, sc, cb, H=corrected, act_order="diag", damp=0.01)
        errs.append(ns["quantization_error"](weight, q, H=corrected))
        sc = ns["compute_min_mse_scaling"](weight, cb, H=standard.diagonal(), **grid_kw)
        q = ns["quantize_with_scaling"](weight, sc, cb, H=standard, damp=0.01)
        errs.append(ns["quantization_error"](weight, q, H=standard))
        sc = ns["compute_min_mse_scaling"](weight, cb, H=corrected.diagonal(), **grid_kw)
        q = ns["quantize_with_scaling"](weight, sc, cb, H=corrected, act_order="sqerr", damp=0.03)
        errs.append(ns["quantization_error"](weight, q, H=corrected))
        sc = ns["compute_obq_scaling"](weight, cb, 0, H=corrected, act_order="sqerr", damp=0.03, **grid_kw)
        q = ns["quantize_with_scaling"](weight, sc, cb, H=corrected, act_order="sqerr", damp=0.03, nb_ls_moves=100)
        errs.append(ns["quantization_error"](weight, q, H=corrected))
        rows.append(errs)
    return rows


BITS_SIZES = {10: (2, 3, 4, 5, 7, 8, 9, 15, 16, 32)}  # codebook sizes behind the 2 x 10 columns of bits.py's table

# columns whose value hangs on a choice ranked by a BLAS product in the reference (full-Hessian / OBQ-aware scale search,
# local-search gains): a near-tie may fall the other way and move the error by a hair
LOOSE = {"local_search": {1, 2}, "scaling": {6, 7}, "compare": {4}}


def run(name, ns, data_dir, argv, n_columns):
    if name == "bits":
        return replay_bits(ns, data_dir, argv, BITS_SIZES[n_columns // 2])
    return globals()["replay_" + name](ns, data_dir, argv)


def check_against_fixture(fixture, ns, tmp_dir, only=None, loose_rtol=2e-3):
    """Replay every recorded run over `ns`; assert the errors equal the reference scripts' printed ones (1e-5 relative;
    the LOOSE columns `loose_rtol`)."""
    import numpy as np

    write_dumps(tmp_dir, fixture["layers"])
    for name, rec in fixture["runs"].items():
        if only and name not in only:
            continue
        lead = 1 if name in ("bits", "scaling", "compare") else 2  # columns before the errors: name [, scaling mode]
        want = [[float(x) for x in row[lead:]] for row in rec["rows"]]
        got = run(name, ns, tmp_dir, rec["argv"], len(want[0]))
        assert len(got) == len(want), name
        for g_row, w_row in zip(got, want):
            assert len(g_row) == len(w_row), name
            for c, (g, w) in enumerate(zip(g_row, w_row)):
                rtol = loose_rtol if c in LOOSE.get(name, ()) else 1e-5
                assert abs(float(g) - w) <= rtol * abs(w), (name, c, float(g), w)
