#!/usr/bin/env python3
"""Diagnostic (GPU box): where do product / oracle / golden indices differ on the large cases?

For each case prints: does the oracle run HERE reproduce the golden hash made in the build
container (different CPU => different OpenBLAS kernels), how many indices / rows differ between
the HIP path and the oracle, and the layer errors.  Output: one JSON line per case.
"""
import hashlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import grid, obq_ref, scaling_ref
from sleekit_amd import codebook, obq, scaling, synth

sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
cases = json.load(open(os.path.join(ROOT, "tests/golden/large_cases.json")))["cases"]
only = sys.argv[1:]
for c in cases:
    tag = f"{c['R']}x{c['n']}s{c['seed']}"
    if only and tag not in only and f"{c['R']}x{c['n']}" not in only:
        continue
    L = synth.make_layer(c["R"], c["n"], c["seed"])
    g = grid.UniformGrid(c["levels"], -1, 1); cb = codebook.UniformCodebook(c["levels"], -1, 1)
    H = obq_ref.strip_input_mean(L["H"], L["mean"]) if c["strip_mean"] else L["H"]
    t0 = time.time()
    o_out = scaling_ref.quantize_scaled(L["W"], L["scale"], g, H, c["order"], c["damp"], c["moves"])
    t1 = time.time()
    o_idx = g.index(scaling_ref.divide_rows(o_out, L["scale"], 0))
    Hp = obq.remove_input_bias(L["H"], L["mean"]) if c["strip_mean"] else L["H"]
    p_out = scaling.quantize_with_scaling(L["W"], L["scale"], cb, Hp, c["order"], c["damp"], c["moves"])
    p_idx = cb.quantize_index(scaling.apply_scaling(p_out, L["scale"], 0))
    diff = p_idx != o_idx
    rows = np.nonzero(diff.any(axis=1))[0]
    o_err = float(obq_ref.mean_error(L["W"], o_out, H)); p_err = float(obq.quantization_error(L["W"], p_out, Hp))
    p_err_cpu = float(obq_ref.mean_error(L["W"], p_out, H))
    print(json.dumps(dict(case=tag, levels=c["levels"], moves=c["moves"], strip=c["strip_mean"],
        H_strip_equal=bool(np.array_equal(H, Hp)),
        oracle_here_matches_golden=sha(o_idx) == c["sha_idx"], product_matches_golden=sha(p_idx) == c["sha_idx"],
        idx_diff=int(diff.sum()), rows_diff=int(len(rows)), rows=rows[:8].tolist(),
        diff_per_row=[int(diff[r].sum()) for r in rows[:8]],
        err_golden=c["err"], err_oracle_here=o_err, err_product=p_err, err_product_evaluated_on_cpu=p_err_cpu,
        rel_err_vs_golden=abs(p_err - c["err"]) / c["err"], oracle_seconds=round(t1 - t0, 2))), flush=True)
