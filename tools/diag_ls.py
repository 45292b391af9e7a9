#!/usr/bin/env python3
"""GPU box: local-search parity, row by row, against the reference's recorded moves (tests/golden/ls_traces.npz).

For every large fixture case with moves, under each way of producing the initial G = (W - Q) H:
which rows end with other indices than the reference, the first move at which the device's sequence leaves the
reference's, how close the reference's decision was there (move_record's ratio), and whether the device took the
reference's runner-up.  Usage: python tools/diag_ls.py [--small]
"""
import hashlib
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import parse_case  # noqa: E402
from sleekit_amd import _lib, codebook, engine, synth  # noqa: E402

T = np.load(os.path.join(ROOT, "tests", "golden", "ls_traces.npz"))


def row_hashes(idx):
    return np.array([int.from_bytes(hashlib.sha256(np.ascontiguousarray(r).tobytes()).digest()[:8], "little") for r in idx], dtype=np.uint64)


def run(name, settings):
    c = parse_case(name)
    L = synth.make_layer(c["R"], c["n"], c["seed"])
    cb = codebook.UniformCodebook(c["levels"], -1, 1)
    W, H, sc = (torch.from_numpy(L[k]).cuda() for k in ("W", "H", "scale"))
    for label, opts in settings:
        for k, v in opts.items():
            _lib.set_option(k, v)
        res = engine.quantize_layer(W, H, cb, sc, c["order"], c["damp"], c["moves"], want_ls_trace=True)
        for k in opts:
            _lib.set_option(k, 0)
        idx, trace = res.idx.cpu().numpy(), res.ls_trace.cpu().numpy()
        bad = np.flatnonzero(row_hashes(idx) != T[name + "/row_hash"])
        near = {int(r): i for i, r in enumerate(T[name + "/rows"])}
        print(f"{name} [{label}]: {len(bad)} rows differ", flush=True)
        # all near-tie rows: does the device follow the reference's moves?
        follows = 0
        for r, i in near.items():
            ref = T[name + "/choice"][i]
            follows += int(np.array_equal(ref, trace[r]))
        print(f"   {follows} of {len(near)} recorded near-tie rows follow the reference's moves exactly")
        for r in bad:
            if int(r) not in near:
                print(f"   row {r}: NOT a recorded near-tie row  device moves {trace[r].tolist()}")
                continue
            i = near[int(r)]
            ref, run_, ratio = T[name + "/choice"][i], T[name + "/runner"][i], T[name + "/ratio"][i]
            m = int(np.flatnonzero(ref != trace[r])[0])
            print(f"   row {r}: first departure at move {m}: reference {ref[m]}, runner-up {run_[m]}, device {trace[r][m]}, ratio {ratio[m]:.3g}"
                  f"  (took the runner-up: {trace[r][m] == run_[m]})")


if __name__ == "__main__":
    names = [str(x) for x in T["names"]]
    if "--small" not in sys.argv:
        names = [x for x in names if parse_case(x)["R"] >= 1024]
    settings = [("bf16x3 G", {}), ("float32-MFMA G", {"no_bf16_error": 1})]
    for name in names:
        run(name, settings)
