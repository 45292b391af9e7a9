#!/usr/bin/env python3
"""One layer through the single-layer API under `rocprofv3 --kernel-trace`: prints the LAST call's kernels in start order with
their offsets (us) from the call's first kernel, durations and queues -- what the latency of a layer alone is made of.
    cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && \
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python3 tools/trace_layer.py run
    python3 tools/trace_layer.py show gpurun_out/tl          (ROWS=4096 COLS=4096 MOVES=0 by default)"""
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if sys.argv[1] == "run":
    import torch

    from sleekit_amd import codebook, engine, synth

    R, n = int(os.environ.get("ROWS", "4096")), int(os.environ.get("COLS", "4096"))
    L = synth.make_layer_device(R, n, 4100, torch.device("cuda"))
    cb = codebook.UniformCodebook(8, -1, 1)
    for _ in range(4):
        res = engine.quantize_layer(L["W"], L["H"], cb, L["scale"], "diag", 0.01, int(os.environ.get("MOVES", "0")))
        err = engine.row_errors(L["W"], res.Q, L["H"]) if os.environ.get("SEPARATE_ERROR") else None
        torch.cuda.synchronize()
else:
    rows = []
    for path in glob.glob(os.path.join(sys.argv[2], "*", "*kernel_trace.csv")):
        rows += list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    starts = [i for i, r in enumerate(rows) if "k_diag_prepare" in r["Kernel_Name"]]  # the last call: from its first kernel on
    first = starts[-1]
    while first > 0 and int(rows[first]["Start_Timestamp"]) - int(rows[first - 1]["End_Timestamp"]) < 200_000:
        first -= 1  # (kernels of the same call before the factorisation's first)
    rows = rows[first:]
    t0 = int(rows[0]["Start_Timestamp"])
    prev_end = t0
    for r in rows:
        name = r["Kernel_Name"].split("(")[0].replace("slk::", "").replace("void ", "")
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print(f"{(s - t0) / 1e3:9.1f} +{(e - s) / 1e3:7.1f} us  gap {max(0, s - prev_end) / 1e3:6.1f}  q{r.get('Queue_Id', '?'):>3s}  {name[:40]:40s} grid {r.get('Grid_Size_X', r.get('Grid_Size', '?'))}")
        prev_end = max(prev_end, e)
    print(f"total {(int(rows[-1]['End_Timestamp']) - t0) / 1e3:.1f} us")
