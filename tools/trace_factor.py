#!/usr/bin/env python3
"""One factorisation under `rocprofv3 --kernel-trace`: prints the LAST call's kernels in start order with their offsets (us)
from the call's first kernel, durations and queues -- what runs beside what in the look-ahead.  Usage (GPU box):
    cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && \
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tf -- python3 tools/trace_factor.py run
    python3 tools/trace_factor.py show gpurun_out/tf          (COLS=4096 LOOKAHEAD=1 by default)"""
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if sys.argv[1] == "run":
    import torch

    from sleekit_amd import _lib, engine, synth

    n = int(os.environ.get("COLS", "4096"))
    L = synth.make_layer_device(8, n, 4100, torch.device("cuda"))
    for _ in range(4):
        engine.factorize(L["H"], n, 0.01, _lib.ORDER_DIAG, lookahead=bool(int(os.environ.get("LOOKAHEAD", "1"))))
        torch.cuda.synchronize()
else:
    rows = []
    for path in glob.glob(os.path.join(sys.argv[2], "*", "*kernel_trace.csv")):
        rows += list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    starts = [i for i, r in enumerate(rows) if "k_diag_prepare" in r["Kernel_Name"]]  # the last call: from its first kernel on
    rows = rows[starts[-1]:]
    t0 = int(rows[0]["Start_Timestamp"])
    for r in rows:
        name = r["Kernel_Name"].split("(")[0].replace("slk::", "").replace("void ", "")
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} +{dur:7.1f} us  q{r.get('Queue_Id', '?'):>3s}  {name[:36]:36s} grid {r.get('Grid_Size_X', r.get('Grid_Size', '?'))}")
    print(f"total {(int(rows[-1]['End_Timestamp']) - t0) / 1e3:.1f} us")
