#!/bin/bash
# PMC pass (GPU box): clock and MFMA-pipe utilisation of the bare MFMA probes.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/pmc
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
  --output-format csv -d gpurun_out/pmc/probe -- python3 tools/micro_mfma.py > gpurun_out/pmc/probe_stdout.log 2>&1
python3 - <<'PY'
import csv, glob
rows = {}
for path in glob.glob("gpurun_out/pmc/probe/*/*counter_collection.csv"):
    for row in csv.DictReader(open(path)):
        rows.setdefault(row["Dispatch_Id"], {"name": row["Kernel_Name"][:40], "grid": row.get("Grid_Size", "")})[row["Counter_Name"]] = float(row["Counter_Value"])
dur = {}
for path in glob.glob("gpurun_out/pmc/probe/*/*kernel_trace.csv"):
    for row in csv.DictReader(open(path)):
        dur[row["Dispatch_Id"]] = float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
for d, c in sorted(rows.items(), key=lambda kv: int(kv[0])):
    if "probe" not in c["name"] or d not in dur: continue
    clock = c["GRBM_GUI_ACTIVE"] / 8 / dur[d]
    util = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8 * 1024)
    print(f"{c['name']:<42s} grid={c['grid']:>8s} {dur[d]/1e6:7.3f} ms clock={clock:5.2f} GHz mfma_util={util:5.2f}")
PY
