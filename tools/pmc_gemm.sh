#!/bin/bash
# PMC pass (GPU box): is a GEMM kernel MFMA-pipe-bound or clock-bound?  GRBM_GUI_ACTIVE / 8 / duration = clock.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/pmc
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
  --output-format csv -d gpurun_out/pmc/gemm -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile --streams 1,1 > gpurun_out/pmc/gemm_stdout.log 2>&1
python3 - <<'PY'
import csv, glob, collections
cnt = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for path in glob.glob("gpurun_out/pmc/gemm/*/*counter_collection.csv"):
    for row in csv.DictReader(open(path)):
        k = row["Kernel_Name"].split("(")[0][-28:]
        cnt[k][row["Counter_Name"]] += float(row["Counter_Value"])
        if row["Counter_Name"] == "GRBM_GUI_ACTIVE": n[k] += 1
dur = collections.defaultdict(float)
for path in glob.glob("gpurun_out/pmc/gemm/*/*kernel_trace.csv"):
    for row in csv.DictReader(open(path)):
        dur[row["Kernel_Name"].split("(")[0][-28:]] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
for k, c in sorted(cnt.items(), key=lambda kv: -dur[kv[0]]):
    if not dur[k] or n[k] == 0: continue
    clock = c["GRBM_GUI_ACTIVE"] / 8 / dur[k]  # GHz (cycles per ns), counter summed over 8 XCDs
    mfma_util = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8 * 1024) if c["GRBM_GUI_ACTIVE"] else 0
    print(f"{k:<30s} launches={n[k]:4d} time={dur[k]/1e6:8.3f} ms clock={clock:5.2f} GHz mfma_busy/(cycles*1024 SIMD)={mfma_util:5.2f} "
          f"wait_inst/wave_cyc={c['SQ_WAIT_INST_ANY']/max(c['SQ_WAVE_CYCLES'],1):.2f} active/wave_cyc={c['SQ_ACTIVE_INST_ANY']/max(c['SQ_WAVE_CYCLES'],1):.2f}")
PY
