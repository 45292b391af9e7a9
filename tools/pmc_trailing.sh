#!/bin/bash
# PMC pass (GPU box): what keeps the float64 tile kernels (gptq_trailing, syrk, trtri) off the MFMA peak?
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/pmc
rm -rf gpurun_out/pmc/trail_a gpurun_out/pmc/trail_b
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU \
  --output-format csv -d gpurun_out/pmc/trail_a -- python3 tools/micro_loop.py > gpurun_out/pmc/trail_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_CVT SQ_INST_LEVEL_LDS \
  --output-format csv -d gpurun_out/pmc/trail_b -- python3 tools/micro_loop.py > gpurun_out/pmc/trail_b.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for sub in ("trail_a", "trail_b"):
    cnt = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for path in glob.glob(f"gpurun_out/pmc/{sub}/*/*counter_collection.csv"):
        for row in csv.DictReader(open(path)):
            k = row["Kernel_Name"].split("(")[0][-24:]
            cnt[k][row["Counter_Name"]] += float(row["Counter_Value"])
    dur = collections.defaultdict(float); calls = collections.Counter()
    for path in glob.glob(f"gpurun_out/pmc/{sub}/*/*kernel_trace.csv"):
        for row in csv.DictReader(open(path)):
            k = row["Kernel_Name"].split("(")[0][-24:]
            dur[k] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"]); calls[k] += 1
    for k, c in sorted(cnt.items(), key=lambda kv: -dur[kv[0]])[:4]:
        print(f"{sub} {k:<26s} calls={calls[k]:4d} time={dur[k]/1e6:8.3f} ms " + " ".join(f"{name}={v:.4g}" for name, v in sorted(c.items())))
PY
