#!/bin/bash
# PMC pass (GPU box): MFMA busy share and LDS conflicts of the layer-error GEMM.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/pmc; rm -rf gpurun_out/pmc/err_a gpurun_out/pmc/err_b
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU \
  --output-format csv -d gpurun_out/pmc/err_a -- python3 tools/micro_error.py > gpurun_out/pmc/err_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LEVEL_WAVES SQ_INST_LEVEL_LDS \
  --output-format csv -d gpurun_out/pmc/err_b -- python3 tools/micro_error.py > gpurun_out/pmc/err_b.log 2>&1
tail -4 gpurun_out/pmc/err_a.log
python3 - <<'PY'
import csv, glob, collections
for sub in ("err_a", "err_b"):
    cnt = collections.defaultdict(lambda: collections.defaultdict(float))
    for path in glob.glob(f"gpurun_out/pmc/{sub}/*/*counter_collection.csv"):
        for row in csv.DictReader(open(path)):
            k = row["Kernel_Name"].split("(")[0][-24:]
            cnt[k][row["Counter_Name"]] += float(row["Counter_Value"])
    dur = collections.defaultdict(float); calls = collections.Counter()
    for path in glob.glob(f"gpurun_out/pmc/{sub}/*/*kernel_trace.csv"):
        for row in csv.DictReader(open(path)):
            k = row["Kernel_Name"].split("(")[0][-24:]
            dur[k] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"]); calls[k] += 1
    for k, c in sorted(cnt.items(), key=lambda kv: -dur[kv[0]])[:2]:
        extra = ""
        if "GRBM_GUI_ACTIVE" in c and c["GRBM_GUI_ACTIVE"]:
            extra = f" mfma_busy_share={c['SQ_VALU_MFMA_BUSY_CYCLES'] / (c['GRBM_GUI_ACTIVE'] / 8 * 1024):.3f} clock={c['GRBM_GUI_ACTIVE'] / 8 / dur[k]:.2f}GHz"
        print(f"{sub} {k:<26s} calls={calls[k]:4d} time={dur[k]/1e6:8.3f} ms " + " ".join(f"{name}={v:.4g}" for name, v in sorted(c.items())) + extra)
PY
