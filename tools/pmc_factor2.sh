#!/bin/bash
# PMC pass (GPU box): MFMA busy share, LDS conflicts and occupancy of the factorisation's tile kernels.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/pmc; rm -rf gpurun_out/pmc/fa gpurun_out/pmc/fb
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU \
  --output-format csv -d gpurun_out/pmc/fa -- python3 tools/micro_factor.py > gpurun_out/pmc/fa.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM \
  --output-format csv -d gpurun_out/pmc/fb -- python3 tools/micro_factor.py > gpurun_out/pmc/fb.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for sub in ("fa", "fb"):
    cnt = collections.defaultdict(lambda: collections.defaultdict(float))
    for path in glob.glob(f"gpurun_out/pmc/{sub}/*/*counter_collection.csv"):
        for row in csv.DictReader(open(path)):
            k = row["Kernel_Name"].split("(")[0][-22:]
            cnt[k][row["Counter_Name"]] += float(row["Counter_Value"])
    dur = collections.defaultdict(float); calls = collections.Counter()
    for path in glob.glob(f"gpurun_out/pmc/{sub}/*/*kernel_trace.csv"):
        for row in csv.DictReader(open(path)):
            k = row["Kernel_Name"].split("(")[0][-22:]
            dur[k] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"]); calls[k] += 1
    for k, c in sorted(cnt.items(), key=lambda kv: -dur[kv[0]])[:5]:
        extra = ""
        if c.get("GRBM_GUI_ACTIVE"):
            cyc = c["GRBM_GUI_ACTIVE"] / 8
            extra = f" mfma_busy={c['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024):.3f} waves_per_simd={c['SQ_WAVE_CYCLES'] * 4 / (cyc * 1024):.2f} wait_share={c['SQ_WAIT_INST_ANY'] / max(c['SQ_WAVE_CYCLES'], 1):.2f}"
        if c.get("SQ_LDS_IDX_ACTIVE"):
            extra = f" lds_conflict_share={c['SQ_LDS_BANK_CONFLICT'] / c['SQ_LDS_IDX_ACTIVE']:.2f} wait_lds={c['SQ_WAIT_INST_LDS']:.3g} vmem_level={c['SQ_INST_LEVEL_VMEM']:.3g} vmem_rd={c['SQ_INSTS_VMEM_RD']:.3g}"
        print(f"{sub} {k:<24s} calls={calls[k]:5d} time={dur[k]/1e6:8.3f} ms{extra}")
PY
