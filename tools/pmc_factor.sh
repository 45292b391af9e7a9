#!/bin/bash
# PMC pass (GPU box): MFMA busy / stall / LDS-conflict counters for the factorisation kernels at n = 4096.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/pmc
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F64 \
  --output-format csv -d gpurun_out/pmc/factor -- python3 tools/micro_factor.py 4096 > gpurun_out/pmc/factor_stdout.log 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/pmc/factor/*/*counter_collection.csv")
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for path in f:
    for row in csv.DictReader(open(path)):
        k = row["Kernel_Name"].split("(")[0][:40]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); 
        if row["Counter_Name"] == "SQ_WAVE_CYCLES": n[k] += 1
for k, c in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", 0)):
    wc = c.get("SQ_WAVE_CYCLES", 1) or 1
    print(f"{k:<42s} n={n[k]:4d} busy={c.get('SQ_BUSY_CYCLES',0):.3e} wave_cyc={wc:.3e} mfma_busy={c.get('SQ_VALU_MFMA_BUSY_CYCLES',0):.3e} "
          f"wait_any/wave={c.get('SQ_WAIT_ANY',0)/wc:.2f} wait_inst/wave={c.get('SQ_WAIT_INST_ANY',0)/wc:.2f} active/wave={c.get('SQ_ACTIVE_INST_ANY',0)/wc:.2f} "
          f"lds_conf={c.get('SQ_LDS_BANK_CONFLICT',0):.3e} mfma_f64_mops={c.get('SQ_INSTS_VALU_MFMA_MOPS_F64',0):.3e}")
PY
