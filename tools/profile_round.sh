#!/bin/bash
# Round profile (GPU box): rocprofv3 kernel stats of the driver's bench command + HBM traffic and MFMA / LDS counters.
#   usage: bash tools/profile_round.sh r02 <commit>
# Writes gpurun_out/profiles_<tag>/ ; copy the summaries into profiles/ and commit them.
# (rocprofv3: the program comes directly after `--`; counters go in passes of their own, with --kernel-trace only.)
TAG=${1:-r02}
COMMIT=${2:-unrecorded}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/profiles_$TAG
rm -rf $OUT
mkdir -p $OUT
# 1. per-kernel time of the command the driver runs: its headline leg (--no-configs: the config legs get a CSV each below,
#    so that a kernel's average is over launches of ONE workload)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-configs > $OUT/bench_under_rocprof.log 2>&1
cp $OUT/stats/*/*kernel_stats.csv $OUT/${TAG}_kernel_stats_default_bench.csv
grep '"metric"' $OUT/bench_under_rocprof.log > $OUT/${TAG}_bench_line_under_rocprof.json
echo "stats pass done"
# 1b. one kernel-stats CSV per BASELINE config (the legs of the default run, as their own commands)
for CFG in cfg2 cfg3 cfg4 cfg5; do
  BLK=""; STEPS=3
  if [ $CFG = cfg5 ]; then BLK="--blocks 16"; STEPS=2; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$CFG -- python3 bench.py --gpus 1 --config $CFG $BLK --steps $STEPS --warmup 1 --no-cpu-baseline --no-extras > $OUT/bench_${CFG}_under_rocprof.log 2>&1
  cp $OUT/stats_$CFG/*/*kernel_stats.csv $OUT/${TAG}_kernel_stats_${CFG}.csv
  grep '"metric"' $OUT/bench_${CFG}_under_rocprof.log > $OUT/${TAG}_bench_line_${CFG}_under_rocprof.json
  rm -rf $OUT/stats_$CFG
  echo "stats pass $CFG done"
done
# 2. counters: one pass each, single-stream run of one step (every kernel alone on the chip)
ARGS="--steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-extras --no-configs --streams 1,1"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 bench.py $ARGS > $OUT/fetch.log 2>&1
echo "fetch pass done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 bench.py $ARGS > $OUT/write.log 2>&1
echo "write pass done"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $OUT/mfma -- python3 bench.py $ARGS > $OUT/mfma.log 2>&1
echo "mfma pass done"
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $OUT/lds -- python3 bench.py $ARGS > $OUT/lds.log 2>&1
echo "lds pass done"
python3 - "$OUT" "$TAG" "$COMMIT" <<'PY'
import csv, glob, json, sys, collections
out, tag, commit = sys.argv[1], sys.argv[2], sys.argv[3]
names = {"k_chol_chain<false>": "chol_chain", "k_chol_chain<true>": "chol_rows_below", "k_chol_rows_below": "chol_rows_below", "k_chol_panel": "chol_panel", "k_panel_below": "chol_panel_below", "k_stack_rows": "stack_rows", "k_gptq_window": "gptq_window", "k_error_tiles_bf16": "error_gemm_bf16", "k_error_tiles(": "error_gemm", "k_split3": "error_split", "k_gptq_trailing": "gptq_trailing",
         "k_syrk_tiles": "chol_syrk_inner", "k_syrk_triangle": "chol_syrk_outer", "k_trtri_level<0>": "trtri_stage0", "k_trtri_level<1>": "trtri_stage1", "k_permute_in": "permute_in",
         "k_permute_out": "permute_out", "k_gather_reversed": "gather_reversed", "k_flip_out": "flip_out", "k_rows_divide": "rows_divide"}
def short(kernel_name):
    return next((v for key, v in names.items() if key in kernel_name), None)
def counters(sub):
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.Counter()
    for path in glob.glob(f"{out}/{sub}/*/*counter_collection.csv"):
        seen = set()
        for row in csv.DictReader(open(path)):
            k = short(row["Kernel_Name"])
            if k is None: continue
            tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
            key = (row.get("Dispatch_Id"), k)
            if key not in seen:
                seen.add(key); launches[k] += 1
    return tot, launches
def durations(sub):
    dur, calls = collections.defaultdict(float), collections.Counter()
    for path in glob.glob(f"{out}/{sub}/*/*kernel_trace.csv"):
        for row in csv.DictReader(open(path)):
            k = short(row["Kernel_Name"])
            if k is None: continue
            dur[k] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"]); calls[k] += 1
    return dur, calls
fetch, nf = counters("fetch"); write, nw = counters("write")
res = {"note": "per-launch HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE in KB counts 64 B per 128-B request on gfx950 "
               "(guide, HBM section), so it is doubled; WRITE_SIZE is exact for 16-B stores. Single-stream run, 8 layers 4096x4096, every layer its own inputs.",
       "made_by": "tools/profile_round.sh", "commit": commit, "bytes_per_launch": {}, "detail": {}}
for k in sorted(set(fetch) | set(write)):
    f = fetch[k].get("FETCH_SIZE", 0.0) / max(nf[k], 1); w = write[k].get("WRITE_SIZE", 0.0) / max(nw[k], 1)
    res["bytes_per_launch"][k] = (2 * f + w) * 1024
    res["detail"][k] = {"FETCH_SIZE_KB_avg": f, "WRITE_SIZE_KB_avg": w, "launches": max(nf[k], nw[k])}
json.dump(res, open(f"{out}/{tag}_pmc_traffic.json", "w"), indent=1)
# MFMA busy / LDS conflicts per kernel
mf, _ = counters("mfma"); md, mc = durations("mfma")
ld, _ = counters("lds")
util = {"note": "rocprofv3 --pmc passes over a single-stream run of one step (8 layers 4096x4096, each kernel alone on the chip). "
                "mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024): busy cycles summed over the 1024 SIMDs against the "
                "kernel's active cycles (GRBM_GUI_ACTIVE is summed over the 8 XCDs); clock_ghz = GRBM_GUI_ACTIVE / 8 / duration; "
                "lds_conflict_share = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE; wait_share = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES.",
        "made_by": "tools/profile_round.sh", "commit": commit, "kernels": {}}
for k in sorted(mf, key=lambda k: -md.get(k, 0.0)):
    c = mf[k]; g = c.get("GRBM_GUI_ACTIVE", 0.0)
    e = {"launches": mc[k], "avg_launch_us": round(md[k] / max(mc[k], 1) / 1e3, 2)}
    if g:
        e["mfma_busy"] = round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (g / 8 * 1024), 4)
        e["clock_ghz"] = round(g / 8 / md[k], 3) if md.get(k) else None
    if c.get("SQ_WAVE_CYCLES"): e["wait_share"] = round(c.get("SQ_WAIT_INST_ANY", 0.0) / c["SQ_WAVE_CYCLES"], 4)
    l = ld.get(k, {})
    if l.get("SQ_LDS_IDX_ACTIVE"): e["lds_conflict_share"] = round(l.get("SQ_LDS_BANK_CONFLICT", 0.0) / l["SQ_LDS_IDX_ACTIVE"], 4)
    util["kernels"][k] = e
json.dump(util, open(f"{out}/{tag}_pmc_mfma.json", "w"), indent=1)
print(json.dumps(util["kernels"], indent=1))
PY
ls -la $OUT | head -20
