#!/bin/bash
# Round profile (GPU box): rocprofv3 kernel stats of the default bench command + HBM traffic counters.
#   usage: bash tools/profile_round.sh r01
# Writes gpurun_out/profiles_<tag>/ ; copy the summaries into profiles/ and commit them.
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/profiles_$TAG
mkdir -p $OUT
# 1. per-kernel time of the default bench command
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py > $OUT/bench_under_rocprof.log 2>&1
cp $OUT/stats/*/*kernel_stats.csv $OUT/${TAG}_kernel_stats.csv
grep '"metric"' $OUT/bench_under_rocprof.log > $OUT/${TAG}_bench_line_under_rocprof.json
# 2. HBM traffic: FETCH_SIZE and WRITE_SIZE need separate passes (TCC slots)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile --streams 1,1 > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile --streams 1,1 > $OUT/write.log 2>&1
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, json, sys, collections
out, tag = sys.argv[1], sys.argv[2]
names = {"k_chol_panel": "chol_panel", "k_gptq_window": "gptq_window", "k_error_tiles_bf16": "error_gemm_bf16", "k_error_tiles(": "error_gemm", "k_split3": "error_split", "k_gptq_trailing": "gptq_trailing",
         "k_syrk_tiles": "chol_syrk_inner", "k_syrk_triangle": "chol_syrk_outer", "k_trtri_level<0>": "trtri_stage0", "k_trtri_level<1>": "trtri_stage1", "k_permute_in": "permute_in",
         "k_permute_out": "permute_out", "k_gather_reversed": "gather_reversed", "k_flip_out": "flip_out", "k_rows_divide": "rows_divide"}
def collect(sub, counter):
    tot, cnt = collections.Counter(), collections.Counter()
    for path in glob.glob(f"{out}/{sub}/*/*counter_collection.csv"):
        for row in csv.DictReader(open(path)):
            if row["Counter_Name"] != counter: continue
            k = next((v for key, v in names.items() if key in row["Kernel_Name"]), None)
            if k is None: continue
            tot[k] += float(row["Counter_Value"]); cnt[k] += 1
    return {k: (tot[k] / cnt[k], cnt[k]) for k in tot}
fetch, write = collect("fetch", "FETCH_SIZE"), collect("write", "WRITE_SIZE")
res = {"note": "per-launch HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE in KB counts 64 B per 128-B request on gfx950 "
               "(guide, HBM section), so it is doubled; WRITE_SIZE is exact for 16-B stores. Single-stream run, 8 layers 4096x4096.",
       "bytes_per_launch": {}, "detail": {}}
for k in sorted(set(fetch) | set(write)):
    f, nf = fetch.get(k, (0.0, 0)); w, nw = write.get(k, (0.0, 0))
    res["bytes_per_launch"][k] = (2 * f + w) * 1024
    res["detail"][k] = {"FETCH_SIZE_KB_avg": f, "WRITE_SIZE_KB_avg": w, "launches": max(nf, nw)}
json.dump(res, open(f"{out}/{tag}_pmc_traffic.json", "w"), indent=1)
print(json.dumps(res["bytes_per_launch"], indent=1))
PY
ls -la $OUT | head -20
