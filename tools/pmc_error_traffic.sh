#!/bin/bash
# PMC pass (GPU box): bytes the layer-error GEMM fetches past the L2 (FETCH_SIZE, in KB of 64 B per 128-B request on gfx950:
# doubled below, as tools/profile_round.sh does) -- square tiles, 256 x 128 and 256 x 256 tiles; WANT_G=1 for the full product.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/pmc
for T in 0 1 2; do
  rm -rf gpurun_out/pmc/errt_$T
  SLK_TALL_ERROR=$T rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc/errt_$T -- python3 tools/micro_error.py > gpurun_out/pmc/errt_$T.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
for T in (0, 1, 2):
    tot, n = collections.defaultdict(float), collections.Counter()
    for path in glob.glob(f"gpurun_out/pmc/errt_{T}/*/*counter_collection.csv"):
        seen = set()
        for row in csv.DictReader(open(path)):
            k = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("slk::", "")
            if "error_tiles" not in k and "split3" not in k: continue
            tot[k] += float(row["Counter_Value"])
            if (row["Dispatch_Id"], k) not in seen:
                seen.add((row["Dispatch_Id"], k)); n[k] += 1
    for k in tot:
        print(f"tall_error={T} {k[:40]:40s} launches {n[k]:3d}  fetched past L2 per launch {2 * tot[k] / n[k] * 1024 / 1e9:7.3f} GB")
PY
