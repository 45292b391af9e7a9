#!/usr/bin/env python3
"""Print a slice of a rocprofv3 kernel trace: start / duration / queue / grid / kernel, library kernels only (tools/trace_overlap.py)."""
import csv, glob, sys
root, frac, count = sys.argv[1], float(sys.argv[2]), int(sys.argv[3])
rows = []
for path in glob.glob(f"{root}/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        if "slk" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("slk::", "")[:22], r["Queue_Id"],
                         int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1) * (int(r.get("Grid_Size_Y", 1)) // max(int(r.get("Workgroup_Size_Y", 1)), 1)) * int(r.get("Grid_Size_Z", 1)),
                         int(r.get("LDS_Block_Size", 0) or 0)))
rows.sort()
i0 = int(frac * len(rows))
t0 = rows[i0][0]
for s, e, name, q, wgs, lds in rows[i0:i0 + count]:
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f}  q{q}  {wgs:5d} wg  lds {lds // 1024:3d}K  {name}")
