#!/usr/bin/env python3
"""How much the kernels of a run overlap: from a rocprofv3 --kernel-trace CSV (Start/End timestamps per dispatch), the share of
the busy interval with 0, 1, 2, ... kernels in flight, per-queue busy time, and the kernels with the most time in flight.

    cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -- python3 bench.py --config cfg2 --steps 5 --warmup 3 --no-cpu-baseline --no-extras --no-profile
    python3 tools/trace_overlap.py gpurun_out/trace [skip_fraction]
"""
import collections, csv, glob, sys

root = sys.argv[1]
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5  # look at the last part of the run (the timed steps)
rows = []
for path in glob.glob(f"{root}/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:48], r.get("Queue_Id", "?")))
rows = [r for r in rows if "slk::" in r[2] or "slk" in r[2]]  # the library's kernels only (set-up runs torch's)
rows.sort()
t0, t1 = rows[0][0], max(r[1] for r in rows)
cut = t0 + skip * (t1 - t0)
rows = [r for r in rows if r[0] >= cut]
t0, t1 = rows[0][0], max(r[1] for r in rows)
events = []
for s, e, _, _ in rows:
    events.append((s, 1))
    events.append((e, -1))
events.sort()
hist = collections.Counter()
depth, last = 0, t0
for t, d in events:
    hist[depth] += t - last
    depth += d
    last = t
span = t1 - t0
print(f"{len(rows)} dispatches over {span / 1e6:.2f} ms")
for k in sorted(hist):
    print(f"  {k:2d} kernels in flight: {100 * hist[k] / span:5.1f} %")
by_q = collections.defaultdict(int)
by_k = collections.defaultdict(int)
for s, e, name, q in rows:
    by_q[q] += e - s
    by_k[name] += e - s
print("busy share per hardware queue:", {q: round(v / span, 2) for q, v in sorted(by_q.items())})
cnt_q = collections.Counter(q for _, _, _, q in rows)
print("dispatches per hardware queue:", dict(sorted(cnt_q.items())))
print("kernel time in flight / span:")
for name, v in sorted(by_k.items(), key=lambda kv: -kv[1])[:10]:
    print(f"  {name:<42s} {v / span:5.2f}")
