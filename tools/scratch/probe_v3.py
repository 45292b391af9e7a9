import os, sys, time
import torch
n_probe = 12
dev = torch.device("cuda", 0)
streams = [torch.cuda.Stream(dev) for _ in range(n_probe)]
default = torch.cuda.default_stream(dev)
everyone = streams + [default]
a = torch.randn(4096, 4096, device=dev)
out = torch.empty_like(a)
tick = torch.zeros(len(everyone), 8, device=dev)
for j, st in enumerate(everyone):
    with torch.cuda.stream(st):
        tick[j].add_(1.0)
with torch.cuda.stream(streams[0]):
    torch.mm(a, a, out=out)
torch.cuda.synchronize()
t0 = time.perf_counter()
group_of, groups = {}, []
for i, si in enumerate(everyone):
    if i in group_of:
        continue
    members = [i]; group_of[i] = len(groups)
    done = torch.cuda.Event()
    with torch.cuda.stream(si):
        for _ in range(4):
            torch.mm(a, a, out=out)
        done.record(si)
    marks = {}
    for j, sj in enumerate(everyone):
        if j in group_of: continue
        e = torch.cuda.Event()
        with torch.cuda.stream(sj):
            tick[j].add_(1.0)
            e.record(sj)
        marks[j] = e
    early = set()
    while not done.query():
        for j, e in marks.items():
            if j not in early and e.query():
                early.add(j)
    torch.cuda.synchronize()
    for j in marks:
        if j not in early:
            members.append(j); group_of[j] = len(groups)
    groups.append(members)
print(f"{1e3*(time.perf_counter()-t0):.1f} ms", groups)
