"""Does the host's enqueue rate depend on which cores the process runs on? (GPU box)"""
import glob
import os
import subprocess
import sys
import time


def child(cpus):
    if cpus:
        os.sched_setaffinity(0, cpus)
    import torch
    x = torch.zeros(1024, device="cuda")
    s = [torch.cuda.Stream() for _ in range(3)]
    for _ in range(2000):
        x.add_(1)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        for i in range(3000):
            with torch.cuda.stream(s[i % 3]):
                x.add_(1)
        dt = time.perf_counter() - t0
        torch.cuda.synchronize()
        best = min(best, dt)
    print(f"cpus {sorted(cpus)[:4] if cpus else 'default'}... on cpu {open('/proc/self/stat').read().rsplit(')', 1)[1].split()[36]}, gpu numa {gpu_numa()}: {best / 3000 * 1e6:.2f} us per launch", flush=True)


def gpu_numa():
    import torch
    p = torch.cuda.get_device_properties(0)
    bdf = f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
    try:
        return bdf + " node " + open(f"/sys/bus/pci/devices/{bdf}/numa_node").read().strip()
    except OSError as e:
        return bdf + " " + str(e)


def parse(lst):
    out = set()
    for part in lst.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        out.update(range(int(a), int(b or a) + 1))
    return out


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(parse(sys.argv[1]) if sys.argv[1] != "default" else None)
        sys.exit(0)
    print("affinity", len(os.sched_getaffinity(0)), sorted(os.sched_getaffinity(0))[:8])
    for d in glob.glob("/sys/class/drm/card*/device"):
        try:
            print(d, open(d + "/numa_node").read().strip(), open(d + "/local_cpulist").read().strip(), os.path.basename(os.path.realpath(d)))
        except OSError as e:
            print(d, e)
    for nd in sorted(glob.glob("/sys/devices/system/node/node*")):
        print(nd, open(nd + "/cpulist").read().strip())
    print(open("/proc/self/status").read().split("Cpus_allowed_list:")[1].split("\n")[0])
    for quota in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpuset.cpus.effective"):
        try:
            print(quota, open(quota).read().strip())
        except OSError as e:
            print(quota, e)
    nodes = [open(nd + "/cpulist").read().strip() for nd in sorted(glob.glob("/sys/devices/system/node/node*"))]
    for spec in ["default", "default", "default"] + nodes:
        subprocess.run([sys.executable, __file__, spec])
