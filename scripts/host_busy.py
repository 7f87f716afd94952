#!/usr/bin/env python3
"""Which CPUs of this host are busy (somebody else's threads included): /proc/stat twice, a second apart.
Prints the CPUs above 30 % and, per NUMA node, how many that is.  A diagnostic for the pinned worker threads of the
lock-step decoder and the many-file compressor on a shared host."""
import glob, os, sys, time


def snap():
    out = {}
    for line in open("/proc/stat"):
        if line.startswith("cpu") and line[3].isdigit():
            f = line.split()
            v = list(map(int, f[1:]))
            out[int(f[0][3:])] = (sum(v), v[3] + v[4])
    return out


a = snap()
time.sleep(float(sys.argv[1]) if len(sys.argv) > 1 else 1.0)
b = snap()
busy = {c: 1.0 - (b[c][1] - a[c][1]) / max(1, b[c][0] - a[c][0]) for c in a}
hot = sorted(c for c, u in busy.items() if u > 0.3)
print("busy CPUs (> 30 %):", " ".join(f"{c}:{busy[c]:.0%}" for c in hot))
for node in sorted(glob.glob("/sys/devices/system/node/node[0-9]*")):
    cpus = []
    for part in open(node + "/cpulist").read().strip().split(","):
        lo, _, hi = part.partition("-")
        cpus += list(range(int(lo), int(hi or lo) + 1))
    print(os.path.basename(node), f"{len(cpus)} CPUs, {sum(1 for c in cpus if c in hot)} busy; list {open(node + '/cpulist').read().strip()}")
for d in sorted(glob.glob("/sys/bus/pci/devices/*")):
    try:
        if open(d + "/vendor").read().strip() != "0x1002":
            continue
        cls = open(d + "/class").read().strip()
        if not (cls.startswith("0x03") or cls.startswith("0x1200")):
            continue
        print("GPU", os.path.basename(d), "class", cls, "numa_node", open(d + "/numa_node").read().strip())
    except OSError:
        pass
