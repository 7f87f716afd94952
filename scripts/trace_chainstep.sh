#!/bin/bash
# Device-side timeline of one gmx_chainstep step (scripts/bench_chainstep.py under rocprofv3), averaged over the
# steps that do not open a byte, and over those that do.   scripts/trace_chainstep.sh [streams = 64] [out]
# (1 000 steps: rocprofv3 7.2 dies with a segmentation fault once a process has replayed about 8 192 graph kernel nodes --
# 1 600 steps of this graph, whatever the stream count; 1 041 steps pass, and so does everything without the profiler.)
cd "$(dirname "$0")/.."
S=${1:-64}; OUT=${2:-gpurun_out/chainstep_S$S.txt}
mkdir -p $(dirname $OUT)
export TMPDIR=/tmp
W=$(mktemp -d)
{
python3 scripts/bench_chainstep.py --streams $S --steps 4000
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $W/prof -o run -- python3 scripts/bench_chainstep.py --streams $S --steps 1000 > $W/j.json 2> $W/err
python3 - $W/prof <<'PY'
import csv, glob, sys
root = sys.argv[1]
ev = []
for f in glob.glob(root + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"].split("<")[0].split("(")[0].split(" ")[-1][-44:]))
for f in glob.glob(root + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C " + r.get("Direction", "?")))
ev.sort()
steps, cur = [], []
for e in ev:  # a step = a run of device operations at most 12 us apart (the host's turn between steps is longer)
    if cur and e[0] - max(x[1] for x in cur) > 12000:
        steps.append(cur)
        cur = []
    cur.append(e)
if cur:
    steps.append(cur)
steps = steps[len(steps) // 4:]
by_len = {}
for s in steps:
    by_len.setdefault(len(s), []).append(s)
for n, group in sorted(by_len.items(), key=lambda kv: -len(kv[1]))[:2]:
    print(f"{len(group)} steps of {n} device operations; microseconds from the step's first operation:")
    for k in range(n):
        st = sum(s[k][0] - s[0][0] for s in group) / len(group) / 1e3
        du = sum(s[k][1] - s[k][0] for s in group) / len(group) / 1e3
        print(f"  {group[0][k][2]:50s} starts {st:7.2f}  lasts {du:6.2f}")
    print(f"  device span {sum(s[-1][1] - s[0][0] for s in group) / len(group) / 1e3:.2f} us")
gaps = [b[0][0] - a[-1][1] for a, b in zip(steps, steps[1:])]
gaps.sort()
print(f"between a step's last operation and the next step's first: median {gaps[len(gaps) // 2] / 1e3:.1f} us (graph launch, stream synchronisation, the driver's own turn)")
PY
} 2>&1 | tee $OUT
rm -rf $W
