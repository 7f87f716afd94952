#!/bin/bash
# Where one lock-step decode step goes (gmx_chainstep: one hipGraph per coded bit for all files): kernel and copy
# timeline of `gmix_chain_many -d` from rocprofv3, averaged over the steps of the run's middle.
#   scripts/trace_decode_step.sh [files = 64] [bytes per file = 1500] [out = gpurun_out/decode_step_S$S.txt]
cd "$(dirname "$0")/.."
S=${1:-64}; N=${2:-1500}; OUT=${3:-gpurun_out/decode_step_S$S.txt}
mkdir -p $(dirname $OUT)
W=$(mktemp -d); mkdir $W/f
for i in $(seq 0 $((S-1))); do python3 scripts/corpus.py $W/f/$(printf %04d $i) $N $((i*1531)) > /dev/null; done
dropin/_build/gmix_chain_many $W/c $W/f/* > $W/c.json 2> $W/c.err || { echo "compress failed"; tail -3 $W/c.err; exit 1; }
export TMPDIR=/tmp
# (keep S x bytes small: rocprofv3 7.2 segfaults past about 8 192 replayed graph kernel nodes, see trace_chainstep.sh)
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $W/prof -o run -- dropin/_build/gmix_chain_many -d $W/back $(for i in $(seq 0 $((S-1))); do echo $W/c/$i.gmix; done) > $W/j.json 2> $W/err
python3 - $W/prof $W/j.json <<'PY' | tee $OUT
import csv, glob, json, sys
root, jf = sys.argv[1:3]
j = json.load(open(jf))
print(f"{j['files']} files, {j['launches']} steps, {j['wall_seconds'] * 1e6 / j['launches']:.1f} us per step under the profiler, {j['bits_per_second']:.3g} bits/s")
ev = []
for f in glob.glob(root + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"].split("(")[0][-44:]))
for f in glob.glob(root + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C " + r.get("Direction", "?")))
ev.sort()
# a step = the events between two D2H copies (the step's last node)
steps, cur = [], []
for e in ev:
    cur.append(e)
    if e[2].startswith("C") and "DEVICE_TO_HOST" in e[2].upper().replace(" ", "_"):
        steps.append(cur)
        cur = []
mid = steps[len(steps) // 3: 2 * len(steps) // 3]
mid = [s for s in mid if len(s) == max(set(len(x) for x in mid), key=[len(x) for x in mid].count)]  # the common shape (no byte opening)
if not mid:
    sys.exit("no steps found")
n = len(mid[0])
print(f"{len(mid)} steps of the common shape ({n} device operations each), microseconds from the step's first operation:")
for k in range(n):
    st = sum(s[k][0] - s[0][0] for s in mid) / len(mid) / 1e3
    du = sum(s[k][1] - s[k][0] for s in mid) / len(mid) / 1e3
    print(f"  {mid[0][k][2]:50s} starts {st:7.2f}  lasts {du:6.2f}")
span = sum(s[-1][1] - s[0][0] for s in mid) / len(mid) / 1e3
gap = sum(b[0][0] - a[-1][1] for a, b in zip(mid, mid[1:])) / max(1, len(mid) - 1) / 1e3
print(f"device span of a step {span:.2f} us; from a step's last operation to the next step's first {gap:.2f} us (host: fibres, barrier, graph launch)")
PY
rm -rf $W
