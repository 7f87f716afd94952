#!/bin/bash
# Kernel AND memory-copy timeline of the run-ahead chain around one round: scripts/trace_copies.sh [bytes] [chunk] [files]
cd "$(dirname "$0")/.."
W=$(mktemp -d); mkdir $W/f
cat DESIGN.md SURVEY.md INTEGRATION.md README.md DESIGN.md SURVEY.md INTEGRATION.md README.md > $W/corpus
for i in $(seq 0 $((${3:-16}-1))); do tail -c +$((i*1531+1)) $W/corpus | head -c ${1:-30000} > $W/f/$i; done
export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $W/prof -o run -- dropin/_build/gmix_chain_many -T ${2:-2048} $W/out $W/f/* > $W/j.json 2> $W/err
python3 - $W/prof <<'PY'
import csv, glob, sys
root = sys.argv[1]
ev = []
for f in glob.glob(root + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"][:40] + f" q{r.get('Queue_Id','?')}"))
for f in glob.glob(root + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), f"C {r.get('Direction','?')} {r.get('Bytes', r.get('Size','?'))} B"))
ev.sort()
t0 = ev[0][0]
mid = ev[len(ev) * 2 // 3][0]
for s, e, n in ev:
    if mid <= s <= mid + 14e6:
        print(f"{(s - t0) / 1e6:10.3f} {(e - t0) / 1e6:10.3f}  {n}")
PY
rm -rf $W
