#!/bin/bash
# Kernel timeline of the run-ahead chain: where the stages of neighbouring chunks overlap and where they wait.
#   scripts/trace_chain_timeline.sh [bytes per file = 20000] [chunk bits = 2048] [files = 1]
cd "$(dirname "$0")/.."
W=$(mktemp -d); mkdir $W/f
cat DESIGN.md SURVEY.md INTEGRATION.md README.md DESIGN.md SURVEY.md INTEGRATION.md README.md > $W/corpus
for i in $(seq 0 $((${3:-1}-1))); do tail -c +$((i*1531+1)) $W/corpus | head -c ${1:-20000} > $W/f/$i; done
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $W/prof -o run -- dropin/_build/gmix_chain_many -T ${2:-2048} $W/out $W/f/* > $W/j.json 2> $W/err
f=$(find $W/prof -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
for r in rows:
    n = r["Kernel_Name"]
    k = "lstm" if "gmx_lstm_kernel" in n else "ind" if "gmx_indirect_kernel" in n else "mix" if "gmx_stock_kernel" in n else "feed" if "scatter" in n else None
    if k: ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), k))
ev.sort()
t0 = ev[0][0]
by = {}
for s, e, k in ev: by.setdefault(k, []).append((s - t0, e - t0))
for k, v in by.items():
    d = [(e - s) / 1e6 for s, e in v]
    gaps = [(v[i + 1][0] - v[i][1]) / 1e6 for i in range(len(v) - 1)]
    mid = slice(len(v) // 4, 3 * len(v) // 4)
    print(f"{k:5s} n={len(v):4d} dur ms avg {sum(d[mid]) / max(1, len(d[mid])):.3f}  gap-to-next ms avg {sum(gaps[mid]) / max(1, len(gaps[mid])):.3f}  period {(v[mid][-1][0] - v[mid][0][0]) / 1e6 / max(1, len(v[mid]) - 1):.3f}")
ls = [s for s, e in by.get("lstm", [])]
print("lstm start-to-start ms, in order:", " ".join(f"{(b - a) / 1e6:.1f}" for a, b in zip(ls, ls[1:])))
if len(ls) > 24:   # the rows around the longest wait between two LSTM launches (past the start-up)
    k = max(range(12, len(ls) - 2), key=lambda i: ls[i + 1] - ls[i])
    lo, hi = ls[k] - 8e6, ls[k + 1] + 8e6
    print(f"around the longest LSTM start-to-start ({(ls[k + 1] - ls[k]) / 1e6:.1f} ms, round {k}): kind start end (all kernels of the trace)")
    for r in rows:
        st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if lo <= st - t0 <= hi:
            print(f"  {r['Kernel_Name'][:44]:44s} {(st - t0) / 1e6:9.3f} {(en - t0) / 1e6:9.3f}  queue {r.get('Queue_Id', '?')} stream {r.get('Stream_Id', '?')} grid {r.get('Grid_Size_X', r.get('Grid_Size', '?'))}")
print("first rounds (ms since first kernel): kind start end")
for s, e, k in ev[len(ev) // 2: len(ev) // 2 + 16]: print(f"  {k:5s} {(s - t0) / 1e6:9.3f} {(e - t0) / 1e6:9.3f}")
PY
rm -rf $W
