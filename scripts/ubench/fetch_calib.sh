#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of scattered 16-bit accesses against known sector counts (scripts/ubench/fetch_calib.hip).
#   bash scripts/ubench/fetch_calib.sh [out = gpurun_out/fetch_calib.txt]
cd "$(dirname "$0")/../.."
OUT=${1:-gpurun_out/fetch_calib.txt}
export TMPDIR=/tmp
B=/tmp/fetch_calib
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 scripts/ubench/fetch_calib.hip -o $B || exit 1
mkdir -p $(dirname $OUT)
: > $OUT
for pat in 0 1 2 3 4 5; do
  for pmc in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ TCC_EA0_RDREQ_32B"; do
    d=$(mktemp -d)
    timeout -k 10 120 rocprofv3 --pmc $pmc --output-format csv -d $d -o c -- $B $pat 16 > $d/log 2>&1
    f=$(find $d -name "*counter_collection.csv" | head -1)
    python3 - "$f" "$d/log" "$pmc" >> $OUT <<'PY'
import csv, json, sys
f, log, pmc = sys.argv[1:4]
line = next((l for l in open(log) if l.startswith("{")), "{}")
j = json.loads(line)
vals = {}
if f:
    for r in csv.DictReader(open(f)):
        if "k_" in r["Kernel_Name"]:
            vals[r["Counter_Name"]] = vals.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
n = j.get("accesses", 1)
out = {"pattern": j.get("pattern"), "accesses": n, "kernel_ms": j.get("kernel_ms")}
for k, v in vals.items():
    out[k] = v
    out[k + "_per_access"] = (v * 1024 if k.endswith("_SIZE") else v) / n   # *_SIZE are KiB
print(json.dumps(out))
PY
    rm -rf $d
  done
done
cat $OUT
