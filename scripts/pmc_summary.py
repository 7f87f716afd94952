#!/usr/bin/env python3
"""Turn the CSVs of scripts/gpu_profile.sh into profiles/rNN_<tag>_* files.

  python scripts/pmc_summary.py <tag> <round> <kernel-name-prefix> <streams> <bits> <alg-bytes-per-bit> "<command>" [ctx_mode]

The library build the passes ran on (gmx_build_info(), printed by bench.py in roofline.build) is read from
the stats pass's JSON line; bench.py marks a quoted figure stale when it no longer matches.

FETCH_SIZE is doubled (gfx950 reports half the bytes of 16-B-per-lane reads, MI355X_MICROARCH.md);
WRITE_SIZE is taken as is; both are in KiB per dispatch."""
import csv
import glob
import json
import os
import shutil
import sys

tag, rnd, kprefix, S, T, apb, cmd = sys.argv[1], int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]), sys.argv[7]
ctx_mode = int(sys.argv[8]) if len(sys.argv) > 8 else 0
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{tag}")
dst = os.path.join(root, "profiles")


def find(sub, suffix):
    f = glob.glob(os.path.join(src, sub, "**", f"*{suffix}"), recursive=True)
    assert f, (sub, suffix)
    return f[0]


def counter_avg(path, name):
    vals = []
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == name and kprefix in r["Kernel_Name"]:
            vals.append(float(r["Counter_Value"]))
    # the bench's timed launches all have the same shape; warm-up launches too
    return sum(vals) / len(vals), len(vals)


stats = find("stats", "kernel_stats.csv")
shutil.copy(stats, os.path.join(dst, f"r{rnd:02d}_{tag}_kernel_stats.csv"))
fetch = find("fetch", "counter_collection.csv")
write = find("write", "counter_collection.csv")
shutil.copy(fetch, os.path.join(dst, f"r{rnd:02d}_{tag}_pmc_fetch_size.csv"))
shutil.copy(write, os.path.join(dst, f"r{rnd:02d}_{tag}_pmc_write_size.csv"))
f_kb, nf = counter_avg(fetch, "FETCH_SIZE")
w_kb, nw = counter_avg(write, "WRITE_SIZE")
kname, kavg = None, None
for r in csv.DictReader(open(stats)):
    if kprefix in r["Name"]:
        kname, kavg = r["Name"], float(r["AverageNs"]) * 1e-6
        break
build = None
try:
    for line in open(os.path.join(src, "stats.log")):
        if line.startswith("{"):
            build = json.loads(line)["roofline"].get("build")   # bench.py, bench_indirect.py and bench_lstm.py all say
except Exception:
    pass
out = {
    "round": rnd, "build": build, "ctx_mode": ctx_mode, "command": cmd, "kernel": kname, "kernel_ms_avg_rocprof": kavg,
    "streams": S, "bits_per_stream": T, "dispatches_counted": [nf, nw],
    "FETCH_SIZE_kb_avg": f_kb, "WRITE_SIZE_kb_avg": w_kb,
    "fetch_bytes_corrected": f_kb * 1024 * 2, "write_bytes": w_kb * 1024,
    "traffic_bytes_per_launch": f_kb * 1024 * 2 + w_kb * 1024,
    "algorithmic_bytes_per_launch": apb * S * T,
    "traffic_over_algorithmic": (f_kb * 1024 * 2 + w_kb * 1024) / (apb * S * T),
    "correction": "FETCH_SIZE doubled (gfx950 reports half the bytes of 16-B-per-lane coalesced reads); WRITE_SIZE taken as is",
}
json.dump(out, open(os.path.join(dst, f"r{rnd:02d}_{tag}_pmc_summary.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
