#!/usr/bin/env python3
"""Average every counter of the rocprofv3 counter_collection CSVs under a directory, per kernel
whose name contains <substr>:  python scripts/pmc_avg.py <dir> <substr>"""
import csv, glob, os, sys, collections
d, sub = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    v = acc[k]
    print(f"{k:40s} {sum(v)/len(v):18.1f}  (n={len(v)})")
