#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel, mean of each counter per dispatch."""
import csv
import glob
import sys
from collections import defaultdict

root = sys.argv[1]
filt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = defaultdict(lambda: defaultdict(list))
for f in sorted(glob.glob(root + "/**/*counter_collection.csv", recursive=True)):
    per = defaultdict(lambda: defaultdict(float))
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if filt and filt not in k:
            continue
        per[(k, row["Dispatch_Id"])][row["Counter_Name"]] += float(row["Counter_Value"])
    for (k, _), cs in per.items():
        for c, v in cs.items():
            acc[k][c].append(v)
for k, cs in acc.items():
    print(k[:100])
    for c, vs in sorted(cs.items()):
        print(f"   {c:32s} mean {sum(vs)/len(vs):18.1f}   n={len(vs)}")
