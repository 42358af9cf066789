#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: per-counter mean over the dispatches of one kernel."""
import csv
import glob
import sys
from collections import defaultdict

root = sys.argv[1]
kern = sys.argv[2] if len(sys.argv) > 2 else "lqr_backward"
acc = defaultdict(list)
for f in sorted(glob.glob(root + "/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        if kern in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, v in acc.items():
    print(f"{k:32s} n={len(v):3d} mean={sum(v)/len(v):.6g}")
