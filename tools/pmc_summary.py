#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counters per kernel: python tools/pmc_summary.py <dir> [name-substring ...]"""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]
subs = sys.argv[2:]
acc = defaultdict(lambda: defaultdict(float))
calls = defaultdict(int)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if subs and not any(s in k for s in subs):
            continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if (r["Dispatch_Id"], k) not in seen:
            seen.add((r["Dispatch_Id"], k))
            calls[k] += 1
for k in sorted(acc):
    print(k[:100], "calls", calls[k])
    for c, v in sorted(acc[k].items()):
        print(f"   {c:32s} {v / calls[k]:18.1f} per call")
