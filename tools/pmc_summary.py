#!/usr/bin/env python
"""Summarises a rocprofv3 --pmc counter_collection.csv per kernel: mean counter value per dispatch (top dispatches by count)."""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
f = glob.glob(d + "/*counter_collection.csv")[0]
acc = defaultdict(lambda: defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:40]
    if pat and pat not in k:
        continue
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    n = max(len(v) for v in cs.values())
    print(f"{k}  dispatches={n}")
    for c, v in sorted(cs.items()):
        v = sorted(v)
        print(f"    {c:28s} mean {sum(v)/len(v):14.1f}  max {v[-1]:14.1f}")
