#!/usr/bin/env python
"""Per-dispatch counter table of the kernels matching a pattern, from a rocprofv3 --pmc output directory (counter_collection.csv).

    python tools/pmc_dispatch.py <dir> <pattern> [first] [count]
"""
import csv
import glob
import sys
from collections import OrderedDict, defaultdict

d, pat = sys.argv[1], sys.argv[2]
first = int(sys.argv[3]) if len(sys.argv) > 3 else 0
count = int(sys.argv[4]) if len(sys.argv) > 4 else 1000
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
rows = OrderedDict()
names = []
for r in csv.DictReader(open(f)):
    if pat not in r["Kernel_Name"]:
        continue
    key = int(r["Dispatch_Id"])
    rows.setdefault(key, defaultdict(float))[r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] not in names:
        names.append(r["Counter_Name"])
print("dispatch " + " ".join(n.replace("SQ_", "")[:14].rjust(14) for n in names))
for i, (k, v) in enumerate(rows.items()):
    if first <= i < first + count:
        print(f"{k:8d} " + " ".join(f"{v[n]:14.0f}" for n in names))
