#!/usr/bin/env python
"""Per-dispatch table of a rocprofv3 --pmc counter_collection.csv for one kernel (rows = dispatches in launch order).

    python tools/pmc_trips.py <dir> <kernel substring> [last N dispatches]
"""
import csv
import glob
import sys
from collections import OrderedDict, defaultdict

d, pat = sys.argv[1], sys.argv[2]
last = int(sys.argv[3]) if len(sys.argv) > 3 else 16
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
rows = OrderedDict()
names = []
for r in csv.DictReader(open(f)):
    if pat not in r["Kernel_Name"]:
        continue
    did = int(r["Dispatch_Id"])
    rows.setdefault(did, defaultdict(float))
    rows[did][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] not in names:
        names.append(r["Counter_Name"])
print("dispatch " + " ".join(f"{n[-18:]:>18s}" for n in names))
for did in sorted(rows)[-last:]:
    print(f"{did:8d} " + " ".join(f"{rows[did][n]:18.0f}" for n in names))
