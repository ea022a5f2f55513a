#!/usr/bin/env python
"""Per-kernel SUMS per frame of a rocprofv3 --pmc counter_collection.csv (instruction counts, wave-cycles): where a frame's issue slots go.

    python tools/pmc_instr.py <dir> <frames> [kernel-name filter]
"""
import csv
import glob
import sys
from collections import defaultdict

d, frames = sys.argv[1], int(sys.argv[2])
pat = sys.argv[3] if len(sys.argv) > 3 else ""
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
acc = defaultdict(lambda: defaultdict(float))
disp = defaultdict(set)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0][:60]
    if pat and pat not in k:
        continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    disp[k].add(r["Dispatch_Id"])
names = sorted({c for v in acc.values() for c in v})
tot = defaultdict(float)
print(f"{'kernel':60s} {'launches/frame':>14s} " + " ".join(f"{c[:18]:>18s}" for c in names))
for k in sorted(acc, key=lambda k_: -acc[k_].get("SQ_INSTS_VALU", acc[k_].get(names[0], 0))):
    print(f"{k:60s} {len(disp[k]) / frames:14.1f} " + " ".join(f"{acc[k].get(c, 0) / frames:18.0f}" for c in names))
    for c in names:
        tot[c] += acc[k].get(c, 0) / frames
print(f"{'TOTAL per frame':60s} {'':14s} " + " ".join(f"{tot[c]:18.0f}" for c in names))
