#!/usr/bin/env python
"""Prints the kernel timeline of the last frame from a rocprofv3 --kernel-trace csv directory."""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# last frame = from the last k_get_rays to the end
idx = max(i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_get_rays"))
t0 = int(rows[idx]["Start_Timestamp"])
prev_end = t0
tot = {}
for r in rows[idx:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("void ", "")[:26]
    tot[name] = tot.get(name, 0) + (e - s)
    if (e - s) > 3000 or "-v" in sys.argv:
        print(f"{name:26s} q{r['Queue_Id']} start {(s - t0) / 1e3:8.1f}  dur {(e - s) / 1e3:7.1f}  gap {(s - prev_end) / 1e3:6.1f}")
    prev_end = max(prev_end, e)
print("frame span us:", (prev_end - t0) / 1e3)
for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[:14]:
    print(f"   {k:26s} {v / 1e3:8.1f} us")
