#!/usr/bin/env python
"""Steady-state analysis of a pipelined run from a rocprofv3 --kernel-trace csv: which queue runs what, how long the
simulator's kernels and the gaps between them get while renders are in flight, and the period of substeps / frames.

    python tools/pipe_trace.py <dir> [fraction of the trace to analyse, from the end: default 0.4]
"""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.4
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
t_end = int(rows[-1]["End_Timestamp"])
t_beg = int(rows[0]["Start_Timestamp"])
cut = t_end - int((t_end - t_beg) * frac)
rows = [r for r in rows if int(r["Start_Timestamp"]) >= cut]


def short(n):
    return n.replace("void ", "").split("(")[0][:28]


perq = defaultdict(lambda: defaultdict(lambda: [0, 0]))
for r in rows:
    e = perq[r["Queue_Id"]][short(r["Kernel_Name"])]
    e[0] += 1
    e[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
span = (t_end - cut) / 1e3
print(f"analysed span {span:.0f} us, {len(rows)} kernels")
for q, ks in perq.items():
    busy = sum(v[1] for v in ks.values()) / 1e3
    top = sorted(ks.items(), key=lambda kv: -kv[1][1])[:4]
    print(f"queue {q}: busy {busy:9.0f} us ({100 * busy / span:5.1f}%)  " + ", ".join(f"{k} x{v[0]} {v[1] / 1e3:.0f}us" for k, v in top))

SIM = ("k_step_begin", "k_matvec3", "k_elastic", "k_rhs_gather", "k_step_end")
sim = [r for r in rows if short(r["Kernel_Name"]).startswith(SIM)]
if sim:
    dur = defaultdict(list)
    gaps = []
    for a, b in zip(sim, sim[1:]):
        dur[short(a["Kernel_Name"])].append(int(a["End_Timestamp"]) - int(a["Start_Timestamp"]))
        if not short(b["Kernel_Name"]).startswith("k_step_begin"):
            gaps.append(int(b["Start_Timestamp"]) - int(a["End_Timestamp"]))
    print("simulator kernels under load: " + ", ".join(f"{k} {sum(v) / len(v) / 1e3:.1f}us" for k, v in dur.items()))
    gaps.sort()
    print(f"gap between consecutive substep kernels: mean {sum(gaps) / len(gaps) / 1e3:.1f} us, median {gaps[len(gaps) // 2] / 1e3:.1f}, p90 {gaps[int(len(gaps) * .9)] / 1e3:.1f}")
    begins = [int(r["Start_Timestamp"]) for r in sim if short(r["Kernel_Name"]).startswith("k_step_begin")]
    ends = [int(r["End_Timestamp"]) for r in sim if short(r["Kernel_Name"]).startswith("k_step_end")]
    if len(begins) > 2:
        per = [(b - a) / 1e3 for a, b in zip(begins, begins[1:])]
        print(f"substep period: mean {sum(per) / len(per):.0f} us (min {min(per):.0f}, max {max(per):.0f}), n={len(per)}")
    if ends and begins:
        lens = [(e - b) / 1e3 for b, e in zip(begins, [x for x in ends if x > begins[0]])]
        if lens:
            print(f"substep length begin->end: mean {sum(lens) / len(lens):.0f} us")
fr = [int(r["Start_Timestamp"]) for r in rows if short(r["Kernel_Name"]).startswith("k_get_rays")]
if len(fr) > 2:
    per = [(b - a) / 1e3 for a, b in zip(fr, fr[1:])]
    print(f"frame period (k_get_rays to k_get_rays): mean {sum(per) / len(per):.0f} us, n={len(per)}")
