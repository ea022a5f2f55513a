#!/usr/bin/env python
"""Per-lane view of a pipelined run from a rocprofv3 --kernel-trace csv: for every hardware queue, how long a frame's launch chain lasts, how
much of that is kernels and how much is the gaps between them, which kernels stretch most against their solo durations, and how many kernels
are running at any time (the GPU's concurrency profile).

    python tools/lane_timeline.py <dir> [fraction of the trace to analyse, from the end: default 0.4]
"""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.4
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
t_end = max(int(r["End_Timestamp"]) for r in rows)
t_beg = int(rows[0]["Start_Timestamp"])
cut = t_end - int((t_end - t_beg) * frac)
rows = [r for r in rows if int(r["Start_Timestamp"]) >= cut]


def short(n):
    n = n.replace("void ", "")
    return n.split("(")[0][:40]


byq = defaultdict(list)
for r in rows:
    byq[r["Queue_Id"]].append(r)
span = (t_end - cut) / 1e3
print(f"analysed span {span:.0f} us, {len(rows)} kernels, {len(byq)} queues")
for q, rs in sorted(byq.items()):
    frames = [i for i, r in enumerate(rs) if short(r["Kernel_Name"]).startswith("k_get_rays")]
    if len(frames) < 3:
        busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs) / 1e3
        print(f"queue {q}: {len(rs)} kernels, busy {busy:.0f} us ({100 * busy / span:.1f} %), first: {short(rs[0]['Kernel_Name'])}")
        continue
    chains = []
    per_kernel = defaultdict(lambda: [0, 0.0, 0.0])  # count, duration, gap before
    for a, b in zip(frames, frames[1:]):
        fr = rs[a:b]
        # the frame's chain ends with k_frame_finish
        fin = [i for i, r in enumerate(fr) if short(r["Kernel_Name"]).startswith("k_frame_finish")]
        if not fin:
            continue
        fr = fr[: fin[-1] + 1]
        t0, t1 = int(fr[0]["Start_Timestamp"]), int(fr[-1]["End_Timestamp"])
        ker = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in fr)
        chains.append(((t1 - t0) / 1e3, ker / 1e3, len(fr)))
        prev = None
        for i, r in enumerate(fr):
            e = per_kernel[f"{i:03d} {short(r['Kernel_Name'])}"]
            e[0] += 1
            e[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            if prev is not None:
                e[2] += (int(r["Start_Timestamp"]) - prev) / 1e3
            prev = int(r["End_Timestamp"])
    if not chains:
        continue
    n = len(chains)
    print(f"queue {q}: {n} frames, chain {sum(c[0] for c in chains) / n:.0f} us (kernels {sum(c[1] for c in chains) / n:.0f} us, gaps "
          f"{sum(c[0] - c[1] for c in chains) / n:.0f} us, {chains[0][2]} launches)")
    if "-v" in sys.argv:
        for k, e in sorted(per_kernel.items()):
            print(f"      {k:46s} dur {e[1] / e[0]:7.1f}  gap {e[2] / e[0]:6.1f}")
# concurrency profile: time-weighted histogram of kernels in flight
ev = []
for r in rows:
    ev.append((int(r["Start_Timestamp"]), 1))
    ev.append((int(r["End_Timestamp"]), -1))
ev.sort()
hist = defaultdict(int)
cur, last = 0, ev[0][0]
for t, dlt in ev:
    hist[cur] += t - last
    cur += dlt
    last = t
tot = sum(hist.values())
print("kernels in flight (share of time): " + ", ".join(f"{k}: {100 * v / tot:.1f} %" for k, v in sorted(hist.items())))
