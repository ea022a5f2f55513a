#!/usr/bin/env python
"""HBM-side traffic per kernel from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE cannot share a pass: TCC has 4 slots, they cost 3 + 2) ->
profiles/pmc_traffic.json, stamped with the hash of the kernel sources it was measured on (pienerf_amd.build.source_hash); bench.py reports
`roofline.traffic` from it only when the stamp matches the tree it runs from.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d OUT/fetch -o fetch --output-format csv -- python tools/run_frames.py --frames 3 --no-sim
    rocprofv3 --pmc WRITE_SIZE --kernel-trace -d OUT/write -o write --output-format csv -- python tools/run_frames.py --frames 3 --no-sim
    python tools/pmc_traffic.py OUT/fetch OUT/write 3 [profiles/pmc_traffic.json]

Units and corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE are in KB per dispatch; on gfx950 FETCH_SIZE reports half the bytes
of a wide coalesced read (128-B requests tallied at 64 B) -> doubled here; other access widths (the 4/8-byte gathers of the network kernel) and
WRITE_SIZE are uncalibrated, so the figures are upper-side indications, not exact HBM bytes; Infinity-Cache hits are included.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pienerf_amd.build import source_hash  # noqa: E402


def per_kernel(d, counter):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    tot, cnt = defaultdict(float), defaultdict(set)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()   # template arguments kept: bench.py checks the instance against the launch it times
        tot[k] += float(r["Counter_Value"])
        cnt[k].add(r["Dispatch_Id"])
    return tot, {k: len(v) for k, v in cnt.items()}


def main():
    fetch_dir, write_dir, frames = sys.argv[1], sys.argv[2], int(sys.argv[3])
    out = sys.argv[4] if len(sys.argv) > 4 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_traffic.json")
    fe, nf = per_kernel(fetch_dir, "FETCH_SIZE")
    wr, _ = per_kernel(write_dir, "WRITE_SIZE")
    kernels = {}
    for k in sorted(fe):
        if not k.startswith("k_"):
            continue
        kernels[k] = {"fetch_bytes_per_frame": int(2 * 1024 * fe[k] / frames), "write_bytes_per_frame": int(1024 * wr.get(k, 0.0) / frames),
                      "dispatches_per_frame": nf[k] / frames}
    js = {"lib_hash": source_hash(), "frames": frames,
          "_provenance": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on python tools/run_frames.py --frames N --no-sim; tools/pmc_traffic.py",
          "_units": "bytes per frame summed over the kernel's dispatches (empty trailing trips move ~nothing); FETCH_SIZE KB x 1024 x 2 (gfx950: 128-B requests "
                    "tallied at 64 B, MI355X_MICROARCH.md), WRITE_SIZE KB x 1024; gather widths and writes are uncalibrated; Infinity-Cache hits included",
          "kernels": kernels}
    with open(out, "w") as f:
        json.dump(js, f, indent=1, sort_keys=True)
        f.write("\n")
    print(json.dumps(js["kernels"], indent=1))


if __name__ == "__main__":
    main()
