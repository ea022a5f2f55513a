#!/usr/bin/env python
"""Start-to-start gaps along the simulator's chain of launches, alone and beside the render lanes (needs the timing build of pn_sim.hip:
    PN_VARIANT_UNITS=pn_sim.hip python tools/build_variant.py simstamps -DPN_SIM_STAMPS=1
    PN_LIB_PATH=pienerf_amd/lib/variants/simstamps.so python tools/sim_stamps.py [--lanes 3]
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pienerf_amd import scene  # noqa: E402
from pienerf_amd._lib import LIB_PATH  # noqa: E402
from pienerf_amd.harness import SimRenderHarness  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--lanes", type=int, default=3)
ap.add_argument("--frames", type=int, default=300)
args = ap.parse_args()
raw = C.CDLL(LIB_PATH)
CAP = 65536
buf = (C.c_uint64 * (1 + CAP))()
NAMES = {1: "k_elastic / k_cells_elastic_gather", 2: "k_rhs_gather_chunk", 3: "k_matvec3", 4: "k_update_F"}


def report(tag):
    assert raw.pn_sim_stamps_read(buf, 1) == 0
    n = int(buf[0])
    a = np.frombuffer(buf, dtype=np.uint64, count=1 + CAP)[1:1 + min(n, CAP)].copy()
    ids, t = (a >> np.uint64(56)).astype(np.int64), (a & np.uint64((1 << 56) - 1)).astype(np.int64)
    o = np.argsort(t, kind="stable")
    ids, t = ids[o], t[o]
    if hasattr(raw, "pn_sim_phase_read"):   # phase clocks of k_cells_elastic_gather, all workgroups
        pb = (C.c_uint64 * 24)()
        assert raw.pn_sim_phase_read(pb, 1) == 0
        if pb[8]:
            names = ["loads + F", "SVD + P", "contributions -> LDS", "LDS sums, store, ack", "arrival atomic", "completed kernels' sums"]
            for p_, nme in enumerate(names):
                print(f"   {nme:26s} mean {pb[p_] / pb[8] * 0.01:5.2f} us   max {pb[10 + p_] * 0.01:5.2f} us")
            print(f"   workgroups {pb[8]}, longest start-to-end of one workgroup {pb[9] * 0.01:.2f} us")
    sim = ids < 4                       # the substep's chain: elastic -> gather -> matvec (update_F runs on the render lanes)
    ids_s, t_s = ids[sim], t[sim]
    gaps = np.diff(t_s) / 100.0         # us: from the start of a launch to the start of the next one of the chain = that launch's latency
    print(f"{tag}: {len(t_s)} substep launches stamped")
    for k in (1, 2, 3):
        g = gaps[ids_s[:-1] == k]
        g = g[g < 500]                  # (gaps across pauses of the host loop)
        if len(g):
            print(f"   after {NAMES[k]:20s} start-to-next-start: median {np.median(g):6.1f} us  mean {g.mean():6.1f}  p90 {np.percentile(g, 90):6.1f}  n {len(g)}")
    el = t_s[ids_s == 1]
    if len(el) > 20:
        per_iter = np.diff(el) / 100.0
        per_iter = per_iter[per_iter < 500]
        print(f"   one local/global iteration (elastic to elastic): median {np.median(per_iter):.1f} us -> a substep of 10: {10 * np.median(per_iter) + 10:.0f} us")


h = SimRenderHarness(scene.default_opt(), device="cuda:0")
with torch.no_grad():
    for _ in range(50):
        h.sim.stepforward()
    torch.cuda.synchronize()
    raw.pn_sim_stamps_read(None, 1)
    if hasattr(raw, 'pn_sim_phase_read'):
        raw.pn_sim_phase_read(None, 1)
    for _ in range(100):
        h.sim.stepforward()
    report("substep alone")
    h.capture_pipelined(lanes=args.lanes, n_trips=None)
    for _ in range(60):
        h.step_pipelined()
    torch.cuda.synchronize()
    raw.pn_sim_stamps_read(None, 1)
    if hasattr(raw, 'pn_sim_phase_read'):
        raw.pn_sim_phase_read(None, 1)
    for _ in range(args.frames):
        h.step_pipelined()
    h.drain_pipeline()
    report(f"substep beside {args.lanes} render lanes")
