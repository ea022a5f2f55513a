#!/usr/bin/env python
"""The trex option set (bench.py's configs[2] scene and force) along its trajectory: every `--every`-th of `--frames` sim+render steps is rendered twice from the
same integration-point state — with the skip pre-pass crossing the static background's empty regions on the ray's t-sequence (the default) and visiting them
voxel by voxel (pn_march_set_skip_dda(0)) — and the two frames' images, depths, sample counts and trip counts are compared bit for bit.  The body stretches
and leaves the grid along the way (points filed under wrapped cell indices), which is the part of the trajectory the small tests do not reach.

    python tools/trex_region_trajectory.py [--frames 300] [--every 10]
"""
import argparse
import hashlib
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from pienerf_amd._lib import check, lib  # noqa: E402
from pienerf_amd.harness import SimRenderHarness  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=300)
ap.add_argument("--every", type=int, default=10)
args = ap.parse_args()
opt, cloud, ckpt, pose, force, _ = bench.make_config("trex")
h = SimRenderHarness(opt, cloud=cloud, ckpt=ckpt, device="cuda:0", overlap_sim=False)
h.pose = pose
h.sim.update_force(h.sim.n_IP // 2, force)
compared, differing, errs, rows = 0, 0, 0, []
with torch.no_grad():
    for f in range(args.frames):
        try:
            h.step(simulate=True, collect_stats=True)
        except RuntimeError as e:        # a point whose flat cell index leaves [0, n_grid): error flag 2, loudly (the reference drops it silently)
            errs += 1
            rows.append((f, "error", str(e)[:60]))
            break
        if f % args.every:
            continue
        res = []
        for on in (0, 1):
            check(lib().pn_march_set_skip_dda(on), "set_skip_dda")
            out = h.step(simulate=False, collect_stats=True)
            torch.cuda.synchronize()
            st = h.model.last_stats
            res.append((hashlib.sha1(out["image"].cpu().numpy().tobytes()).hexdigest()[:12], hashlib.sha1(torch.nan_to_num(out["depth"], nan=-1.0).cpu().numpy().tobytes()).hexdigest()[:12],
                        st["samples"], st["trips"], st["err"]))
        check(lib().pn_march_set_skip_dda(-1), "set_skip_dda")
        compared += 1
        differing += res[0] != res[1]
        rows.append((f, res[0][2], res[0] == res[1]))
disp = float((h.model.p_def - h.model.p_ori).abs().max())
print(json.dumps({"frames": args.frames, "compared": compared, "differing": int(differing), "stopped_by_error_flag": errs, "max_ip_displacement_at_end": round(disp, 3),
                  "samples_first_last": [rows[0][1], rows[-1][1]], "rows": rows[:3] + rows[-3:]}))
sys.exit(1 if differing else 0)
