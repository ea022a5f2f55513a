#!/usr/bin/env python
"""BASELINE configs[2] (trex option set, synthetic assets) at full size: eager sim+render step time and frame statistics.

    python tools/time_trex.py [--frames 50]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pienerf_amd import scene  # noqa: E402
from pienerf_amd.harness import SimRenderHarness  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=50)
ap.add_argument("--lanes", type=int, default=3)
ap.add_argument("--background", type=float, default=0.0, help="fraction of 8^3-voxel blocks of the density grid marked occupied outside the object "
                "(a real LLFF scene has a dense static background; the synthetic checkpoint has none)")
args = ap.parse_args()
W, H = 1008, 756
opt = scene.default_opt(bound=2.0, scale=0.33, dt_gamma=1.0 / 128, max_steps=300, T_thresh=5e-2, num_seek_IP=1, max_iter_num=1, cut=True,
                        cut_bounds=[-0.62, 1.0, -0.82, 0.42, -0.52, 0.28], sim_dx=0.05, W=W, H=H, radius=4.5)
cloud = scene.make_chair_points(hgs=opt["hash_grid_size"], bound=opt["bound"])
ckpt = scene.make_checkpoint(bound=2.0, seed=3)
if args.background > 0:
    blobs = np.repeat(np.random.default_rng(5).random(len(ckpt["density_bitfield"]) // 64) < args.background, 64)
    ckpt["density_bitfield"] = ckpt["density_bitfield"] | np.where(blobs, 0xFF, 0).astype(np.uint8)
h = SimRenderHarness(opt, cloud=cloud, ckpt=ckpt, device="cuda:0")
h.pose = scene.orbit_pose(4.5, 25.0, -10.0)
h.sim.update_force(h.sim.n_IP // 2, np.array([250.0, 120.0, -180.0]))
with torch.no_grad():
    for _ in range(5):
        h.step(collect_stats=True)
    torch.cuda.synchronize()
    st = dict(h.model.last_stats)
    t0 = time.time()
    for _ in range(args.frames):
        h.step()
    torch.cuda.synchronize()
    eager = (time.time() - t0) / args.frames
    h.capture_pipelined(lanes=args.lanes, n_trips=8)
    for _ in range(10):
        h.step_pipelined()
    h.drain_pipeline()
    t0 = time.time()
    for _ in range(args.frames * 4):
        h.step_pipelined()
    h.drain_pipeline()
    piped = (time.time() - t0) / (args.frames * 4)
print(json.dumps({"config": "configs[2] trex option set, 1008x756, bound 2, cut, dt_gamma 1/128, max_steps 300, num_seek_IP 1", "rays": W * H, "n_IP": h.sim.n_IP,
                  "eager_ms_per_step": round(eager * 1e3, 3), "pipelined_ms_per_step": round(piped * 1e3, 3), "pipelined_steps_per_s": round(1 / piped, 1),
                  "lanes": args.lanes, "background": args.background, **st}))
