#!/usr/bin/env python
"""Runs a few sim+render steps of the 800x800 chair config (profiling target for rocprofv3).

    python tools/run_frames.py [--frames 5] [--no-sim] [--W 800]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pienerf_amd import scene  # noqa: E402
from pienerf_amd.harness import SimRenderHarness  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=5)
ap.add_argument("--presim", type=int, default=20, help="untimed simulator steps first, so the rendered state is deformed")
ap.add_argument("--no-sim", action="store_true")
ap.add_argument("--W", type=int, default=800)
ap.add_argument("--graph", action="store_true")
ap.add_argument("--no-counters", action="store_true", help="skip the extra frame rendered with the march work counters on")
ap.add_argument("--config", choices=("chair", "stress", "trex"), default="chair", help="bench.py's workloads (stress: with its 4096-ray batches)")
ap.add_argument("--form", choices=("blocking", "pipeline", "fold"), default="blocking",
                help="pipeline: the kernels of the pipelined bench (first trip one lane per ray, fused launch on half the CUs), launched one frame at a time; "
                     "fold: the launch set the three-lane bench really times on the chair — the same with the first trip's network / composite / compaction "
                     "folded into the fused launch (k_trips_fused<.., 2>), where the frame allows it")
args = ap.parse_args()
if args.config == "chair":
    opt = scene.default_opt(W=args.W, H=args.W)
    h = SimRenderHarness(opt, device="cuda:0")
else:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    opt, cloud, ckpt, pose, force, _ = bench.make_config(args.config)
    if args.config == "stress":
        opt["ray_batch"] = int(opt.get("max_ray_batch", 4096))
    h = SimRenderHarness(opt, cloud=cloud, ckpt=ckpt, device="cuda:0")
    h.pose = pose
    if force is not None:
        h.sim.update_force(h.sim.n_IP // 2, force)
if args.form in ("pipeline", "fold") and not h.opt.get("ray_batch"):
    h.opt.update(march_throughput=64, fused_grid=max(torch.cuda.get_device_properties(0).multi_processor_count // 2, 1))
if args.form == "fold" and not h.opt.get("ray_batch"):
    h.opt.update(fused_fold=True)
for _ in range(args.presim):
    h.sim.stepforward()
if args.graph:
    h.capture(n_trips=8)
    for _ in range(args.frames):
        h.step_graph()
    torch.cuda.synchronize()
    print(h.model.render_status())
else:
    for _ in range(args.frames):
        h.step(simulate=not args.no_sim, collect_stats=True)
    torch.cuda.synchronize()
    print(h.model.last_stats)
    if args.no_counters:
        sys.exit(0)
    h.model.march_counters(1)
    h.step(simulate=False, collect_stats=True)
    print("counters", h.model.march_counters(1, read=True))
    print("trips (n_alive, n_step, step_base, n_samples, n_tail)", h.model.trip_records())
