#!/usr/bin/env python
"""Times the headless training loop (pienerf_amd/training.py) on the synthetic teacher: steps/s, rays/s, samples/s, PSNR.

    python tools/time_train.py [--steps 300] [--rays 4096] [--W 128]
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
from pienerf_amd.nerf.network import NeRFNetwork  # noqa: E402
from pienerf_amd.nerf.utils import get_rays  # noqa: E402
from pienerf_amd.training import RayImageSet, Trainer  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=300)
ap.add_argument("--rays", type=int, default=4096)
ap.add_argument("--W", type=int, default=128)
args = ap.parse_args()
dev = "cuda:0"
ck = scene.make_checkpoint(bound=1.0, seed=0, shaped=True)
teacher = NeRFNetwork(encoding="hashgrid", bound=1.0, cuda_ray=True).to(dev).load_checkpoint_dict(ck)
intr = scene.orbit_intrinsics(args.W, args.W, 50.0)
poses = np.stack([scene.orbit_pose(4.0, a, e) for a in range(0, 360, 30) for e in (-20.0, -50.0)]).astype(np.float32)
images = []
with torch.no_grad():
    for p in poses:
        r = get_rays(torch.from_numpy(p[None]).to(dev), intr, args.W, args.W)
        images.append(teacher.run_cuda(r["rays_o"], r["rays_d"], bg_color=1, max_steps=1024, T_thresh=1e-2)["image"].view(args.W, args.W, 3))
images = torch.stack(images)
torch.manual_seed(1)
student = NeRFNetwork(encoding="hashgrid", bound=1.0, cuda_ray=True, density_thresh=10).to(dev)
data = RayImageSet(torch.from_numpy(poses).to(dev), intr, images, generator=torch.Generator().manual_seed(2))
tr = Trainer(student, dict(dt_gamma=0, max_steps=1024, T_thresh=1e-2), lr=1e-2, iters=args.steps + 64, num_rays=args.rays)
tr.train(data, 64)  # warm-up: grid updates, allocator, mean_count
torch.cuda.synchronize()
samples = 0
t0 = time.time()
losses = tr.train(data, args.steps)
torch.cuda.synchronize()
dt = time.time() - t0
psnr, _ = tr.evaluate(data, 0)
print(json.dumps({"train_steps_per_s": round(args.steps / dt, 2), "ms_per_step": round(1e3 * dt / args.steps, 3), "rays_per_s": round(args.steps * args.rays / dt),
                  "mean_count": student.mean_count, "samples_per_s": round(student.mean_count * args.steps / dt), "loss_first": round(float(np.mean(losses[:5])), 5),
                  "loss_last": round(float(np.mean(losses[-20:])), 5), "psnr_view0": round(psnr, 2), "rays": args.rays, "W": args.W}))
