#!/usr/bin/env python
"""Times the widened rows on cuda:0: point sampling at sub_res 60 / 180, the static 800x800 render
(NeRFRenderer.run_cuda, eval branch), update_extra_state, and the headless front end's frame loop incl. PNG encoding.

    python tools/time_widened.py
"""
import json
import os
import sys
import tempfile
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pienerf_amd import main_render, scene  # noqa: E402
from pienerf_amd.nerf.network import NeRFNetwork  # noqa: E402
from pienerf_amd.nerf.utils import get_rays  # noqa: E402
from pienerf_amd.sampling import AdaptiveUniformSampling  # noqa: E402

dev = "cuda:0"


def timed(fn, n, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.time() - t0) / n


out = {}
ck = scene.make_checkpoint(bound=1.0, seed=0, shaped=True)
net = NeRFNetwork(encoding="hashgrid", bound=1.0, cuda_ray=True).to(dev).load_checkpoint_dict(ck)
for sub_res, coeff in ((60, 0.55), (180, 0.75)):  # README.md:91,115 of the reference
    o = scene.default_opt(sub_res=sub_res, sub_coeff=coeff, density_threshold=0.05)
    s = AdaptiveUniformSampling(o, net)
    g = torch.Generator(device=dev)
    dt = timed(lambda: s.sample(generator=g.manual_seed(0)), 3, warm=1)
    out[f"sampling_sub_res_{sub_res}"] = {"seconds": round(dt, 4), "lattice_sites": sub_res ** 3, "points": int(s.sample(generator=g.manual_seed(0))[0].shape[0])}
pose = torch.from_numpy(scene.orbit_pose(4.0, 40.0, -20.0)[None]).to(dev)
rays = get_rays(pose, scene.orbit_intrinsics(800, 800, 50.0), 800, 800)
dt = timed(lambda: net.run_cuda(rays["rays_o"], rays["rays_d"], bg_color=1, max_steps=1024, T_thresh=1e-2), 20)
out["static_render_800x800"] = {"ms_per_frame": round(dt * 1e3, 3), "frames_per_s": round(1 / dt, 1), **net.last_stats}
net.reset_extra_state()
dt = timed(lambda: (setattr(net, "iter_density", 0), net.update_extra_state()), 5)
out["update_extra_state_full_sweep"] = {"ms": round(dt * 1e3, 3), "cells": 128 ** 3}
dt = timed(lambda: (setattr(net, "iter_density", 16), net.update_extra_state()), 5)
out["update_extra_state_partial"] = {"ms": round(dt * 1e3, 3)}
with tempfile.TemporaryDirectory() as td:
    args = main_render.parser().parse_args(["--out", td, "--frames", "30", "--quiet"])
    t0 = time.time()
    main_render.run(args)
    out["main_render_30_frames_800x800_png"] = {"seconds_incl_init": round(time.time() - t0, 3)}
print(json.dumps(out))
