"""Headless simulate-and-render front end: what ``main_gui.py`` does with the window closed and ``main_render.py``
does for one frame — load a point cloud and a checkpoint, step the simulator, render every frame, write PNGs (and optionally the
deformed point cloud and the per-frame IP state the reference's ``main_render.py`` reads back from ``./debug``).

    python -m pienerf_amd.main_render --frames 30 --out output_img/chair [--ply model/chair_0.ply] [--ckpt ws/checkpoints/ngp_ep0300.pth]
           [--W 800 --H 800] [--radius 5 --azimuth 0 --elevation 0 --fovy 50] [--force fx fy fz] [--save_ply] [--save_ip_state]

Without --ply / --ckpt the synthetic chair of pienerf_amd.scene is used (there are no assets on the GPU box).
Reference: main_gui.py:20-66 (model + simulator construction), nerf/gui.py:556-645 (test_step: IP info -> substep -> render),
main_render.py:47-104 (frame loop, save_image), simulator/solver.py:109-113 (OutputToPly).
"""
import argparse
import os
import time

import numpy as np
import torch

from . import io, scene
from .harness import SimRenderHarness


def build_harness(args):
    opt = scene.default_opt(W=args.W, H=args.H, radius=args.radius, fovy=args.fovy, sim_dx=args.sim_dx, sim_iters=args.sim_iters,
                            max_iter_num=args.max_iter_num, num_seek_IP=args.num_seek_IP, bound=args.bound, dt_gamma=args.dt_gamma,
                            max_steps=args.max_steps, T_thresh=args.T_thresh)
    cloud = scene.cloud_from_ply(args.ply) if args.ply else None
    h = SimRenderHarness(opt, cloud=cloud, ckpt=None, device=args.device)
    if args.ckpt:
        path = args.ckpt if os.path.isfile(args.ckpt) else io.latest_checkpoint(args.ckpt)
        if path is None:
            raise FileNotFoundError(f"no checkpoint under {args.ckpt}")
        io.load_checkpoint(h.model, path, model_only=True, allow_pickle=args.trust_ckpt)
    return h


def run(args):
    h = build_harness(args)
    pose = scene.orbit_pose(args.radius, args.azimuth, args.elevation)
    os.makedirs(args.out, exist_ok=True)
    if args.force is not None:
        vid = args.force_vid if args.force_vid >= 0 else h.sim.IP_pos.shape[0] // 2
        h.sim.update_force(vid, torch.tensor(args.force, dtype=torch.float64, device=h.device))
    written, t0 = [], time.time()
    for f in range(args.frames):
        out = h.to_host(h.step(pose=pose, collect_stats=True))
        path = os.path.join(args.out, f"img_{f}.png")
        io.save_image(out["image"], path, args.W, args.H)
        written.append(path)
        if args.save_ip_state:  # what gui.py dumps and main_render.py:90-100 reads back
            m = h.model
            for name, t in (("ip_pos", m.p_def), ("ip_F", m.IP_F), ("ip_dF", m.IP_dF)):
                np.save(os.path.join(args.out, f"{name}_{f}.npy"), t.detach().cpu().numpy())
        if args.save_ply:
            h.synchronize()
            h.sim.OutputToPly(os.path.join(args.out, f"points_{f}.ply"))
    h.synchronize()
    if not args.quiet:
        print(f"{args.frames} frames -> {os.path.abspath(args.out)} in {time.time() - t0:.2f} s; last frame: {h.model.last_stats}")
    return written


def parser():
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--ply", default=None, help="simulation point cloud (x,y,z,mass,mu,lam,pin vertex properties); default: synthetic chair")
    ap.add_argument("--trust-ckpt", dest="trust_ckpt", action="store_true",
                    help="allow a checkpoint that needs arbitrary pickle globals (runs code from the file; default: tensors and plain scalars only)")
    ap.add_argument("--ckpt", default=None, help="a reference-format .pth or a checkpoints directory; default: synthetic chair checkpoint")
    ap.add_argument("--out", default="output_img/run")
    ap.add_argument("--frames", type=int, default=10)
    ap.add_argument("--W", type=int, default=800)
    ap.add_argument("--H", type=int, default=800)
    ap.add_argument("--radius", type=float, default=5.0)
    ap.add_argument("--azimuth", type=float, default=0.0)
    ap.add_argument("--elevation", type=float, default=0.0)
    ap.add_argument("--fovy", type=float, default=50.0)
    ap.add_argument("--bound", type=float, default=1.0)
    ap.add_argument("--dt_gamma", type=float, default=0.0)
    ap.add_argument("--max_steps", type=int, default=1024)
    ap.add_argument("--T_thresh", type=float, default=1e-2)
    ap.add_argument("--sim_dx", type=float, default=0.05)
    ap.add_argument("--sim_iters", type=int, default=10)
    ap.add_argument("--max_iter_num", type=int, default=1)
    ap.add_argument("--num_seek_IP", type=int, default=3)
    ap.add_argument("--force", type=float, nargs=3, default=None, help="constant force on one IP (gui.py drag), e.g. 300 100 -200")
    ap.add_argument("--force_vid", type=int, default=-1)
    ap.add_argument("--save_ply", action="store_true")
    ap.add_argument("--save_ip_state", action="store_true")
    ap.add_argument("--device", default="cuda:0")
    ap.add_argument("--quiet", action="store_true")
    return ap


if __name__ == "__main__":
    run(parser().parse_args())
