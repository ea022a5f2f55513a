#!/usr/bin/env python
"""Golden fixtures produced by RUNNING THE REFERENCE ITSELF (build container only; /root/reference does not exist on the GPU box).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_ref.py

Three reference modules import cleanly here and are imported as they are:

  * get_opts.get_shared_opts            -> opts_chair.json / opts_trex.json : the option namespace of README.md:123 and :134
  * nerf.activation.trunc_exp           -> ref_kat.npz trunc_exp_*          : forward and the clamped backward (activation.py:5-18)

Every other reference module fails at import time on a third-party package that is not installed (cv2, trimesh, warp, dearpygui,
the CUDA extensions).  A few of their functions are nevertheless pure numpy / torch / scipy.  For those this script parses the
reference file with `ast`, compiles the ORIGINAL definition of the named function / class — unmodified, straight from the file
under /root/reference — into a namespace that holds only the real libraries it needs, and runs it on CPU:

  * nerf/utils.py     custom_meshgrid, get_rays (N = -1 path), linear_to_srgb, srgb_to_linear   -> ref_kat.npz rays_* / srgb_*
  * nerf/gui.py       OrbitCamera (pose, intrinsics, orbit, scale, pan)                          -> ref_kat.npz cam_*
  * nerf/provider.py  nerf_matrix_to_ngp                                                          -> ref_kat.npz ngp_*
  * gridencoder/grid.py  GridEncoder.__init__ (level offsets, per_level_scale)                    -> ref_kat.npz grid_*

Nothing of the reference's text is written anywhere: the outputs are data (inputs + the values the reference code returned).
tests/test_golden_ref.py holds the oracle and the product's host code to these vectors on CPU; tests/test_gpu_golden.py holds the
HIP kernels (k_get_rays) to them on the GPU.
"""
import ast
import json
import os
import sys

sys.dont_write_bytecode = True  # never leave __pycache__ inside /root/reference

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("PN_REFERENCE", "/root/reference")

CHAIR_ARGV = "--dataset_type synthetic --workspace model/chair --exp_name chair_0 -O --max_iter_num 1 --num_seek_IP 3 --sim_dx 0.05".split()  # README.md:123
TREX_ARGV = ("--path D:/Data/nerf_llff_data/trex --workspace model/trex --exp_name trex_0 -O --max_iter_num 1 --num_seek_IP 1 --sim_dx 0.05 --cut "
             "--cut_bounds -0.62 1.0 -0.82 0.42 -0.52 0.28 --max_steps 300 --T_thresh 5e-2 --W 1008 --H 756").split()                          # README.md:134


def extract(path, names, namespace):
    """Compile the top-level definitions `names` of reference file `path`, unmodified, into `namespace`."""
    with open(os.path.join(REF, path)) as f:
        tree = ast.parse(f.read(), filename=path)
    wanted = [n for n in tree.body if isinstance(n, (ast.FunctionDef, ast.ClassDef, ast.Assign)) and
              (getattr(n, "name", None) in names or (isinstance(n, ast.Assign) and any(getattr(t, "id", None) in names for t in n.targets)))]
    found = {getattr(n, "name", None) or n.targets[0].id for n in wanted}
    missing = set(names) - found
    if missing:
        raise RuntimeError(f"{path}: definitions not found: {sorted(missing)}")
    mod = ast.Module(body=wanted, type_ignores=[])
    namespace.setdefault("__name__", "reference_" + os.path.splitext(os.path.basename(path))[0])  # torch.jit.script wants a module name
    exec(compile(mod, os.path.join(REF, path), "exec"), namespace)
    return namespace


def opts(argv):
    import argparse
    sys.path.insert(0, REF)
    import get_opts  # the reference module itself
    old = sys.argv
    sys.argv = ["main_gui.py"] + argv
    try:
        o = get_opts.get_shared_opts(argparse.ArgumentParser())
    finally:
        sys.argv = old
    return {k: (list(v) if isinstance(v, (list, tuple)) else v) for k, v in sorted(vars(o).items())}


def main():
    out = {}
    # ---- 1. option namespaces (imported module)
    for name, argv in (("chair", CHAIR_ARGV), ("trex", TREX_ARGV)):
        with open(os.path.join(HERE, f"opts_{name}.json"), "w") as f:
            json.dump({"argv": argv, "opt": opts(argv)}, f, indent=1, sort_keys=True)
            f.write("\n")

    # ---- 2. trunc_exp (imported module)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        from nerf.activation import trunc_exp
    x = torch.tensor([-30.0, -15.5, -15.0, -3.25, -1e-3, 0.0, 0.5, 4.0, 14.75, 15.0, 15.5, 20.0, 88.0], dtype=torch.float32, requires_grad=True)
    g = torch.tensor([1.0, -2.0, 0.5, 3.0, 1.0, 1.0, -1.0, 0.25, 1.0, 2.0, 1.0, -1.0, 1.0], dtype=torch.float32)
    y = trunc_exp(x)
    y.backward(g)
    out.update(trunc_exp_x=x.detach().numpy(), trunc_exp_g=g.numpy(), trunc_exp_y=y.detach().numpy(), trunc_exp_dx=x.grad.numpy())
    xh = torch.tensor([-4.0, 0.33, 7.5], dtype=torch.float16)
    out.update(trunc_exp_xh=xh.numpy(), trunc_exp_yh=trunc_exp(xh).numpy())  # CPU: custom_fwd's cast only acts under CUDA autocast; dtype recorded as returned

    # ---- 3. get_rays & colour-space helpers (definitions compiled from nerf/utils.py)
    import packaging.version as pver
    ns = extract("nerf/utils.py", ["custom_meshgrid", "get_rays", "linear_to_srgb", "srgb_to_linear"], {"torch": torch, "pver": pver})
    # ---- 4. OrbitCamera (definitions compiled from nerf/gui.py)
    from scipy.spatial.transform import Rotation as R
    cam_ns = extract("nerf/gui.py", ["OrbitCamera"], {"np": np, "R": R})
    OrbitCamera = cam_ns["OrbitCamera"]
    cam = OrbitCamera(800, 800, r=5, fovy=50)  # NeRFSimGUI: OrbitCamera(opt.W, opt.H, r=opt.radius, fovy=opt.fovy)
    out.update(cam_pose_default=cam.pose.astype(np.float32), cam_intrinsics_800=np.asarray(cam.intrinsics, np.float64))
    cam2 = OrbitCamera(1008, 756, r=5, fovy=50)
    out.update(cam_intrinsics_trex=np.asarray(cam2.intrinsics, np.float64))
    cam.orbit(250.0, -120.0)
    out.update(cam_pose_orbit=cam.pose.astype(np.float32))
    cam.scale(3.0)
    cam.pan(12.0, -7.0, 2.0)
    out.update(cam_pose_orbit_scale_pan=cam.pose.astype(np.float32), cam_radius_after=np.float64(cam.radius), cam_center_after=cam.center.astype(np.float64))
    for tag, pose, (W, H), intr in (("a", out["cam_pose_default"], (16, 12), np.array([14.0, 13.0, 8.0, 6.0])),
                                    ("b", out["cam_pose_orbit_scale_pan"], (10, 14), np.array([9.5, 11.25, 5.0, 7.0])),
                                    ("c", out["cam_pose_orbit"], (800, 800), out["cam_intrinsics_800"])):
        r = ns["get_rays"](torch.from_numpy(pose).unsqueeze(0), intr, H, W, -1)
        if tag == "c":  # full frame: keep a strided subset (the generator of the subset is the index list itself)
            idx = np.arange(0, W * H, 4999)
            out.update(rays_c_idx=idx, rays_c_o=r["rays_o"][0].numpy()[idx], rays_c_d=r["rays_d"][0].numpy()[idx])
        else:
            out.update({f"rays_{tag}_o": r["rays_o"][0].numpy().copy(), f"rays_{tag}_d": r["rays_d"][0].numpy().copy()})
        out.update({f"rays_{tag}_pose": pose, f"rays_{tag}_intr": np.asarray(intr, np.float64), f"rays_{tag}_WH": np.array([W, H])})
    v = torch.tensor([0.0, 0.001, 0.0031308, 0.004, 0.04045, 0.05, 0.2, 0.5, 0.9, 1.0], dtype=torch.float32)
    out.update(srgb_in=v.numpy(), srgb_lin2srgb=ns["linear_to_srgb"](v).numpy(), srgb_srgb2lin=ns["srgb_to_linear"](v).numpy())

    # ---- 5. nerf_matrix_to_ngp (definition compiled from nerf/provider.py)
    p_ns = extract("nerf/provider.py", ["nerf_matrix_to_ngp"], {"np": np})
    rng = np.random.default_rng(7)
    q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
    pose = np.eye(4, dtype=np.float32)  # nerf/provider.py builds its poses as float32 arrays before this call
    pose[:3, :3] = q
    pose[:3, 3] = [1.5, -2.0, 4.0]
    out.update(ngp_in=pose, ngp_out_033=p_ns["nerf_matrix_to_ngp"](pose, scale=0.33, offset=[0, 0, 0]),
               ngp_out_08=p_ns["nerf_matrix_to_ngp"](pose, scale=0.8, offset=[0.1, -0.2, 0.3]))

    # ---- 6. hash-grid level layout (class compiled from gridencoder/grid.py; only __init__ runs — forward needs the CUDA backend)
    import torch.nn as nn
    g_ns = extract("gridencoder/grid.py", ["_gridtype_to_id", "_interp_to_id", "GridEncoder"], {"np": np, "torch": torch, "nn": nn})
    for tag, bound in (("b1", 1.0), ("b2", 2.0)):  # nerf/network.py:34: desired_resolution = 2048 * bound
        enc = g_ns["GridEncoder"](input_dim=3, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=19, desired_resolution=2048 * bound,
                                  gridtype="hash", align_corners=False)
        out.update({f"grid_{tag}_offsets": enc.offsets.numpy().astype(np.int64), f"grid_{tag}_per_level_scale": np.float64(enc.per_level_scale),
                    f"grid_{tag}_output_dim": np.int64(enc.output_dim), f"grid_{tag}_n_embeddings": np.int64(enc.embeddings.shape[0])})
    # ---- 7. names and arities of the three pybind backends (read from the reference's headers / bindings.cpp; data, not code)
    import re
    bind = {}
    for mod, bfile, hfile in (("_raymarching", "raymarching/src/bindings.cpp", "raymarching/src/raymarching.h"),
                              ("_gridencoder", "gridencoder/src/bindings.cpp", "gridencoder/src/gridencoder.h"),
                              ("_shencoder", "shencoder/src/bindings.cpp", "shencoder/src/shencoder.h")):
        names = re.findall(r'm\.def\("(\w+)"', open(os.path.join(REF, bfile)).read())
        header = re.sub(r"//[^\n]*", "", open(os.path.join(REF, hfile)).read())
        bind[mod] = {}
        for n in names:
            m = re.search(r"void\s+" + n + r"\s*\(([^;]*?)\)\s*;", header, re.S)
            params = [p_.strip().split()[-1] for p_ in m.group(1).split(",") if p_.strip()]
            bind[mod][n] = params
    with open(os.path.join(HERE, "ref_bindings.json"), "w") as f:
        json.dump(bind, f, indent=1, sort_keys=True)
        f.write("\n")
    np.savez_compressed(os.path.join(HERE, "ref_kat.npz"), **out)
    print("wrote opts_chair.json, opts_trex.json, ref_kat.npz:", sorted(out))


if __name__ == "__main__":
    main()
