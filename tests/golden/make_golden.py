#!/usr/bin/env python
"""Regenerates the golden fixtures in this directory.

    python tests/golden/make_golden.py

Provenance: the reference (FYTalon/pienerf) has no tests or golden vectors and its CUDA/Warp code cannot be imported or
built in the build container, so these vectors are produced by the CPU oracle (oracle/) from seeded
synthetic inputs (pienerf_amd.scene).  They are data only — inputs and expected outputs — and pin (a) the oracle against
regressions (tests/test_golden.py, CPU) and (b) the HIP path on the GPU box without needing anything but numpy
(tests/test_gpu_golden.py).  Inputs that are cheap to regenerate deterministically (the 47 MB hash table, the point
cloud) are not stored; their SHA-1 is, and the tests check it before use.
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

import oracle  # noqa: E402
from conftest import SMALL, make_oracle_sim  # noqa: E402
from pienerf_amd import scene  # noqa: E402


def sha(a):
    return hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    opt = scene.default_opt(sim_dx=SMALL["sim_dx"], sim_iters=SMALL["sim_iters"], W=48, H=48)
    cloud = scene.make_chair_points(sub_res=SMALL["sub_res"], hgs=opt["hash_grid_size"])
    ck = scene.make_checkpoint(bound=1.0, seed=0)
    meta = dict(cloud_pos_sha=sha(cloud["pos"]), emb_sha=sha(ck["embeddings"]), bits_sha=sha(ck["density_bitfield"]),
                W_sha=sha(np.concatenate([ck[f"W{i}"].ravel() for i in range(5)])))

    # ---- simulator: 3 substeps with a pick force
    s = make_oracle_sim(cloud, opt)
    p_ori, _, _ = s.get_IP_info()
    rest = dict(n_IP=s.n_IP, n_k=s.n_k, n_active=len(s.active), rhs_rest=s.rhs_rest.copy(), rhs_gravity=s.rhs_gravity.copy(),
                Nx_head=s.IP_Nx[:16].copy(), dNx_head=s.IP_dNx[:4].copy(), ddNx_head=s.IP_ddNx[:2].copy(), kernel_pos=s.kernel_pos.numpy().copy(),
                Ainv_diag=np.diag(s.Ainv).copy(), Mmat_rowsum=s.Mmat.sum(1))
    vid, f = s.n_IP // 2, np.array([300.0, 100.0, -200.0])
    s.update_force(vid, f)
    dofs = []
    for _ in range(3):
        s.stepforward()
        dofs.append(s.dof.copy())
    for _ in range(9):
        s.stepforward()
    p_def, F, dF = s.get_IP_info()
    np.savez_compressed(os.path.join(HERE, "sim_kat.npz"), force_vid=vid, force=f, dof_steps=np.stack(dofs), dof_vel_12=s.dof_vel.copy(),
                        p_def_12=p_def, F_12=F, dF_12=dF, **rest, **{k: np.array(v) for k, v in meta.items()})
    ip = dict(p_def=p_def, p_ori=p_ori, F=F, dF=dF, IP_dx=s.dx * 1.05)

    # ---- encoders / network on 96 points
    rng = np.random.default_rng(123)
    x = (rng.random((96, 3)).astype(np.float32) * 2 - 1) * 0.95
    x[:2] = [[0, 0, 0], [1.2, 0, 0]]
    d = rng.standard_normal((96, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    u = ((x + 1) / 2).astype(np.float32)
    feats = oracle.grid_encode_forward(u, ck["embeddings"], ck["offsets"], ck["per_level_scale"], ck["base_resolution"])
    sh = oracle.sh_encode_forward(d, 4)
    sig, rgb = oracle.nerf_forward(x, d, ck, 1.0)
    np.savez_compressed(os.path.join(HERE, "ops_kat.npz"), x=x, d=d, u=u, grid=feats, sh=sh, sigma=sig, rgb=rgb, offsets=ck["offsets"],
                        per_level_scale=ck["per_level_scale"], **{k: np.array(v) for k, v in meta.items()})

    # ---- march + frame on the deformed state
    W = 32
    pose, intr = scene.orbit_pose(5.0, 20.0, -15.0), scene.orbit_intrinsics(W, W, 50.0)
    o, dd = oracle.get_rays(pose, intr, W, W)
    hgs = np.float32(opt["hash_grid_size"])
    bbmin, bbmax, res = oracle.render_bbox(ip["p_def"], hgs)
    n_grid = int(res.prod())
    pig = oracle.get_pnts_in_grids(len(p_def), n_grid, p_def, bbmin, bbmax, hgs, res)
    nears, fars = oracle.near_far_from_aabb(o, dd, np.concatenate([bbmin, bbmax]), 0.2)
    alive = np.arange(W * W, dtype=np.int32)
    xyzs, dirs, deltas = oracle.march_rays_quadratic_bending(*pig, len(p_def), n_grid, p_def, p_ori, F, dF, 1, bbmin, bbmax, hgs, res, 3,
                                                             np.float32(ip["IP_dx"]), False, np.zeros(6, np.float32), W * W, 4, alive, nears, o, dd, 1.0,
                                                             ck["density_bitfield"], 1, 128, nears, fars, 128)
    fr = oracle.render_deformed(o, dd, ip, ck, opt)
    np.savez_compressed(os.path.join(HERE, "render_kat.npz"), pose=pose, intrinsics=intr, W=W, rays_o=o, rays_d=dd, p_def=p_def, p_ori=p_ori, F=F, dF=dF,
                        IP_dx=ip["IP_dx"], hgs=hgs, bbmin=bbmin, bbmax=bbmax, resolution=res, pig_cnt=pig[0], pig_bgn=pig[1], pig_idx=pig[2],
                        nears=nears, fars=fars, xyzs=xyzs, deltas=deltas, image=fr["image"], depth_0=fr["depth_0"], weights_sum=fr["weights_sum"],
                        trips=fr["trips"], samples=fr["samples"], **{k: np.array(v) for k, v in meta.items()})
    for f_ in ("sim_kat.npz", "ops_kat.npz", "render_kat.npz"):
        print(f_, os.path.getsize(os.path.join(HERE, f_)) // 1024, "KiB")


if __name__ == "__main__":
    main()
