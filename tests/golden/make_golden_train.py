"""Generates tests/golden/train_kat.npz: known-answer vectors of the widened ops produced by the CPU oracle from seeded
synthetic inputs — the static march, march_rays_train, composite_rays_train forward / backward, the hash grid's dy_dx / backward /
total-variation gradient, the SH encoder's dy_dx / backward, packbits and morton3D.  The reference ships no vectors of its own;
the fixture pins the oracle against regressions (tests/test_golden.py) and gives the HIP path a fixed target (tests/test_gpu_golden.py).

    python tests/golden/make_golden_train.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle  # noqa: E402
from oracle import training as otr  # noqa: E402
from pienerf_amd import scene  # noqa: E402
from pienerf_amd.gridencoder.grid import level_table_offsets  # noqa: E402


def build():
    rng = np.random.default_rng(2024)
    ck = scene.make_checkpoint(bound=1.0, seed=0)
    W = 20
    o, d = oracle.get_rays(scene.orbit_pose(3.4, 25.0, -20.0), scene.orbit_intrinsics(W, W, 50.0), W, W)
    nears, fars = oracle.near_far_from_aabb(o, d, np.array([-1, -1, -1, 1, 1, 1], np.float32), 0.2)
    out = dict(rays_o=o, rays_d=d, nears=nears, fars=fars)
    noise = rng.random(len(o)).astype(np.float32)
    counter = np.zeros(2, np.int32)
    xyzs, dirs, deltas, rays = otr.march_rays_train(o, d, 1.0, ck["density_bitfield"], 1, 128, nears, fars, counter, -1, noise, 128, False, 0.0, 256)
    out.update(noise=noise, train_xyzs=xyzs, train_deltas=deltas, train_rays=rays, train_counter=counter)
    alive = np.arange(0, len(o), 2, dtype=np.int32)
    sx, sd, sl = oracle.march_rays(len(alive), 8, alive, nears, o, d, 1.0, ck["density_bitfield"], 1, 128, nears, fars, 128, None, 0.0, 256)
    out.update(static_alive=alive, static_xyzs=sx, static_deltas=sl)
    M = int(counter[0])
    sig = rng.uniform(0, 40, M).astype(np.float32)
    rgb = rng.uniform(0, 1, (M, 3)).astype(np.float32)
    ws, depth, image = otr.composite_rays_train_forward(sig, rgb, deltas[:M], rays, 1e-2)
    gws, gim = rng.standard_normal(len(o)).astype(np.float32), rng.standard_normal((len(o), 3)).astype(np.float32)
    gs, gc = otr.composite_rays_train_backward(gws, gim, sig, rgb, deltas[:M], rays, ws, image, 1e-2)
    out.update(sig=sig, rgb=rgb, comp_ws=ws, comp_depth=depth, comp_image=image, gws=gws, gim=gim, comp_gs=gs, comp_gc=gc)
    pls, base, L = 1.6, 8, 6
    offsets = level_table_offsets(3, L, pls, base, 12, False)
    emb = rng.uniform(-1, 1, (int(offsets[-1]), 2)).astype(np.float32)
    x = rng.uniform(0, 1, (256, 3)).astype(np.float32)
    x[0] = [1.1, 0.5, 0.5]
    grad = rng.standard_normal((256, L * 2)).astype(np.float32)
    dy_dx = otr.grid_encode_dy_dx(x, emb, offsets, pls, base)
    gi, ge = otr.grid_encode_backward(grad, x, emb.shape, offsets, pls, base, dy_dx)
    tv = otr.grad_total_variation(x, emb, np.zeros_like(emb), offsets, pls, base, weight=0.3)
    out.update(grid_offsets=offsets, grid_emb=emb, grid_x=x, grid_grad=grad, grid_dy_dx=dy_dx, grid_gi=gi, grid_ge=ge, grid_tv=tv)
    dirs3 = rng.standard_normal((128, 3))
    dirs3 = (dirs3 / np.linalg.norm(dirs3, axis=1, keepdims=True)).astype(np.float32)
    shg = rng.standard_normal((128, 16)).astype(np.float32)
    sh_dy = otr.sh_encode_dy_dx(dirs3, 4)
    out.update(sh_dirs=dirs3, sh_grad=shg, sh_dy_dx=sh_dy, sh_gi=otr.sh_encode_backward(shg, sh_dy, 4))
    g = rng.random((1, 4096)).astype(np.float32)
    c = rng.integers(0, 128, (300, 3)).astype(np.int32)
    out.update(pack_grid=g, pack_bits=oracle.packbits(g, 0.5), mort_coords=c, mort_idx=oracle.morton3D(c))
    return out


if __name__ == "__main__":
    np.savez_compressed(os.path.join(HERE, "train_kat.npz"), **build())
    print("wrote", os.path.join(HERE, "train_kat.npz"))
