"""TEST INFRASTRUCTURE ONLY — CPU restatement of /root/reference/main_sample.py (AdaptiveUniformSampling.sample, :202-308, and the
Warp kernels :29-140) for the parity tests of pienerf_amd/sampling.py.  Parity unpinned: the reference ships no test or vector for it.

numpy / torch-CPU, one function per reference kernel, fp32 arithmetic in the reference's order.  Where the reference's behaviour is
undefined (out-of-range scatter in get_grid_coords, reads past the end of grid_density in get_sub_grid) or racy (get_sub_bgn's atomic
ranges), this restatement takes the same documented decisions as the mirror: in-range cells only, out-of-range neighbours count 0,
ranges in cell order.  `density_fn(points [n,3] fp32) -> sigma [n]` supplies NeRFNetwork.density(...)['sigma']; `rand` the uniform
numbers of torch.rand (:268).
"""
import numpy as np
import torch

F32 = np.float32


def nerf_density(xyzs, ckpt, bound):
    """NeRFNetwork.density (nerf/network.py:129-146): hash grid -> Linear(32,64) -> ReLU -> Linear(64,16); sigma = exp(h0)."""
    from . import grid_encode_forward
    x = np.asarray(xyzs, F32).reshape(-1, 3)
    u = (x + F32(bound)) / F32(2 * bound)                                     # gridencoder/grid.py:149
    enc = grid_encode_forward(u, ckpt["embeddings"], ckpt["offsets"], ckpt["per_level_scale"], ckpt["base_resolution"]).reshape(len(x), -1)
    h = np.maximum(enc.astype(F32) @ np.asarray(ckpt["W0"], F32).T, F32(0))
    out = h @ np.asarray(ckpt["W1"], F32).T
    return np.exp(out[:, 0]).astype(F32), out[:, 1:].astype(F32)


def lattice(opt):
    """main_sample.py:204-226 — torch.linspace on the CPU, exactly as the reference builds it."""
    res, bound = int(opt["sub_res"]), float(opt["bound"])
    if opt.get("cut", False):
        cb = list(opt["cut_bounds"])
        for a in (0, 2, 4):
            cb[a] = max(cb[a], -bound)
        for a in (1, 3, 5):
            cb[a] = min(cb[a], bound)
        xs, ys, zs = (torch.linspace(cb[2 * a], cb[2 * a + 1], res).numpy() for a in range(3))
    else:
        xs = ys = zs = torch.linspace(-bound, bound, res).numpy()
    pts = np.empty((res, res, res, 3), F32)                                   # [i,j,k] = (x_k, y_j, z_i)
    pts[..., 0] = xs[None, None, :]
    pts[..., 1] = ys[None, :, None]
    pts[..., 2] = zs[:, None, None]
    return pts.reshape(-1, 3)


def _hash_g(g, res):
    r = F32(res)
    return (g[..., 2].astype(F32) * r * r + g[..., 1].astype(F32) * r + g[..., 0].astype(F32)).astype(np.int64)   # :45-46


def sample(opt, density_fn, rand):
    res, bound = int(opt["sub_res"]), F32(opt["bound"])
    grid_size = F32(2 * float(opt["bound"]) / res)
    n_grid = res ** 3
    grid_pts = lattice(opt)
    dens = lambda p: (F32(1) - np.exp(-np.asarray(density_fn(p), F32) / F32(128.0))).astype(F32)                 # get_density :164-168
    grid_density = dens(grid_pts)
    # get_grid_coords :50-66
    grid_coords = np.zeros((n_grid, 3), np.int32)
    for n in range(n_grid):
        g = np.floor((grid_pts[n] + bound) / grid_size).astype(np.int32)
        if (g >= 0).all() and (g < res).all():
            grid_coords[_hash_g(g, res)] = g
    # get_sub_grid :101-140
    sub_mins, sub_maxs, sub_dims = np.zeros((n_grid, 3), F32), np.zeros((n_grid, 3), F32), np.zeros(n_grid, np.int32)
    offs = [(0, 0, 0), (0, 0, 1), (0, 1, 0), (0, 1, 1), (1, 0, 0), (1, 0, 1), (1, 1, 0), (1, 1, 1)]
    for gid in range(n_grid):
        g0 = grid_coords[gid]
        d = []
        for o in offs:
            h = int(_hash_g(g0 + np.array(o, np.int32), res))
            d.append(grid_density[h] if h < n_grid else F32(0))
        gx = (d[4] + d[5] + d[6] + d[7]) - (d[0] + d[1] + d[2] + d[3])
        gy = (d[2] + d[3] + d[6] + d[7]) - (d[0] + d[1] + d[4] + d[5])
        gz = (d[1] + d[3] + d[5] + d[7]) - (d[0] + d[2] + d[4] + d[6])
        gn = np.sqrt(F32(gx * gx + gy * gy + gz * gz))
        if gn == 0.0:
            continue
        sub_mins[gid] = g0.astype(F32) * grid_size - bound
        sub_maxs[gid] = (g0 + 1).astype(F32) * grid_size - bound
        sub_dims[gid] = np.int32(F32(F32(F32(sub_maxs[gid][0] - sub_mins[gid][0]) * F32(opt["sub_coeff"])) * F32(res)) * gn)
    # get_sub_bgn :74-82 (cell order instead of atomic order) + get_pnts_add :84-99
    rand = np.asarray(rand, F32)
    chunks = []
    for gid in range(n_grid):
        k = int(sub_dims[gid]) ** 3
        if k:
            chunks.append((sub_maxs[gid] - sub_mins[gid])[None, :] * rand[:k] + sub_mins[gid][None, :])
    pnts_add = np.concatenate(chunks, 0).astype(F32) if chunks else np.zeros((0, 3), F32)
    # :288-295
    pts = np.concatenate([pnts_add, grid_pts + F32(0.5 * 2 * float(opt["bound"]) / float(res))], 0)
    keep = dens(pts) > F32(opt["density_threshold"])
    pts = pts[keep]
    return pts, point_volumes(pts, opt), dict(boundary_points=len(pnts_add), kept=len(pts))


def point_volumes(pts, opt):
    """get_point_volumes :182-200 with the spatial hash of nerf/utils.py:355-443 restated in place (fp32 cell index)."""
    pts = np.asarray(pts, F32)
    bbmin = pts.min(0) - F32(1e-3) * np.ones(3, F32)
    bbmax = pts.max(0) + F32(1e-3) * np.ones(3, F32)
    hgs = float(opt["hash_grid_size"])
    resolution = np.ceil((bbmax - bbmin) / F32(hgs)).astype(np.int32)
    g = np.floor((pts - bbmin[None, :]) / F32(hgs)).astype(np.int64)
    cell = g[:, 2] * int(resolution[1]) * int(resolution[0]) + g[:, 1] * int(resolution[0]) + g[:, 0]
    cnt = np.bincount(cell, minlength=int(resolution[0]) * int(resolution[1]) * int(resolution[2]))
    with np.errstate(divide="ignore"):
        vol = (hgs ** 3 / cnt.astype(F32)).astype(F32)
    return vol[cell]
