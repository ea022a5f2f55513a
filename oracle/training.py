"""ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED.

numpy-level wrappers of the training-side restatements in render_oracle.cpp: march_rays_train,
composite_rays_train forward / backward (raymarching/src/raymarching.cu:314-700, raymarching/raymarching.py:163-292), the hash
grid's dy_dx / backward / total-variation gradient (gridencoder/src/gridencoder.cu:199-245,248-369,506-611, gridencoder/grid.py:24-92,
168-190) and the SH encoder's dy_dx / backward (shencoder/src/shencoder.cu:125-383).  Only tests/ import this module.
"""
import ctypes as C

import numpy as np

from . import F, I, U8, _f32, _i32, _p, lib

U32 = C.c_uint32


def march_rays_train(rays_o, rays_d, bound, density_bitfield, Cc, H, nears, fars, step_counter=None, mean_count=-1, noises=None, align=-1,
                     force_all_rays=False, dt_gamma=0.0, max_steps=1024):
    """_march_rays_train.forward (raymarching.py:163-236) with the noise vector passed in (perturb=False <=> None).
    Returns xyzs, dirs, deltas, rays; rays keep ray order (see orc_march_rays_train)."""
    rays_o, rays_d = _f32(rays_o).reshape(-1, 3), _f32(rays_d).reshape(-1, 3)
    N = rays_o.shape[0]
    M = N * int(max_steps)
    if not force_all_rays and mean_count > 0:
        if align > 0:
            mean_count += align - mean_count % align
        M = int(mean_count)
    xyzs, dirs, deltas = np.zeros((M, 3), np.float32), np.zeros((M, 3), np.float32), np.zeros((M, 2), np.float32)
    rays = np.empty((N, 3), np.int32)
    counter = np.zeros(2, np.int32) if step_counter is None else step_counter
    assert counter.dtype == np.int32 and counter.flags.c_contiguous
    base = counter.copy()
    noises = np.zeros(N, np.float32) if noises is None else _f32(noises)
    grid = np.ascontiguousarray(density_bitfield, np.uint8)
    nears, fars = _f32(nears), _f32(fars)
    # the C function appends ray rows at counter[1]; the wrapper always starts a fresh `rays`, so run it on a zeroed ray counter
    tmp = np.array([base[0] * 0, 0], np.int32)
    lib().orc_march_rays_train(_p(rays_o, F), _p(rays_d, F), _p(grid, U8), F(bound), F(dt_gamma), U32(int(max_steps)), U32(N), U32(int(Cc)), U32(int(H)),
                               U32(M), _p(nears, F), _p(fars, F), _p(xyzs, F), _p(dirs, F), _p(deltas, F), _p(rays, I), _p(tmp, I), _p(noises, F))
    counter[0] = base[0] + tmp[0]
    counter[1] = base[1] + tmp[1]
    if force_all_rays or mean_count <= 0:
        m = int(tmp[0])
        if align > 0:
            m += align - m % align
        xyzs, dirs, deltas = xyzs[:m], dirs[:m], deltas[:m]
    return xyzs, dirs, deltas, rays


def composite_rays_train_forward(sigmas, rgbs, deltas, rays, T_thresh=1e-4):
    sigmas, rgbs, deltas, rays = _f32(sigmas), _f32(rgbs).reshape(-1, 3), _f32(deltas).reshape(-1, 2), _i32(rays).reshape(-1, 3)
    M, N = sigmas.shape[0], rays.shape[0]
    ws, depth, image = np.empty(N, np.float32), np.empty(N, np.float32), np.empty((N, 3), np.float32)
    lib().orc_composite_rays_train_forward(_p(sigmas, F), _p(rgbs, F), _p(deltas, F), _p(rays, I), U32(M), U32(N), F(T_thresh), _p(ws, F), _p(depth, F),
                                           _p(image, F))
    return ws, depth, image


def composite_rays_train_backward(grad_weights_sum, grad_image, sigmas, rgbs, deltas, rays, weights_sum, image, T_thresh=1e-4):
    sigmas, rgbs, deltas, rays = _f32(sigmas), _f32(rgbs).reshape(-1, 3), _f32(deltas).reshape(-1, 2), _i32(rays).reshape(-1, 3)
    gws, gim, ws, im = _f32(grad_weights_sum), _f32(grad_image).reshape(-1, 3), _f32(weights_sum), _f32(image).reshape(-1, 3)
    M, N = sigmas.shape[0], rays.shape[0]
    gs, gc = np.zeros(M, np.float32), np.zeros((M, 3), np.float32)
    lib().orc_composite_rays_train_backward(_p(gws, F), _p(gim, F), _p(sigmas, F), _p(rgbs, F), _p(deltas, F), _p(rays, I), _p(ws, F), _p(im, F), U32(M),
                                            U32(N), F(T_thresh), _p(gs, F), _p(gc, F))
    return gs, gc


def grid_encode_dy_dx(inputs, embeddings, offsets, per_level_scale, base_resolution, gridtype=0, align_corners=False, interpolation=0):
    """dy_dx [B, L*3*C] as `_grid_encode.forward` saves it (grid.py:49-52)."""
    inputs, embeddings, offsets = _f32(inputs).reshape(-1, 3), _f32(embeddings), _i32(offsets)
    B, L, Cf = inputs.shape[0], offsets.shape[0] - 1, embeddings.shape[1]
    out = np.empty((B, L * 3 * Cf), np.float32)
    lib().orc_grid_encode_dy_dx(_p(inputs, F), _p(embeddings, F), _p(offsets, I), _p(out, F), U32(B), U32(Cf), U32(L), F(np.float32(np.log2(per_level_scale))),
                                U32(base_resolution), U32(gridtype), I(int(align_corners)), U32(interpolation))
    return out


def grid_encode_backward(grad, inputs, embeddings_shape, offsets, per_level_scale, base_resolution, dy_dx=None, gridtype=0, align_corners=False,
                         interpolation=0):
    """`_grid_encode.backward` (grid.py:65-90): grad [B, L*C] -> (grad_inputs [B,3] or None, grad_embeddings [sO, C])."""
    inputs, offsets = _f32(inputs).reshape(-1, 3), _i32(offsets)
    B, L, Cf = inputs.shape[0], offsets.shape[0] - 1, embeddings_shape[1]
    g = np.ascontiguousarray(_f32(grad).reshape(B, L, Cf).transpose(1, 0, 2))  # [L, B, C]
    ge = np.zeros(tuple(embeddings_shape), np.float32)
    gi = np.zeros((B, 3), np.float32) if dy_dx is not None else None
    dd = _f32(dy_dx) if dy_dx is not None else None
    lib().orc_grid_encode_backward(_p(g, F), _p(inputs, F), _p(offsets, I), _p(ge, F), U32(B), U32(Cf), U32(L), F(np.float32(np.log2(per_level_scale))),
                                   U32(base_resolution), _p(dd, F) if dd is not None else None, _p(gi, F) if gi is not None else None, U32(gridtype),
                                   I(int(align_corners)), U32(interpolation))
    return gi, ge


def grad_total_variation(inputs, embeddings, grad, offsets, per_level_scale, base_resolution, weight=1e-7, gridtype=0, align_corners=False):
    """GridEncoder.grad_total_variation (grid.py:168-190) on inputs already in [0,1]; accumulates into `grad` in place."""
    inputs, embeddings, offsets = _f32(inputs).reshape(-1, 3), _f32(embeddings), _i32(offsets)
    assert grad.dtype == np.float32 and grad.flags.c_contiguous and grad.shape == embeddings.shape
    B, L, Cf = inputs.shape[0], offsets.shape[0] - 1, embeddings.shape[1]
    lib().orc_grad_total_variation(_p(inputs, F), _p(embeddings, F), _p(grad, F), _p(offsets, I), F(weight), U32(B), U32(Cf), U32(L),
                                   F(np.float32(np.log2(per_level_scale))), U32(base_resolution), U32(gridtype), I(int(align_corners)))
    return grad


def sh_encode_dy_dx(inputs, degree=4):
    inputs = _f32(inputs).reshape(-1, 3)
    out = np.empty((inputs.shape[0], 3 * degree * degree), np.float32)
    lib().orc_sh_encode_dy_dx(_p(inputs, F), _p(out, F), U32(inputs.shape[0]), U32(degree))
    return out


def sh_encode_backward(grad, dy_dx, degree=4):
    grad, dy_dx = _f32(grad).reshape(-1, degree * degree), _f32(dy_dx)
    B = grad.shape[0]
    gi = np.zeros((B, 3), np.float32)
    lib().orc_sh_encode_backward(_p(grad, F), U32(B), U32(degree), _p(dy_dx, F), _p(gi, F))
    return gi


def trunc_exp_forward(x):
    """nerf/activation.py:8-10: exp of the float32-cast input."""
    import torch
    return torch.exp(torch.from_numpy(_f32(x))).numpy()


def trunc_exp_backward(x, g):
    """nerf/activation.py:14-16: g * exp(clamp(x, -15, 15))."""
    import torch
    return (torch.from_numpy(_f32(g)) * torch.exp(torch.from_numpy(_f32(x)).clamp(-15, 15))).numpy()


# ---------------------------------------------------------------------------------------------- density-grid state (renderer.py:390-549)
def _morton_invert(idx):
    def compact(x):
        x = x & 0x49249249
        x = (x | (x >> 2)) & 0xc30c30c3
        x = (x | (x >> 4)) & 0x0f00f00f
        x = (x | (x >> 8)) & 0xff0000ff
        x = (x | (x >> 16)) & 0x0000ffff
        return x
    idx = np.asarray(idx, np.uint32)
    return np.stack([compact(idx), compact(idx >> 1), compact(idx >> 2)], -1)


def _cell_centres(cas, H, bound):
    """Cascade-space centres of all H^3 cells in morton order: (2 c / (H - 1) - 1) * (bound_c - bound_c / H), float32 op by op
    (renderer.py:420,425-428 / :481,486-489)."""
    f = np.float32
    c = _morton_invert(np.arange(H ** 3, dtype=np.uint32)).astype(f)
    bnd = f(min(2 ** cas, bound))
    half = f(bnd / f(H))
    return (f(2.0) * c / f(H - 1) - f(1.0)) * f(bnd - half), half


def mark_untrained_grid(poses, intrinsic, cascade, H, bound):
    """NeRFRenderer.mark_untrained_grid (renderer.py:390-452): boolean [cascade, H^3] (morton order), True where no camera sees the cell.
    float32 arithmetic in the order of the reference's tensor ops; the batched matmul `cam @ R` is summed over the three rows in order."""
    f = np.float32
    poses = np.asarray(poses, f).reshape(-1, 4, 4)
    fx, fy, cx, cy = (float(v) for v in intrinsic)
    cxfx, cyfy = f(cx / fx), f(cy / fy)
    out = np.zeros((cascade, H ** 3), bool)
    for cas in range(cascade):
        w, half = _cell_centres(cas, H, bound)
        pad = f(half * f(2.0))
        count = np.zeros(H ** 3, np.int64)
        for P in poses:
            d = w - P[:3, 3][None, :]
            cam = [f(0)] * 3
            for j in range(3):
                cam[j] = (d[:, 0] * P[0, j] + d[:, 1] * P[1, j]) + d[:, 2] * P[2, j]
            count += (cam[2] > 0) & (np.abs(cam[0]) < cxfx * cam[2] + pad) & (np.abs(cam[1]) < cyfy * cam[2] + pad)
        out[cas] = count == 0
    return out


def density_cells_full(cascade, H, bound, noise):
    """The jittered cell samples of update_extra_state's full sweep (renderer.py:486-491): [cascade * H^3, 3], row = cas * H^3 + morton."""
    f = np.float32
    noise = np.asarray(noise, f).reshape(cascade, H ** 3, 3)
    out = np.empty((cascade, H ** 3, 3), f)
    for cas in range(cascade):
        w, half = _cell_centres(cas, H, bound)
        out[cas] = w + (noise[cas] * f(2.0) - f(1.0)) * half
    return out.reshape(-1, 3)


def density_grid_update(grid, tmp, decay, density_thresh):
    """renderer.py:535-543: EMA-max where both are valid, mean of the clamped grid, threshold, bitfield."""
    f = np.float32
    grid, tmp = np.asarray(grid, f).copy(), np.asarray(tmp, f)
    valid = (grid >= 0) & (tmp >= 0)
    grid[valid] = np.maximum(grid[valid] * f(decay), tmp[valid])
    mean = float(np.mean(np.maximum(grid, 0).astype(np.float64)))
    thresh = min(mean, float(density_thresh))
    bits = np.packbits((grid.reshape(-1) > f(thresh)).reshape(-1, 8), axis=1, bitorder="little").reshape(-1)  # bit i of byte n = cell 8n+i (raymarching.cu:270-292)
    return grid, mean, bits
