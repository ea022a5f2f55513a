"""Drop-in for the reference's ``_raymarching`` extension module (raymarching/src/bindings.cpp:5-19, signatures of raymarching.h:7-36):
the same positional arguments, outputs allocated (and zero-filled where the reference does) by the caller, nothing returned.  Every function is a
thin call into libpienerf_hip.so on torch's current stream (the reference launches on the legacy default stream)."""
import torch

from pienerf_amd._lib import check, lib, ptr, require_gpu, stream_ptr


def _c(*ts):
    """CHECK_CUDA / CHECK_CONTIGUOUS, and the element types the reference's kernels hard-code (data_ptr<float>() / <int>() / <uint8_t>() throw on anything
    else): a half or int64 tensor must not be reinterpreted."""
    for t in ts:
        if not t.is_contiguous():
            raise RuntimeError("expected a contiguous tensor")
        if t.dtype not in (torch.float32, torch.int32, torch.uint8):
            raise RuntimeError(f"expected a float32, int32 or uint8 tensor, got {t.dtype}")
    require_gpu(*ts)


def _f32(*ts):
    for t in ts:
        if t.dtype != torch.float32:
            raise RuntimeError(f"expected a float tensor, got {t.dtype}")


def _i32(*ts):
    for t in ts:
        if t.dtype != torch.int32:
            raise RuntimeError(f"expected an int tensor, got {t.dtype}")


def near_far_from_aabb(rays_o, rays_d, aabb, N, min_near, nears, fars):
    _c(rays_o, rays_d, aabb, nears, fars)
    check(lib().pn_near_far_from_aabb(ptr(rays_o), ptr(rays_d), ptr(aabb), int(N), float(min_near), ptr(nears), ptr(fars), stream_ptr()), "near_far_from_aabb")


def sph_from_ray(rays_o, rays_d, radius, N, coords):
    _c(rays_o, rays_d, coords)
    _f32(rays_o, rays_d, coords)
    check(lib().pn_sph_from_ray(ptr(rays_o), ptr(rays_d), float(radius), int(N), ptr(coords), stream_ptr()), "sph_from_ray")


def morton3D(coords, N, indices):
    _c(coords, indices)
    check(lib().pn_morton3D(ptr(coords), int(N), ptr(indices), stream_ptr()), "morton3D")


def morton3D_invert(indices, N, coords):
    _c(coords, indices)
    check(lib().pn_morton3D_invert(ptr(indices), int(N), ptr(coords), stream_ptr()), "morton3D_invert")


def packbits(grid, N, density_thresh, bitfield):
    _c(grid, bitfield)
    check(lib().pn_packbits(ptr(grid), int(N), float(density_thresh), ptr(bitfield), stream_ptr()), "packbits")


def march_rays_train(rays_o, rays_d, grid, bound, dt_gamma, max_steps, N, C, H, M, nears, fars, xyzs, dirs, deltas, rays, counter, noises):
    _c(rays_o, rays_d, grid, nears, fars, xyzs, dirs, deltas, rays, counter, noises)
    check(lib().pn_march_rays_train(ptr(rays_o), ptr(rays_d), ptr(grid), float(bound), float(dt_gamma), int(max_steps), int(N), int(C), int(H), int(M), ptr(nears),
                                    ptr(fars), ptr(xyzs), ptr(dirs), ptr(deltas), ptr(rays), ptr(counter), ptr(noises), stream_ptr()), "march_rays_train")


def composite_rays_train_forward(sigmas, rgbs, deltas, rays, M, N, T_thresh, weights_sum, depth, image):
    _c(sigmas, rgbs, deltas, rays, weights_sum, depth, image)
    check(lib().pn_composite_rays_train_forward(ptr(sigmas), ptr(rgbs), ptr(deltas), ptr(rays), int(M), int(N), float(T_thresh), ptr(weights_sum), ptr(depth),
                                                ptr(image), stream_ptr()), "composite_rays_train_forward")


def composite_rays_train_backward(grad_weights_sum, grad_image, sigmas, rgbs, deltas, rays, weights_sum, image, M, N, T_thresh, grad_sigmas, grad_rgbs):
    _c(grad_weights_sum, grad_image, sigmas, rgbs, deltas, rays, weights_sum, image, grad_sigmas, grad_rgbs)
    check(lib().pn_composite_rays_train_backward(ptr(grad_weights_sum), ptr(grad_image), ptr(sigmas), ptr(rgbs), ptr(deltas), ptr(rays), ptr(weights_sum), ptr(image),
                                                 int(M), int(N), float(T_thresh), ptr(grad_sigmas), ptr(grad_rgbs), stream_ptr()), "composite_rays_train_backward")


def march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, C, H, grid, nears, fars, xyzs, dirs, deltas, noises):
    _c(rays_alive, rays_t, rays_o, rays_d, grid, nears, fars, xyzs, dirs, deltas, noises)
    check(lib().pn_march_rays(int(n_alive), int(n_step), ptr(rays_alive), ptr(rays_t), ptr(rays_o), ptr(rays_d), float(bound), float(dt_gamma), int(max_steps), int(C),
                              int(H), ptr(grid), ptr(nears), ptr(fars), ptr(xyzs), ptr(dirs), ptr(deltas), ptr(noises), stream_ptr()), "march_rays")


def march_rays_quadratic_bending(pig_cnt, pig_bgn, pig_idx, n_vtx, n_grid, p_def, p_ori, F_IP, dF_IP, max_iter_num, bbmin, bbmax, hgs, resolution, num_seek_IP, IP_dx,
                                 cut, cut_bounds, n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, C, H, grid, near, far, xyzs, dirs,
                                 deltas, noises):
    _c(pig_cnt, pig_bgn, pig_idx, p_def, p_ori, F_IP, dF_IP, bbmin, bbmax, resolution, cut_bounds, rays_alive, rays_t, rays_o, rays_d, grid, near, far, xyzs, dirs,
       deltas, noises)
    _i32(pig_cnt, pig_bgn, pig_idx, resolution, rays_alive)
    _f32(p_def, p_ori, F_IP, dF_IP, bbmin, bbmax, cut_bounds, rays_t, rays_o, rays_d, near, far, xyzs, dirs, deltas, noises)
    if grid.dtype != torch.uint8:
        raise RuntimeError(f"expected a uint8 density bitfield, got {grid.dtype}")
    err = torch.zeros(1, dtype=torch.int32, device=xyzs.device)
    check(lib().pn_march_rays_quadratic_bending(ptr(pig_cnt), ptr(pig_bgn), ptr(pig_idx), int(n_vtx), int(n_grid), ptr(p_def), ptr(p_ori), ptr(F_IP), ptr(dF_IP),
                                                int(max_iter_num), ptr(bbmin), ptr(bbmax), float(hgs), ptr(resolution), int(num_seek_IP), float(IP_dx), int(bool(cut)),
                                                ptr(cut_bounds), int(n_alive), int(n_step), ptr(rays_alive), ptr(rays_t), ptr(rays_o), ptr(rays_d), float(bound),
                                                float(dt_gamma), int(max_steps), int(C), int(H), ptr(grid), ptr(near), ptr(far), ptr(xyzs), ptr(dirs), ptr(deltas),
                                                ptr(noises), ptr(err), stream_ptr()), "march_rays_quadratic_bending")
    # the reference blocks the host on an event in every call (raymarching.cu:1481-1482) and printf's "ERROR: g0=..." for points outside the spatial
    # hash (:1221-1222) and carries on; here the same wait reads the device flags: the printf case becomes a warning (same results as the reference:
    # such a sample finds no IP), anything else — a table that overflowed — raises
    flags = int(err.item())
    if flags & 1:
        import warnings
        warnings.warn("march_rays_quadratic_bending: a sample point fell outside the spatial hash (the reference prints 'ERROR: g0=...' and goes on)", RuntimeWarning)
    if flags & ~1:
        raise RuntimeError(f"march_rays_quadratic_bending: device error flags {flags:#x} (8: candidate-list capacity)")


def composite_rays(n_alive, n_step, T_thresh, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image):
    _c(rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image)
    check(lib().pn_composite_rays(int(n_alive), int(n_step), float(T_thresh), ptr(rays_alive), ptr(rays_t), ptr(sigmas), ptr(rgbs), ptr(deltas), ptr(weights_sum),
                                  ptr(depth), ptr(image), stream_ptr()), "composite_rays")
