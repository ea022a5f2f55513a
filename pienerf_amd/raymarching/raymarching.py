"""Drop-in mirror of the reference's ``raymarching`` package for the simulate-and-render path.

Same function names, argument order and in-place/ownership behaviour as
/root/reference/raymarching/raymarching.py (cited per function); the native side is
libpienerf_hip.so (include/pienerf_hip.h) instead of the pybind ``_raymarching`` module.
Wrappers allocate every output, exactly like the reference.  GPU tensors only.
"""
import torch

from .._lib import check, lib, ptr, require_gpu, stream_ptr

__all__ = ["near_far_from_aabb", "sph_from_ray", "march_rays_quadratic_bending", "march_rays", "composite_rays", "compact_rays", "morton3D", "morton3D_invert",
           "packbits", "march_rays_train", "composite_rays_train"]


def _f32(t):
    return t.to(torch.float32).contiguous()  # custom_fwd(cast_inputs=torch.float32) in the reference


def near_far_from_aabb(rays_o, rays_d, aabb, min_near=0.2):
    """raymarching/raymarching.py:21-51.  rays_o, rays_d [N,3]; aabb [6] -> nears [N], fars [N]."""
    if not rays_o.is_cuda:
        rays_o = rays_o.cuda()
    if not rays_d.is_cuda:
        rays_d = rays_d.cuda()
    rays_o = _f32(rays_o).view(-1, 3)
    rays_d = _f32(rays_d).view(-1, 3)
    aabb = _f32(aabb)
    require_gpu(rays_o, rays_d, aabb)
    N = rays_o.shape[0]
    nears = torch.empty(N, dtype=rays_o.dtype, device=rays_o.device)
    fars = torch.empty(N, dtype=rays_o.dtype, device=rays_o.device)
    check(lib().pn_near_far_from_aabb(ptr(rays_o), ptr(rays_d), ptr(aabb), N, float(min_near), ptr(nears), ptr(fars), stream_ptr()),
          "near_far_from_aabb")
    return nears, fars


def sph_from_ray(rays_o, rays_d, radius):
    """raymarching/raymarching.py:54-82.  rays_o, rays_d [N,3] -> coords [N,2] in [-1,1] (theta, phi of the point where the ray leaves the sphere)."""
    if not rays_o.is_cuda:
        rays_o = rays_o.cuda()
    if not rays_d.is_cuda:
        rays_d = rays_d.cuda()
    rays_o = _f32(rays_o).view(-1, 3)
    rays_d = _f32(rays_d).view(-1, 3)
    require_gpu(rays_o, rays_d)
    N = rays_o.shape[0]
    coords = torch.empty(N, 2, dtype=rays_o.dtype, device=rays_o.device)
    check(lib().pn_sph_from_ray(ptr(rays_o), ptr(rays_d), float(radius), N, ptr(coords), stream_ptr()), "sph_from_ray")
    return coords


def march_rays_quadratic_bending(pig_cnt, pig_bgn, pig_idx, n_vtx, n_grid, p_def, p_ori, F_IP, dF_IP, max_iter_num, bbmin, bbmax, hgs, res,
                                 num_seek_IP, IP_dx, cut, cut_bounds, n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound,
                                 density_bitfield, C, H, near, far, align=-1, perturb=False, dt_gamma=0, max_steps=1024):
    """raymarching/raymarching.py:387-441.  Returns zero-initialised xyzs [M,3], dirs [M,3], deltas [M,2]."""
    if not rays_o.is_cuda:
        rays_o = rays_o.cuda()
    if not rays_d.is_cuda:
        rays_d = rays_d.cuda()
    rays_o = _f32(rays_o).view(-1, 3)
    rays_d = _f32(rays_d).view(-1, 3)
    n_alive, n_step, n_grid = int(n_alive), int(n_step), int(n_grid)
    M = n_alive * n_step
    if align > 0:
        M += align - (M % align)
    dev = rays_o.device
    xyzs = torch.zeros(M, 3, dtype=rays_o.dtype, device=dev)
    dirs = torch.zeros(M, 3, dtype=rays_o.dtype, device=dev)
    deltas = torch.zeros(M, 2, dtype=rays_o.dtype, device=dev)
    if perturb:
        noises = torch.rand(n_alive, dtype=rays_o.dtype, device=dev)
    else:
        noises = torch.zeros(n_alive, dtype=rays_o.dtype, device=dev)
    ts = [pig_cnt, pig_bgn, pig_idx, p_def, p_ori, F_IP, dF_IP, bbmin, bbmax, res, cut_bounds, rays_alive, rays_t, density_bitfield, near, far]
    require_gpu(*ts)
    p_def, p_ori, F_IP, dF_IP, bbmin, bbmax, cut_bounds, rays_t, near, far = (_f32(t) for t in (p_def, p_ori, F_IP, dF_IP, bbmin, bbmax, cut_bounds,
                                                                                              rays_t, near, far))
    check(lib().pn_march_rays_quadratic_bending(
        ptr(pig_cnt), ptr(pig_bgn), ptr(pig_idx), int(n_vtx), n_grid, ptr(p_def), ptr(p_ori), ptr(F_IP), ptr(dF_IP), int(max_iter_num), ptr(bbmin),
        ptr(bbmax), float(hgs), ptr(res), int(num_seek_IP), float(IP_dx), int(bool(cut)), ptr(cut_bounds), n_alive, n_step, ptr(rays_alive),
        ptr(rays_t), ptr(rays_o), ptr(rays_d), float(bound), float(dt_gamma), int(max_steps), int(C), int(H), ptr(density_bitfield), ptr(near),
        ptr(far), ptr(xyzs), ptr(dirs), ptr(deltas), ptr(noises), None, stream_ptr()), "march_rays_quadratic_bending")
    return xyzs, dirs, deltas


def march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, density_bitfield, C, H, near, far, align=-1, perturb=False, dt_gamma=0,
               max_steps=1024):
    """raymarching/raymarching.py:306-358: the undeformed march (static inference).  Returns zero-initialised
    xyzs [M,3], dirs [M,3], deltas [M,2]."""
    if not rays_o.is_cuda:
        rays_o = rays_o.cuda()
    if not rays_d.is_cuda:
        rays_d = rays_d.cuda()
    rays_o = _f32(rays_o).view(-1, 3)
    rays_d = _f32(rays_d).view(-1, 3)
    n_alive, n_step = int(n_alive), int(n_step)
    M = n_alive * n_step
    if align > 0:
        M += align - (M % align)
    dev = rays_o.device
    xyzs = torch.zeros(M, 3, dtype=rays_o.dtype, device=dev)
    dirs = torch.zeros(M, 3, dtype=rays_o.dtype, device=dev)
    deltas = torch.zeros(M, 2, dtype=rays_o.dtype, device=dev)
    noises = torch.rand(n_alive, dtype=rays_o.dtype, device=dev) if perturb else torch.zeros(n_alive, dtype=rays_o.dtype, device=dev)
    require_gpu(rays_alive, rays_t, density_bitfield, near, far)
    rays_t, near, far = _f32(rays_t), _f32(near), _f32(far)
    check(lib().pn_march_rays(n_alive, n_step, ptr(rays_alive), ptr(rays_t), ptr(rays_o), ptr(rays_d), float(bound), float(dt_gamma), int(max_steps),
                              int(C), int(H), ptr(density_bitfield), ptr(near), ptr(far), ptr(xyzs), ptr(dirs), ptr(deltas), ptr(noises),
                              stream_ptr()), "march_rays")
    return xyzs, dirs, deltas


def morton3D(coords):
    """raymarching.py:85-106: coords [N,3] int32 in [0,128) -> indices [N] int32."""
    if not coords.is_cuda:
        coords = coords.cuda()
    coords = coords.int().contiguous()
    N = coords.shape[0]
    indices = torch.empty(N, dtype=torch.int32, device=coords.device)
    check(lib().pn_morton3D(ptr(coords), N, ptr(indices), stream_ptr()), "morton3D")
    return indices


def morton3D_invert(indices):
    """raymarching.py:108-128: indices [N] int32 -> coords [N,3] int32."""
    if not indices.is_cuda:
        indices = indices.cuda()
    indices = indices.int().contiguous()
    N = indices.shape[0]
    coords = torch.empty(N, 3, dtype=torch.int32, device=indices.device)
    check(lib().pn_morton3D_invert(ptr(indices), N, ptr(coords), stream_ptr()), "morton3D_invert")
    return coords


def packbits(grid, thresh, bitfield=None):
    """raymarching.py:131-157: grid float [C, H^3] -> bitfield uint8 [C*H^3/8], bit i of byte n = grid[8n+i] > thresh."""
    if not grid.is_cuda:
        grid = grid.cuda()
    grid = _f32(grid)
    N = grid.shape[0] * grid.shape[1] // 8
    if bitfield is None:
        bitfield = torch.empty(N, dtype=torch.uint8, device=grid.device)
    require_gpu(bitfield)
    check(lib().pn_packbits(ptr(grid), N, float(thresh), ptr(bitfield), stream_ptr()), "packbits")
    return bitfield


def composite_rays(n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image, T_thresh=1e-2):
    """raymarching/raymarching.py:362-384.  In place on rays_alive, rays_t, weights_sum, depth, image."""
    require_gpu(rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image)
    sigmas, rgbs, deltas = _f32(sigmas), _f32(rgbs), _f32(deltas)
    check(lib().pn_composite_rays(int(n_alive), int(n_step), float(T_thresh), ptr(rays_alive), ptr(rays_t), ptr(sigmas), ptr(rgbs), ptr(deltas),
                                  ptr(weights_sum), ptr(depth), ptr(image), stream_ptr()), "composite_rays")
    return tuple()


def compact_rays(rays_alive):
    """Stable filter ``rays_alive[rays_alive >= 0]`` (nerf/renderer.py:887) as a HIP kernel; returns the new int32 tensor."""
    require_gpu(rays_alive)
    n = rays_alive.shape[0]
    out = torch.empty_like(rays_alive)
    n_out = torch.zeros(1, dtype=torch.int32, device=rays_alive.device)
    scratch = torch.empty(int(lib().pn_compact_scratch_ints(n)), dtype=torch.int32, device=rays_alive.device)
    check(lib().pn_compact_rays(ptr(rays_alive), n, ptr(out), ptr(n_out), ptr(scratch), stream_ptr()), "compact_rays")
    return out[: int(n_out.item())]  # the .item() is the same D2H sync the reference's boolean mask implies


# ----------------------------------------------------------------------------- training ops
class _march_rays_train(torch.autograd.Function):
    """raymarching/raymarching.py:163-236 (forward only, like the reference).  Ray rows keep ray order and point ranges are the
    exclusive prefix sum of the per-ray counts (the reference's atomicAdd order is a race; include/pienerf_hip.h)."""

    @staticmethod
    def forward(ctx, rays_o, rays_d, bound, density_bitfield, C, H, nears, fars, step_counter=None, mean_count=-1, perturb=False, align=-1,
                force_all_rays=False, dt_gamma=0, max_steps=1024):
        if not rays_o.is_cuda:
            rays_o = rays_o.cuda()
        if not rays_d.is_cuda:
            rays_d = rays_d.cuda()
        if not density_bitfield.is_cuda:
            density_bitfield = density_bitfield.cuda()
        rays_o = _f32(rays_o).view(-1, 3)
        rays_d = _f32(rays_d).view(-1, 3)
        density_bitfield = density_bitfield.contiguous()
        nears, fars = _f32(nears), _f32(fars)
        require_gpu(nears, fars, step_counter)
        N = rays_o.shape[0]
        M = N * int(max_steps)
        if not force_all_rays and mean_count > 0:
            if align > 0:
                mean_count += align - mean_count % align
            M = int(mean_count)
        dev = rays_o.device
        xyzs = torch.zeros(M, 3, dtype=rays_o.dtype, device=dev)
        dirs = torch.zeros(M, 3, dtype=rays_o.dtype, device=dev)
        deltas = torch.zeros(M, 2, dtype=rays_o.dtype, device=dev)
        rays = torch.empty(N, 3, dtype=torch.int32, device=dev)
        if step_counter is None:
            step_counter = torch.zeros(2, dtype=torch.int32, device=dev)
        noises = torch.rand(N, dtype=rays_o.dtype, device=dev) if perturb else torch.zeros(N, dtype=rays_o.dtype, device=dev)
        check(lib().pn_march_rays_train(ptr(rays_o), ptr(rays_d), ptr(density_bitfield), float(bound), float(dt_gamma), int(max_steps), N, int(C), int(H), M,
                                        ptr(nears), ptr(fars), ptr(xyzs), ptr(dirs), ptr(deltas), ptr(rays), ptr(step_counter), ptr(noises), stream_ptr()),
              "march_rays_train")
        if force_all_rays or mean_count <= 0:
            m = int(step_counter[0].item())  # D2H copy, as in the reference (raymarching.py:224-231)
            if align > 0:
                m += align - m % align
            xyzs, dirs, deltas = xyzs[:m], dirs[:m], deltas[:m]
        return xyzs, dirs, deltas, rays


march_rays_train = _march_rays_train.apply


class _composite_rays_train(torch.autograd.Function):
    """raymarching/raymarching.py:241-289: weights_sum [N], depth [N], image [N,3]; backward to sigmas and rgbs (grad_depth is not
    propagated, as in the reference)."""

    @staticmethod
    def forward(ctx, sigmas, rgbs, deltas, rays, T_thresh=1e-4):
        sigmas, rgbs, deltas = _f32(sigmas), _f32(rgbs), _f32(deltas)
        rays = rays.contiguous()
        require_gpu(sigmas, rgbs, deltas, rays)
        M, N = sigmas.shape[0], rays.shape[0]
        weights_sum = torch.empty(N, dtype=sigmas.dtype, device=sigmas.device)
        depth = torch.empty(N, dtype=sigmas.dtype, device=sigmas.device)
        image = torch.empty(N, 3, dtype=sigmas.dtype, device=sigmas.device)
        check(lib().pn_composite_rays_train_forward(ptr(sigmas), ptr(rgbs), ptr(deltas), ptr(rays), M, N, float(T_thresh), ptr(weights_sum), ptr(depth),
                                                    ptr(image), stream_ptr()), "composite_rays_train_forward")
        ctx.save_for_backward(sigmas, rgbs, deltas, rays, weights_sum, depth, image)
        ctx.dims = [M, N, T_thresh]
        return weights_sum, depth, image

    @staticmethod
    def backward(ctx, grad_weights_sum, grad_depth, grad_image):
        grad_weights_sum = _f32(grad_weights_sum)
        grad_image = _f32(grad_image)
        sigmas, rgbs, deltas, rays, weights_sum, depth, image = ctx.saved_tensors
        M, N, T_thresh = ctx.dims
        grad_sigmas = torch.zeros_like(sigmas)
        grad_rgbs = torch.zeros_like(rgbs)
        check(lib().pn_composite_rays_train_backward(ptr(grad_weights_sum), ptr(grad_image), ptr(sigmas), ptr(rgbs), ptr(deltas), ptr(rays), ptr(weights_sum),
                                                     ptr(image), M, N, float(T_thresh), ptr(grad_sigmas), ptr(grad_rgbs), stream_ptr()),
              "composite_rays_train_backward")
        return grad_sigmas, grad_rgbs, None, None, None


composite_rays_train = _composite_rays_train.apply
