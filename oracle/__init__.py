"""ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY: RENDER HALF PINNED AGAINST THE REFERENCE'S OWN KERNELS, SIMULATOR HALF UNPINNED (see below).

CPU restatement of the PIE-NeRF simulate-and-render hot path.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package; nothing under ``pienerf_amd/`` does.

Pinning: the reference (FYTalon/pienerf) ships no tests, golden vectors or fixtures.
  * Host-side pieces that import or can be extracted from its source in the build container were RUN there and their outputs committed as
    fixtures (tests/golden/make_golden_ref.py -> opts_*.json, ref_kat.npz, ref_bindings.json: get_opts, trunc_exp forward/backward, get_rays,
    OrbitCamera, nerf_matrix_to_ngp, the colour-space helpers, the GridEncoder table layout, the pybind signatures); tests/test_golden_ref.py
    holds this package against them.
  * Its three CUDA extensions (raymarching, gridencoder, shencoder) are compiled for gfx950 from the sources where they lie by
    ``oracle/ref_build.py`` -> ``oracle/_ref/*.so`` (test-only torch extensions, git-ignored, travel to the GPU box).  tests/test_gpu_ref.py
    runs them beside the HIP path, which the other GPU tests hold against THIS restatement on the same inputs: march / static march / training
    march / near-far / morton / packbits / composite bit for bit with the no-contraction build, encoders bit-identical to the contracting build.
  * The Warp simulator kernels (warp-lang is absent: SURVEY.md §8c) can neither be compiled nor imported: for R1-R6 the status is
    "parity unpinned" — pinned by independent-maths checks in ``tests/test_oracle_sim.py`` instead (numpy SVD polar factor, partition of unity,
    the literal (30 n_k)^2 matrices, a numpy transcription of stepforward).

Layout:
  render_oracle.cpp  R7-R16  (march w/ inverse-GMLS warp, composite, compaction,
                              hash grid, SH, MLP, get_rays, rund_cuda loop)
  sim_oracle.cpp     R1-R6   (update_F, calc_elastic + svd3, collect_rhs, step)
  sim_init.py        R18     (grid/topology construction, Q-GMLS shape functions,
                              system/mass matrices) — numpy/torch-CPU
This module: ctypes loader + numpy-level wrappers.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None


def build(force=False):
    """Compile liboracle.so with the committed Makefile (g++)."""
    srcs = [os.path.join(_HERE, f) for f in ("render_oracle.cpp", "sim_oracle.cpp", "grid_nd_oracle.cpp", "Makefile")]
    stale = (not os.path.exists(_LIB_PATH)) or any(os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs)
    if force or stale:
        subprocess.run(["make", "-C", _HERE, "-B"], check=True, stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_num_threads.restype = C.c_int
    return _lib


def _p(a, ty):
    return a.ctypes.data_as(C.POINTER(ty))


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


F, I, D, U8 = C.c_float, C.c_int, C.c_double, C.c_uint8


def num_threads():
    return lib().orc_num_threads()


# ----------------------------------------------------------------------------- render
def near_far_from_aabb(rays_o, rays_d, aabb, min_near=0.2):
    rays_o, rays_d, aabb = _f32(rays_o).reshape(-1, 3), _f32(rays_d).reshape(-1, 3), _f32(aabb)
    N = rays_o.shape[0]
    nears, fars = np.empty(N, np.float32), np.empty(N, np.float32)
    lib().orc_near_far_from_aabb(_p(rays_o, F), _p(rays_d, F), _p(aabb, F), C.c_uint32(N), F(min_near), _p(nears, F), _p(fars, F))
    return nears, fars


def sph_from_ray(rays_o, rays_d, radius):
    rays_o, rays_d = _f32(rays_o).reshape(-1, 3), _f32(rays_d).reshape(-1, 3)
    N = rays_o.shape[0]
    coords = np.empty((N, 2), np.float32)
    lib().orc_sph_from_ray(_p(rays_o, F), _p(rays_d, F), F(radius), C.c_uint32(N), _p(coords, F))
    return coords


def get_pnts_in_grids(n_vtx, n_grid, pnts, bbmin, bbmax, hgs, resolution):
    pnts, bbmin, resolution = _f32(pnts), _f32(bbmin), _i32(resolution)
    cnt, bgn, idx = np.zeros(n_grid, np.int32), np.zeros(n_grid, np.int32), np.zeros(n_vtx, np.int32)
    bad = lib().orc_pnts_in_grids(I(n_vtx), I(n_grid), _p(pnts, F), _p(bbmin, F), F(hgs), _p(resolution, I), _p(cnt, I), _p(bgn, I), _p(idx, I))
    assert bad == 0, f"{bad} points fell outside the spatial-hash grid"
    return cnt, bgn, idx


def march_rays_quadratic_bending(pig_cnt, pig_bgn, pig_idx, n_vtx, n_grid, p_def, p_ori, F_IP, dF_IP, max_iter_num, bbmin, bbmax, hgs, res,
                                 num_seek_IP, IP_dx, cut, cut_bounds, n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound,
                                 density_bitfield, Cc, H, near, far, align=-1, perturb=False, dt_gamma=0, max_steps=1024, noises=None):
    """Same argument list as raymarching/raymarching.py:390-402 (+ optional explicit noises)."""
    M = n_alive * n_step
    if align > 0:
        M += align - (M % align)
    xyzs, dirs, deltas = np.zeros((M, 3), np.float32), np.zeros((M, 3), np.float32), np.zeros((M, 2), np.float32)
    if noises is None:
        noises = np.zeros(n_alive, np.float32)
    a = dict(pig_cnt=_i32(pig_cnt), pig_bgn=_i32(pig_bgn), pig_idx=_i32(pig_idx), p_def=_f32(p_def), p_ori=_f32(p_ori), F=_f32(F_IP),
             dF=_f32(dF_IP), bbmin=_f32(bbmin), bbmax=_f32(bbmax), res=_i32(res), cb=_f32(cut_bounds), alive=_i32(rays_alive),
             t=_f32(rays_t), o=_f32(rays_o).reshape(-1, 3), d=_f32(rays_d).reshape(-1, 3), grid=np.ascontiguousarray(density_bitfield, np.uint8),
             near=_f32(near), far=_f32(far), noises=_f32(noises))
    oob = lib().orc_march_rays_quadratic_bending(
        _p(a["pig_cnt"], I), _p(a["pig_bgn"], I), _p(a["pig_idx"], I), I(n_vtx), I(n_grid), _p(a["p_def"], F), _p(a["p_ori"], F), _p(a["F"], F),
        _p(a["dF"], F), I(max_iter_num), _p(a["bbmin"], F), _p(a["bbmax"], F), F(hgs), _p(a["res"], I), I(num_seek_IP), F(IP_dx), I(int(cut)),
        _p(a["cb"], F), C.c_uint32(n_alive), C.c_uint32(n_step), _p(a["alive"], I), _p(a["t"], F), _p(a["o"], F), _p(a["d"], F), F(bound),
        F(dt_gamma), C.c_uint32(max_steps), C.c_uint32(Cc), C.c_uint32(H), _p(a["grid"], U8), _p(a["near"], F), _p(a["far"], F), _p(xyzs, F),
        _p(dirs, F), _p(deltas, F), _p(a["noises"], F))
    march_rays_quadratic_bending.last_oob = bool(oob)
    return xyzs, dirs, deltas


def warp_point(x, p_ori, p_def, F9, dF27, max_iter_num, IP_dx):
    """Per-IP Newton inverse warp (raymarching.cu:1262-1324). Returns (rest point [3], rejected flag)."""
    x, p_ori, p_def, F9, dF27 = _f32(x), _f32(p_ori), _f32(p_def), _f32(F9), _f32(dF27)
    out = np.empty(3, np.float32)
    lib().orc_warp_point.restype = I
    rej = lib().orc_warp_point(_p(x, F), _p(p_ori, F), _p(p_def, F), _p(F9, F), _p(dF27, F), I(max_iter_num), F(IP_dx), _p(out, F))
    return out, bool(rej)


def march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, density_bitfield, Cc, H, near, far, align=-1, noises=None,
               dt_gamma=0.0, max_steps=1024):
    """raymarching.march_rays (raymarching.py:306-358 / raymarching.cu:703-824).  Returns zero-initialised xyzs, dirs, deltas."""
    rays_o, rays_d = _f32(rays_o).reshape(-1, 3), _f32(rays_d).reshape(-1, 3)
    M = int(n_alive) * int(n_step)
    if align > 0:
        M += align - (M % align)
    xyzs, dirs, deltas = np.zeros((M, 3), np.float32), np.zeros((M, 3), np.float32), np.zeros((M, 2), np.float32)
    noises = np.zeros(int(n_alive), np.float32) if noises is None else _f32(noises)
    alive, rt, far = _i32(rays_alive), _f32(rays_t), _f32(far)
    grid = np.ascontiguousarray(density_bitfield, np.uint8)
    lib().orc_march_rays(C.c_uint32(int(n_alive)), C.c_uint32(int(n_step)), _p(alive, I), _p(rt, F), _p(rays_o, F), _p(rays_d, F), F(bound),
                         F(dt_gamma), C.c_uint32(int(max_steps)), C.c_uint32(int(Cc)), C.c_uint32(int(H)), _p(grid, U8), _p(far, F), _p(xyzs, F), _p(dirs, F), _p(deltas, F), _p(noises, F))
    return xyzs, dirs, deltas


def packbits(grid, thresh):
    g = _f32(grid).reshape(-1)
    out = np.empty(g.size // 8, np.uint8)
    lib().orc_packbits(_p(g, F), C.c_uint32(out.size), F(thresh), _p(out, U8))
    return out


def morton3D(coords):
    c = _i32(coords).reshape(-1, 3)
    out = np.empty(c.shape[0], np.int32)
    lib().orc_morton3D(_p(c, I), C.c_uint32(c.shape[0]), _p(out, I))
    return out


def morton3D_invert(indices):
    i = _i32(indices).reshape(-1)
    out = np.empty((i.size, 3), np.int32)
    lib().orc_morton3D_invert(_p(i, I), C.c_uint32(i.size), _p(out, I))
    return out


def composite_rays(n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image, T_thresh=1e-2):
    """In place on rays_alive, rays_t, weights_sum, depth, image (all must be contiguous numpy arrays of the right dtype)."""
    for a, ty in ((rays_alive, np.int32), (rays_t, np.float32), (weights_sum, np.float32), (depth, np.float32), (image, np.float32)):
        assert a.dtype == ty and a.flags["C_CONTIGUOUS"]
    s, r, d = _f32(sigmas), _f32(rgbs), _f32(deltas)
    lib().orc_composite_rays(C.c_uint32(n_alive), C.c_uint32(n_step), F(T_thresh), _p(rays_alive, I), _p(rays_t, F), _p(s, F), _p(r, F), _p(d, F),
                             _p(weights_sum, F), _p(depth, F), _p(image, F))


def compact_rays(rays_alive):
    a = _i32(rays_alive)
    out = np.empty_like(a)
    lib().orc_compact_rays.restype = I
    m = lib().orc_compact_rays(_p(a, I), I(a.shape[0]), _p(out, I))
    return out[:m].copy()


def grid_encode_forward(inputs, embeddings, offsets, per_level_scale, base_resolution, gridtype=0, align_corners=False, interpolation=0):
    """Returns [B, L*C] like gridencoder/grid.py:24-63 (kernel output [L,B,C] permuted)."""
    inputs, embeddings, offsets = _f32(inputs).reshape(-1, 3), _f32(embeddings), _i32(offsets)
    B, L, Cf = inputs.shape[0], offsets.shape[0] - 1, embeddings.shape[1]
    S = np.float32(np.log2(per_level_scale))
    out = np.empty((L, B, Cf), np.float32)
    lib().orc_grid_encode_forward(_p(inputs, F), _p(embeddings, F), _p(offsets, I), _p(out, F), C.c_uint32(B), C.c_uint32(3), C.c_uint32(Cf),
                                  C.c_uint32(L), F(S), C.c_uint32(base_resolution), C.c_uint32(gridtype), I(int(align_corners)),
                                  C.c_uint32(interpolation))
    return np.ascontiguousarray(out.transpose(1, 0, 2).reshape(B, L * Cf))


def grid_nd_forward(inputs, embeddings, offsets, per_level_scale, base_resolution, gridtype=0, align_corners=False, interpolation=0, dy_dx=False):
    """kernel_grid<float, D, C> for D = inputs.shape[1] in 2..5 (gridencoder.cu:87-245, grid_nd_oracle.cpp): [B, L*C] (and dy_dx [B, L*D*C])."""
    inputs, embeddings, offsets = _f32(inputs), _f32(embeddings), _i32(offsets)
    B, Dd = inputs.shape
    L, Cf = offsets.shape[0] - 1, embeddings.shape[1]
    out = np.empty((L, B, Cf), np.float32)
    dd = np.empty((B, L * Dd * Cf), np.float32) if dy_dx else None
    lib().orc_grid_nd_forward(_p(inputs, F), _p(embeddings, F), _p(offsets, I), _p(out, F), _p(dd, F) if dy_dx else None, C.c_uint32(B), C.c_uint32(Dd),
                              C.c_uint32(Cf), C.c_uint32(L), F(np.float32(np.log2(per_level_scale))), C.c_uint32(base_resolution), C.c_uint32(gridtype),
                              I(int(align_corners)), C.c_uint32(interpolation))
    y = np.ascontiguousarray(out.transpose(1, 0, 2).reshape(B, L * Cf))
    return (y, dd) if dy_dx else y


def grid_nd_backward(grad, inputs, embeddings_shape, offsets, per_level_scale, base_resolution, dy_dx=None, gridtype=0, align_corners=False, interpolation=0):
    """kernel_grid_backward + kernel_input_backward for D in 2..5 (gridencoder.cu:248-369): grad [B, L*C] -> (grad_inputs [B, D] or None, grad_embeddings)."""
    inputs, offsets = _f32(inputs), _i32(offsets)
    B, Dd = inputs.shape
    L, Cf = offsets.shape[0] - 1, embeddings_shape[1]
    g = np.ascontiguousarray(_f32(grad).reshape(B, L, Cf).transpose(1, 0, 2))
    ge = np.zeros(tuple(embeddings_shape), np.float32)
    gi = np.zeros((B, Dd), np.float32) if dy_dx is not None else None
    dd = _f32(dy_dx) if dy_dx is not None else None
    lib().orc_grid_nd_backward(_p(g, F), _p(inputs, F), _p(offsets, I), _p(ge, F), C.c_uint32(B), C.c_uint32(Dd), C.c_uint32(Cf), C.c_uint32(L),
                               F(np.float32(np.log2(per_level_scale))), C.c_uint32(base_resolution), _p(dd, F) if dd is not None else None,
                               _p(gi, F) if gi is not None else None, C.c_uint32(gridtype), I(int(align_corners)), C.c_uint32(interpolation))
    return gi, ge


def grid_nd_grad_tv(inputs, embeddings, offsets, weight, per_level_scale, base_resolution, gridtype=0, align_corners=False):
    """kernel_grad_tv<float, D, C> for D in 2..5 (gridencoder.cu:506-611): the gradient it adds to a zero tensor."""
    inputs, embeddings, offsets = _f32(inputs), _f32(embeddings), _i32(offsets)
    B, Dd = inputs.shape
    L, Cf = offsets.shape[0] - 1, embeddings.shape[1]
    g = np.zeros_like(embeddings)
    lib().orc_grid_nd_grad_tv(_p(inputs, F), _p(embeddings, F), _p(g, F), _p(offsets, I), F(weight), C.c_uint32(B), C.c_uint32(Dd), C.c_uint32(Cf), C.c_uint32(L),
                              F(np.float32(np.log2(per_level_scale))), C.c_uint32(base_resolution), C.c_uint32(gridtype), I(int(align_corners)))
    return g


def grid_level_params(L, per_level_scale, base_resolution):
    scales, res = np.empty(L, np.float32), np.empty(L, np.uint32)
    lib().orc_grid_level_params(C.c_uint32(L), F(np.float32(np.log2(per_level_scale))), C.c_uint32(base_resolution), _p(scales, F),
                                _p(res, C.c_uint32))
    return scales, res


def sh_encode_forward(inputs, degree=4):
    inputs = _f32(inputs).reshape(-1, 3)
    out = np.empty((inputs.shape[0], degree * degree), np.float32)
    lib().orc_sh_encode_forward(_p(inputs, F), _p(out, F), C.c_uint32(inputs.shape[0]), C.c_uint32(3), C.c_uint32(degree))
    return out


def nerf_forward(xyzs, dirs, ckpt, bound):
    """ckpt: dict with embeddings, offsets, per_level_scale, base_resolution, W0..W4 (see pienerf_amd.scene)."""
    xyzs, dirs = _f32(xyzs).reshape(-1, 3), _f32(dirs).reshape(-1, 3)
    M = xyzs.shape[0]
    emb, off = _f32(ckpt["embeddings"]), _i32(ckpt["offsets"])
    W = [_f32(ckpt[f"W{i}"]) for i in range(5)]
    sig, rgb = np.empty(M, np.float32), np.empty((M, 3), np.float32)
    lib().orc_nerf_forward(_p(xyzs, F), _p(dirs, F), C.c_uint32(M), F(bound), _p(emb, F), _p(off, I), C.c_uint32(off.shape[0] - 1),
                           C.c_uint32(emb.shape[1]), F(np.float32(np.log2(ckpt["per_level_scale"]))), C.c_uint32(ckpt["base_resolution"]),
                           _p(W[0], F), _p(W[1], F), _p(W[2], F), _p(W[3], F), _p(W[4], F), _p(sig, F), _p(rgb, F))
    return sig, rgb


class half_precision:
    """Context manager: inside it ``nerf_forward`` and ``render_deformed`` restate the network as the reference runs it under
    ``torch.cuda.amp.autocast`` with fp16 (trainer.py:561 with Trainer(fp16=True)): half hash tables with kernel_grid<at::Half>'s half
    accumulation (gridencoder/grid.py:43-44, gridencoder.cu:184), half nn.Linear layers, half sigmoid (render_oracle.cpp: nerf_one)."""

    def __init__(self, on=True):
        self.on = bool(on)

    def __enter__(self):
        lib().orc_set_half.restype = I
        self.prev = lib().orc_set_half(I(int(self.on)))
        return self

    def __exit__(self, *a):
        lib().orc_set_half(I(self.prev))
        return False


def hround(a):
    """a rounded to the nearest fp16 value (ties to even), as float32 — the oracle's software rounding."""
    a = _f32(a)
    out = np.empty_like(a)
    lib().orc_hround(_p(a.reshape(-1), F), _p(out.reshape(-1), F), C.c_uint32(a.size))
    return out


def grid_encode_forward_half(inputs, embeddings, offsets, per_level_scale, base_resolution, gridtype=0, align_corners=False, interpolation=0):
    """kernel_grid<at::Half> (the autocast branch of gridencoder/grid.py:24-63): [B, L*C] half values as float32."""
    inputs, embeddings, offsets = _f32(inputs).reshape(-1, 3), _f32(embeddings), _i32(offsets)
    B, L, Cf = inputs.shape[0], offsets.shape[0] - 1, embeddings.shape[1]
    out = np.empty((L, B, Cf), np.float32)
    lib().orc_grid_encode_forward_half(_p(inputs, F), _p(embeddings, F), _p(offsets, I), _p(out, F), C.c_uint32(B), C.c_uint32(Cf), C.c_uint32(L),
                                       F(np.float32(np.log2(per_level_scale))), C.c_uint32(base_resolution), C.c_uint32(gridtype),
                                       I(int(align_corners)), C.c_uint32(interpolation))
    return np.ascontiguousarray(out.transpose(1, 0, 2).reshape(B, L * Cf))


def get_rays(pose, intrinsics, H, W):
    pose = _f32(pose).reshape(4, 4)
    fx, fy, cx, cy = [float(np.float32(v)) for v in intrinsics]
    o, d = np.empty((H * W, 3), np.float32), np.empty((H * W, 3), np.float32)
    lib().orc_get_rays(_p(pose, F), F(fx), F(fy), F(cx), F(cy), I(H), I(W), _p(o, F), _p(d, F))
    return o, d


def render_bbox(p_def, hgs, cut=False, bound=1.0):
    """nerf/renderer.py:782-791 in fp32: bbox of deformed IPs +-1e-3, resolution = ceil(extent / hgs)."""
    p_def = _f32(p_def)
    if cut:
        bmin = -np.float32(bound) * np.ones(3, np.float32)
        bmax = np.float32(bound) * np.ones(3, np.float32)
    else:
        bmin, bmax = p_def.min(axis=0), p_def.max(axis=0)
    marg = np.float32(1e-3)
    bbmin = (bmin - marg * np.ones(3, np.float32)).astype(np.float32)
    bbmax = (bmax + marg * np.ones(3, np.float32)).astype(np.float32)
    resolution = np.ceil((bbmax - bbmin) / np.float32(hgs)).astype(np.int32)
    return bbmin, bbmax, resolution


def render_deformed(rays_o, rays_d, ip_state, ckpt, opt, bg_color=1.0, bbox=None):
    """NeRFRenderer.rund_cuda (nerf/renderer.py:755-907), perturb=False.

    ip_state: dict(p_def, p_ori, F, dF, IP_dx); ckpt: dict(embeddings, offsets, per_level_scale, base_resolution,
    W0..W4, density_bitfield, cascade, grid_size, bound, min_near, density_scale); opt: dict with the reference's option
    names (max_iter_num, hash_grid_size, num_seek_IP, cut, cut_bounds, dt_gamma, max_steps, T_thresh).
    """
    rays_o, rays_d = _f32(rays_o).reshape(-1, 3), _f32(rays_d).reshape(-1, 3)
    N = rays_o.shape[0]
    p_def, p_ori, Fm, dFm = _f32(ip_state["p_def"]), _f32(ip_state["p_ori"]), _f32(ip_state["F"]), _f32(ip_state["dF"])
    n_vtx = p_def.shape[0]
    hgs = float(np.float32(opt["hash_grid_size"]))
    cut = bool(opt.get("cut", False))
    bound = float(ckpt["bound"])
    bbmin, bbmax, res = bbox if bbox is not None else render_bbox(p_def, hgs, cut, bound)
    bbmin, bbmax, res = _f32(bbmin), _f32(bbmax), _i32(res)
    cb = _f32(opt.get("cut_bounds", [0, 0, 0, 0, 0, 0]))
    emb, off = _f32(ckpt["embeddings"]), _i32(ckpt["offsets"])
    Wt = [_f32(ckpt[f"W{i}"]) for i in range(5)]
    bits = np.ascontiguousarray(ckpt["density_bitfield"], np.uint8)
    image, depth, depth0, ws = (np.empty((N, 3), np.float32), np.empty(N, np.float32), np.empty(N, np.float32), np.empty(N, np.float32))
    stats = np.zeros(3, np.int64)
    lib().orc_render_deformed(
        _p(rays_o, F), _p(rays_d, F), C.c_uint32(N), _p(p_def, F), _p(p_ori, F), _p(Fm, F), _p(dFm, F), I(n_vtx), _p(bbmin, F), _p(bbmax, F),
        _p(res, I), F(hgs), I(int(opt["max_iter_num"])), I(int(opt["num_seek_IP"])), F(float(np.float32(ip_state["IP_dx"]))), I(int(cut)), _p(cb, F),
        F(bound), F(float(ckpt.get("min_near", 0.2))), F(float(opt.get("dt_gamma", 0.0))), C.c_uint32(int(opt.get("max_steps", 1024))),
        F(float(opt.get("T_thresh", 1e-2))), C.c_uint32(int(ckpt["cascade"])), C.c_uint32(int(ckpt["grid_size"])), _p(bits, U8),
        F(float(ckpt.get("density_scale", 1.0))), F(float(bg_color)), _p(emb, F), _p(off, I), C.c_uint32(off.shape[0] - 1),
        C.c_uint32(emb.shape[1]), F(np.float32(np.log2(ckpt["per_level_scale"]))), C.c_uint32(int(ckpt["base_resolution"])), _p(Wt[0], F),
        _p(Wt[1], F), _p(Wt[2], F), _p(Wt[3], F), _p(Wt[4], F), _p(image, F), _p(depth, F), _p(depth0, F), _p(ws, F),
        _p(stats, C.c_int64))
    return dict(image=image, depth=depth, depth_0=depth0, weights_sum=ws, trips=int(stats[0]), samples=int(stats[1]), slots=int(stats[2]))


# ----------------------------------------------------------------------------- sim
def render_static(rays_o, rays_d, ckpt, opt, bg_color=1.0):
    """NeRFRenderer.run_cuda, inference branch (nerf/renderer.py:267-387), perturb=False, op by op on the CPU."""
    rays_o, rays_d = _f32(rays_o).reshape(-1, 3), _f32(rays_d).reshape(-1, 3)
    N = rays_o.shape[0]
    bound = float(ckpt["bound"])
    aabb = np.array([-bound] * 3 + [bound] * 3, np.float32)                       # renderer.py:33-36
    nears, fars = near_far_from_aabb(rays_o, rays_d, aabb, ckpt.get("min_near", 0.2))
    ws, depth, image = np.zeros(N, np.float32), np.zeros(N, np.float32), np.zeros((N, 3), np.float32)
    alive = np.arange(N, dtype=np.int32)
    rays_t = nears.copy()
    step, trips, samples = 0, 0, 0
    max_steps = int(opt["max_steps"])
    while step < max_steps:
        n_alive = alive.shape[0]
        if n_alive <= 0:
            break
        n_step = max(min(N // n_alive, 8), 1)
        xyzs, dirs, deltas = march_rays(n_alive, n_step, alive, rays_t, rays_o, rays_d, bound, ckpt["density_bitfield"], ckpt["cascade"],
                                        ckpt["grid_size"], nears, fars, 128, None, float(opt["dt_gamma"]), max_steps)
        live = deltas[:, 0] != 0
        sig, rgb = np.zeros(len(xyzs), np.float32), np.zeros((len(xyzs), 3), np.float32)
        if live.any():                                                              # slots with delta 0 are never read by composite
            sig[live], rgb[live] = nerf_forward(xyzs[live], dirs[live], ckpt, bound)
        sig = np.float32(ckpt.get("density_scale", 1.0)) * sig
        composite_rays(n_alive, n_step, alive, rays_t, sig, rgb, deltas, ws, depth, image, float(opt["T_thresh"]))
        alive = compact_rays(alive)
        samples += int(live.sum())
        step += n_step
        trips += 1
    image = image + (1 - ws)[:, None] * np.float32(bg_color)
    with np.errstate(invalid="ignore", divide="ignore"):
        depth = np.maximum(depth - nears, 0) / (fars - nears)
    return dict(image=image.astype(np.float32), depth=depth.astype(np.float32), weights_sum=ws, trips=trips, samples=samples)


def svd3(Fm):
    Fm = _f64(Fm).reshape(3, 3)
    U, s, V = np.empty((3, 3)), np.empty(3), np.empty((3, 3))
    lib().orc_svd3(_p(Fm, D), _p(U, D), _p(s, D), _p(V, D))
    return U, s, V


class svd_mode:
    """Which restatement of wp.svd3 (simulator/cuda_utils.py:107) the simulator oracle uses inside the `with` block:
    svd_mode("converged") — the contract (default); svd_mode("mcadams", sweeps=8, rsqrt="exact"|"seeded", qr_eps=1e-12) — the published
    algorithm (McAdams et al., TR1690) with a fixed sweep count.  Process-global in liboracle.so; restored on exit."""

    def __init__(self, mode="converged", sweeps=8, rsqrt="exact", qr_eps=1e-12, constants="published"):
        self.args = (dict(converged=0, mcadams=1)[mode], int(sweeps), dict(exact=0, seeded=1)[rsqrt], float(qr_eps), dict(published=0, exact=1)[constants])

    def __enter__(self):
        assert lib().orc_get_svd_mode() == 0, "svd_mode blocks do not nest"
        lib().orc_set_svd(I(self.args[0]), I(self.args[1]), I(self.args[2]), D(self.args[3]), I(self.args[4]))
        return self

    def __exit__(self, *a):
        lib().orc_set_svd(I(0), I(8), I(0), D(1e-12), I(0))
        return False


def volume_invariant_project(sig):
    sig = _f64(sig)
    out = np.empty(3)
    lib().orc_volume_invariant_project(_p(sig, D), _p(out, D))
    return out


def update_F(topo, dof, Nx, dNx, ddNx):
    topo, dof, Nx, dNx, ddNx = _i32(topo), _f64(dof), _f64(Nx), _f64(dNx), _f64(ddNx)
    n = topo.shape[0]
    pos, Fm, dFm = np.empty((n, 3), np.float32), np.empty((n, 9), np.float32), np.empty((n, 27), np.float32)
    lib().orc_update_F(I(n), _p(topo, I), _p(dof, D), _p(Nx, D), _p(dNx, D), _p(ddNx, D), _p(pos, F), _p(Fm, F), _p(dFm, F))
    return pos, Fm, dFm


def calc_elastic(topo, dNx, dof):
    topo, dNx, dof = _i32(topo), _f64(dNx), _f64(dof)
    n = topo.shape[0]
    RF, VF, FF = np.empty((n, 3, 3)), np.empty((n, 3, 3)), np.empty((n, 3, 3))
    lib().orc_calc_elastic(I(n), _p(topo, I), _p(dNx, D), _p(dof, D), _p(RF, D), _p(VF, D), _p(FF, D))
    return RF, VF, FF


def collect_rhs_IP(dx, topo, mu, lam, dNx, RF, VF, n_dof10):
    topo, mu, lam, dNx, RF, VF = _i32(topo), _f64(mu), _f64(lam), _f64(dNx), _f64(RF), _f64(VF)
    rhs = np.zeros((n_dof10, 3))
    lib().orc_collect_rhs_IP(I(topo.shape[0]), D(dx), _p(topo, I), _p(mu, D), _p(lam, D), _p(dNx, D), _p(rhs, D), _p(RF, D), _p(VF, D))
    return rhs


def matvec3(A, X):
    A, X = _f64(A), _f64(X)
    n = A.shape[0]
    Y = np.empty((n, 3))
    lib().orc_matvec3(I(n), _p(A, D), _p(X.reshape(n, 3), D), _p(Y, D))
    return Y


def stepforward(st):
    """One Simulator.stepforward() on an oracle state dict (see sim_init.OracleSimulator.state()). In place on dof, dof_vel."""
    n = st["Ainv"].shape[0]
    for k in ("dof", "dof_vel"):
        assert st[k].dtype == np.float64 and st[k].flags["C_CONTIGUOUS"]
    lib().orc_stepforward(I(n), I(st["IP_kernel"].shape[0]), I(st["iters"]), D(st["dt"]), D(st["dx"]), _p(st["IP_kernel"], I), _p(st["IP_mu"], D),
                          _p(st["IP_lam"], D), _p(st["IP_dNx"], D), _p(st["Ainv"], D), _p(st["Mmat"], D), _p(st["dof_rest"], D),
                          _p(st["rhs_rest"], D), _p(st["rhs_gravity"], D), _p(st["dof_f"], D), _p(st["dof"], D), _p(st["dof_vel"], D))
