"""PINNING against the reference's OWN kernels.  `oracle/_ref/_ref_{raymarching,gridencoder,shencoder}[_fma].so` are the reference's
CUDA extensions (raymarching/src, gridencoder/src, shencoder/src) built for gfx950 from the sources where they lie by the committed
recipe `oracle/ref_build.py` (test infrastructure; see its header for exactly what is compiled and the one statement that cannot be).
Here every native function of the three modules is called TWICE with the same positional arguments — once on the reference's module,
once on this repository's drop-in backend (`shim/_raymarching.py`, `_gridencoder.py`, `_shencoder.py` -> the C ABI of libpienerf_hip.so)
— and the outputs are compared:

  * against the `-ffp-contract=off` build of the reference (one rounding per source operation — the semantics the CPU restatement and the
    bit-exact kernels are written to): integer / index outputs and the march's float outputs BIT FOR BIT;
  * against the default-contraction build (`*_fma`, the analogue of nvcc's -fmad=true): the mismatch RATE is measured, printed, written to
    `gpurun_out/ref_parity_report.json`, and bounded — contraction moves last bits and, rarely, a floor()/comparison; it must not move more.

Floating-point work (encoders, compositing) is held to the north-star tolerance (1e-4 relative; the bars in the tests are tighter).
The Warp simulator kernels (warp-lang is absent) stay unpinned."""
import json
import os
import sys

import numpy as np
import pytest
import torch

import oracle
from conftest import ROOT
from oracle import ref_build
from pienerf_amd import scene
from test_gpu_parity import DEV, T, _march_inputs

pytestmark = pytest.mark.gpu
SHIM = os.path.join(ROOT, "shim")
REPORT = {}

if not all(os.path.exists(ref_build.so_path(e, f)) for e in ref_build.EXTS for f in (False, True)):
    pytest.skip("oracle/_ref is not built (python -m oracle.ref_build, needs /root/reference)", allow_module_level=True)


@pytest.fixture(scope="module")
def mods():
    """(reference modules, fma-contracted reference modules, this repository's drop-in modules)"""
    sys.path.insert(0, SHIM)
    try:
        import _gridencoder
        import _raymarching
        import _shencoder
    finally:
        sys.path.remove(SHIM)
    ours = dict(raymarching=_raymarching, gridencoder=_gridencoder, shencoder=_shencoder)
    for k in ("_raymarching", "_gridencoder", "_shencoder"):
        sys.modules.pop(k, None)
    ref = {e: ref_build.load(e, False) for e in ref_build.EXTS}
    fma = {e: ref_build.load(e, True) for e in ref_build.EXTS}
    yield ref, fma, ours
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "ref_parity_report.json"), "w") as f:
        json.dump(REPORT, f, indent=1, sort_keys=True)


def bits_equal(a, b):
    a, b = a.detach().cpu().numpy(), b.detach().cpu().numpy()
    return a.shape == b.shape and np.array_equal(a.reshape(-1).view(np.uint8), b.reshape(-1).view(np.uint8))


def mismatch(a, b):
    """fraction of elements whose bits differ, largest absolute difference"""
    a, b = a.detach().cpu().numpy(), b.detach().cpu().numpy()
    ne = a.reshape(-1).view(np.uint32) != b.reshape(-1).view(np.uint32)
    d = np.abs(a.astype(np.float64) - b.astype(np.float64))
    return float(ne.mean()), float(np.nanmax(d)) if d.size else 0.0


def rel(a, b):
    a, b = a.detach().cpu().numpy().astype(np.float64), b.detach().cpu().numpy().astype(np.float64)
    return float(np.abs(a - b).max() / max(1e-30, np.abs(b).max()))


# ------------------------------------------------------------------------------------------------ R9, morton, packbits
def test_near_far_morton_packbits_bit_exact(mods):
    ref, fma, ours = mods
    W = 96
    o, d = oracle.get_rays(scene.orbit_pose(5.0, 30.0, -20.0), scene.orbit_intrinsics(W, W, 50.0), W, W)
    N = len(o)
    aabb = T(np.array([-0.6, -0.8, -0.5, 0.7, 0.9, 0.55], np.float32))
    out = {}
    for name, m in (("ref", ref), ("fma", fma), ("ours", ours)):
        n, f = torch.empty(N, device=DEV), torch.empty(N, device=DEV)
        m["raymarching"].near_far_from_aabb(T(o), T(d), aabb, N, 0.2, n, f)
        torch.cuda.synchronize()
        out[name] = (n, f)
    assert bits_equal(out["ours"][0], out["ref"][0]) and bits_equal(out["ours"][1], out["ref"][1])
    assert (out["ref"][0] > 1e30).any() and (out["ref"][0] < 10).any()
    REPORT["near_far_vs_fma"] = dict(nears=mismatch(out["ours"][0], out["fma"][0]), fars=mismatch(out["ours"][1], out["fma"][1]))
    sph = {}
    for name, m in (("ref", ref), ("ours", ours)):   # kernel_sph_from_ray (raymarching.cu:165-202): same device libm, no contraction -> bit for bit
        c = torch.empty(N, 2, device=DEV)
        m["raymarching"].sph_from_ray(T(o), T(d), 8.0, N, c)
        sph[name] = c
    assert bits_equal(sph["ours"], sph["ref"]) and float(sph["ref"].abs().max()) <= 1.0
    assert np.abs(sph["ours"].cpu().numpy() - oracle.sph_from_ray(o, d, 8.0)).max() < 2e-6
    coords = torch.randint(0, 1024, (5000, 3), dtype=torch.int32, device=DEV)
    grid = torch.rand(128 ** 3, device=DEV) * 20
    res = {}
    for name, m in (("ref", ref), ("ours", ours)):
        idx, back = torch.empty(5000, dtype=torch.int32, device=DEV), torch.empty(5000, 3, dtype=torch.int32, device=DEV)
        bits = torch.zeros(128 ** 3 // 8, dtype=torch.uint8, device=DEV)
        m["raymarching"].morton3D(coords, 5000, idx)
        m["raymarching"].morton3D_invert(idx, 5000, back)
        m["raymarching"].packbits(grid, 128 ** 3 // 8, 10.0, bits)
        torch.cuda.synchronize()
        res[name] = (idx, back, bits)
    for a, b in zip(res["ours"], res["ref"]):
        assert torch.equal(a, b)
    assert torch.equal(res["ref"][1], coords)


# ------------------------------------------------------------------------------------------------ R7 + R7q
def _call_march(mod, m, ip, ck, alive, n_step, num_seek_IP, max_iter_num, cut, cb, dt_gamma, max_steps, noises, rays_t, bound=1.0):
    n_alive = len(alive)
    M = n_alive * n_step
    M += 128 - (M % 128)
    xyzs, dirs, deltas = torch.zeros(M, 3, device=DEV), torch.zeros(M, 3, device=DEV), torch.zeros(M, 2, device=DEV)
    keep = [T(a) for a in m["pig"]] + [T(ip[k]) for k in ("p_def", "p_ori", "F", "dF")] + [T(m["bbmin"]), T(m["bbmax"]), T(m["res"]), T(cb), T(alive),
                                                                                              T(rays_t), T(m["o"]), T(m["d"]), T(ck["density_bitfield"]),
                                                                                              T(m["nears"]), T(m["fars"]), T(noises)]
    k = keep
    mod.march_rays_quadratic_bending(k[0], k[1], k[2], len(ip["p_def"]), m["n_grid"], k[3], k[4], k[5], k[6], max_iter_num, k[7], k[8], float(m["hgs"]), k[9],
                                     num_seek_IP, float(ip["IP_dx"]), bool(cut), k[10], n_alive, n_step, k[11], k[12], k[13], k[14], bound, dt_gamma, max_steps,
                                     ck["cascade"], ck["grid_size"], k[15], k[16], k[17], xyzs, dirs, deltas, k[18])
    torch.cuda.synchronize()
    return xyzs, dirs, deltas, k[12]


@pytest.mark.parametrize("num_seek_IP,max_iter_num,n_step,noise", [(1, 1, 1, False), (3, 1, 4, True), (2, 3, 8, False), (3, 5, 8, False), (3, 1, 16, False)])
def test_march_quadratic_bending_equals_the_reference_kernel(mods, deformed_ip_state, small_opt, ckpt, num_seek_IP, max_iter_num, n_step, noise):
    """kernel_march_rays_quadratic_bending (raymarching.cu:1121-1434) itself, num_seek_IP 1 / 2 / 3 (find_closest_IP vs find_closest_IPs and every
    R7q quirk on those paths), 1 / 3 / 5 Newton iterations, rays that miss the box, explicit noises."""
    ref, fma, ours = mods
    ip, ck = deformed_ip_state, ckpt
    m = _march_inputs(ip, small_opt, ck, W=96)
    N = m["o"].shape[0]
    alive = np.nonzero(m["nears"] < 1e30)[0].astype(np.int32)
    alive = np.concatenate([alive, np.arange(0, N, 97, dtype=np.int32)])
    noises = np.random.default_rng(1).random(len(alive)).astype(np.float32) if noise else np.zeros(len(alive), np.float32)
    cb = np.zeros(6, np.float32)
    args = (m, ip, ck, alive, n_step, num_seek_IP, max_iter_num, False, cb, 0.0, 1024, noises, m["nears"])
    r = _call_march(ref["raymarching"], *args)
    g = _call_march(ours["raymarching"], *args)
    f = _call_march(fma["raymarching"], *args)
    emitted = int((r[2][:, 0] != 0).sum())
    assert emitted > 500
    for name, a, b in zip(("xyzs", "dirs", "deltas"), g, r):
        assert bits_equal(a, b), f"{name}: {mismatch(a, b)}"
    # the reference kernel never writes rays_t (the composite does); neither does the drop-in
    assert bits_equal(g[3], r[3])
    rate = {name: mismatch(a, b) for name, a, b in zip(("xyzs", "dirs", "deltas"), g, f)}
    rows = (g[0] != f[0]).any(1) | (g[2] != f[2]).any(1)
    REPORT[f"march_vs_fma[seek{num_seek_IP},iter{max_iter_num},step{n_step}]"] = dict(rate, sample_rows_differing=float(rows.float().mean()), emitted=emitted)
    print("vs default-contraction build:", REPORT[f"march_vs_fma[seek{num_seek_IP},iter{max_iter_num},step{n_step}]"])
    # contraction may move last bits of a warped position and, rarely, a sample across a cell / voxel boundary — not more
    assert rate["xyzs"][1] < 2e-2 and rate["deltas"][0] < 0.05


@pytest.mark.parametrize("background,num_seek_IP,n_step", [(False, 1, 6), (True, 1, 6), (True, 3, 2), (True, 2, 16)])
def test_march_cut_mode_equals_the_reference_kernel(mods, deformed_ip_state, small_opt, ckpt, background, num_seek_IP, n_step):
    """--cut: bbox = +-bound, cut_bounds test with the reference's own `x < cut_bounds[3]` (raymarching.cu:1195-1210), static background samples,
    dt_gamma 1/128, max_steps 300 (the trex option set)."""
    ref, fma, ours = mods
    ip, ck = deformed_ip_state, dict(ckpt)
    if background:
        rng = np.random.default_rng(5)
        blobs = np.repeat(rng.random(len(ck["density_bitfield"]) // 64) < 0.04, 64)
        ck["density_bitfield"] = ck["density_bitfield"] | np.where(blobs, 0xFF, 0).astype(np.uint8)
    W = 40
    o, d = oracle.get_rays(scene.orbit_pose(4.0, 10.0, -5.0), scene.orbit_intrinsics(W, W, 50.0), W, W)
    hgs = np.float32(small_opt["hash_grid_size"])
    bbmin, bbmax, res = oracle.render_bbox(ip["p_def"], hgs, cut=True, bound=1.0)
    n_grid = int(res.prod())
    pig = oracle.get_pnts_in_grids(len(ip["p_def"]), n_grid, ip["p_def"], bbmin, bbmax, hgs, res)
    nears, fars = oracle.near_far_from_aabb(o, d, np.concatenate([bbmin, bbmax]), 0.2)
    m = dict(o=o, d=d, hgs=hgs, bbmin=bbmin, bbmax=bbmax, res=res, n_grid=n_grid, pig=pig, nears=nears, fars=fars)
    alive = np.arange(W * W, dtype=np.int32)
    cb = np.array([-0.3, 0.9, -0.9, 0.5, -0.9, 0.9], np.float32)
    args = (m, ip, ck, alive, n_step, num_seek_IP, 1, True, cb, 1.0 / 128, 300, np.zeros(len(alive), np.float32), nears)
    r = _call_march(ref["raymarching"], *args)
    g = _call_march(ours["raymarching"], *args)
    f = _call_march(fma["raymarching"], *args)
    assert int((r[2][:, 0] != 0).sum()) > 100
    for name, a, b in zip(("xyzs", "dirs", "deltas"), g, r):
        assert bits_equal(a, b), f"{name}: {mismatch(a, b)}"
    REPORT[f"march_cut_vs_fma[bg{int(background)},seek{num_seek_IP},step{n_step}]"] = {n: mismatch(a, b) for n, a, b in zip(("xyzs", "dirs", "deltas"), g, f)}


def test_march_rays_static_equals_the_reference_kernel(mods, ckpt):
    """kernel_march_rays (raymarching.cu:703-810), the un-deformed march of run_cuda."""
    ref, fma, ours = mods
    for bound, dt_gamma, n_step, max_steps in ((1.0, 0.0, 8, 1024), (2.0, 1.0 / 128, 8, 300), (1.0, 1.0 / 64, 64, 512)):
        ck = scene.make_checkpoint(bound=bound, seed=1)
        W = 64
        o, d = oracle.get_rays(scene.orbit_pose(3.4 * bound, 25.0, -20.0), scene.orbit_intrinsics(W, W, 50.0), W, W)
        nears, fars = oracle.near_far_from_aabb(o, d, np.array([-bound] * 3 + [bound] * 3, np.float32), 0.2)
        alive = np.nonzero(nears < 1e30)[0].astype(np.int32)
        n_alive = len(alive)
        M = n_alive * n_step
        M += 128 - (M % 128)
        outs = {}
        for name, mm in (("ref", ref), ("ours", ours), ("fma", fma)):
            xyzs, dirs, deltas = torch.zeros(M, 3, device=DEV), torch.zeros(M, 3, device=DEV), torch.zeros(M, 2, device=DEV)
            keep = (T(alive), T(nears), T(o), T(d), T(ck["density_bitfield"]), T(nears), T(fars), torch.zeros(n_alive, device=DEV))
            mm["raymarching"].march_rays(n_alive, n_step, keep[0], keep[1], keep[2], keep[3], bound, dt_gamma, max_steps, ck["cascade"], ck["grid_size"], keep[4],
                                         keep[5], keep[6], xyzs, dirs, deltas, keep[7])
            torch.cuda.synchronize()
            outs[name] = (xyzs, dirs, deltas)
        assert int((outs["ref"][2][:, 0] != 0).sum()) > 300
        for a, b in zip(outs["ours"], outs["ref"]):
            assert bits_equal(a, b)
        REPORT[f"march_static_vs_fma[bound{bound},n_step{n_step}]"] = {n: mismatch(a, b) for n, a, b in zip(("xyzs", "dirs", "deltas"), outs["ours"], outs["fma"])}


# ------------------------------------------------------------------------------------------------ R13
def test_composite_rays_equals_the_reference_kernel(mods):
    ref, fma, ours = mods
    rng = np.random.default_rng(6)
    N, n_alive, n_step = 3000, 1700, 8
    alive = np.sort(rng.choice(N, n_alive, replace=False)).astype(np.int32)
    M = n_alive * n_step
    sig = (rng.random(M).astype(np.float32) * 120)
    rgb = rng.random((M, 3)).astype(np.float32)
    deltas = np.stack([np.full(M, 0.0034, np.float32), (rng.random(M) * 0.01 + 0.0034).astype(np.float32)], 1)
    for n in range(0, n_alive, 3):
        deltas[n * n_step + rng.integers(0, n_step):(n + 1) * n_step] = 0
    st = dict(t=(rng.random(N).astype(np.float32) + 3), ws=(rng.random(N).astype(np.float32) * 0.9), dep=rng.random(N).astype(np.float32),
              img=rng.random((N, 3)).astype(np.float32))
    out = {}
    for name, mm in (("ref", ref), ("ours", ours), ("fma", fma)):
        g = {k: T(v.copy()) for k, v in st.items()}
        al = T(alive.copy())
        keep = (T(sig), T(rgb), T(deltas))
        mm["raymarching"].composite_rays(n_alive, n_step, 1e-2, al, g["t"], keep[0], keep[1], keep[2], g["ws"], g["dep"], g["img"])
        torch.cuda.synchronize()
        out[name] = dict(g, alive=al)
    assert torch.equal(out["ours"]["alive"], out["ref"]["alive"]) and (out["ref"]["alive"] < 0).any() and (out["ref"]["alive"] >= 0).any()
    assert torch.equal(out["ours"]["alive"], out["fma"]["alive"])
    for k in ("t", "ws", "dep", "img"):
        assert rel(out["ours"][k], out["ref"][k]) < 2e-6, k
        REPORT[f"composite_{k}"] = dict(vs_ref=mismatch(out["ours"][k], out["ref"][k]), vs_fma=mismatch(out["ours"][k], out["fma"][k]))
    assert bits_equal(out["ours"]["t"], out["ref"]["t"])   # rays_t is a running sum of deltas: no products to contract


# ------------------------------------------------------------------------------------------------ R10 / R11
@pytest.mark.parametrize("gridtype,align,interp", [(0, False, 0), (1, False, 0), (0, True, 0), (0, False, 1)])
def test_grid_encode_forward_equals_the_reference_kernel(mods, ckpt, gridtype, align, interp):
    """kernel_grid<float,3,2> (gridencoder.cu:87-245) on the chair's 16-level table, with dy_dx.  The product states `pos = fmaf(u, scale, 0.5)` as ONE
    rounding (what a contracting compiler makes of gridencoder.cu:129) — so the tight bar is against the contracting build of the reference and the
    no-contraction build differs by what that one rounding moves (<= 1 ulp of pos at the finest level: 1.2e-4 of a weight)."""
    ref, fma, ours = mods
    rng = np.random.default_rng(2)
    B, L, C = 20000, 16, 2
    x = rng.random((B, 3)).astype(np.float32)
    x[:7] = [[0, 0, 0], [1, 1, 1], [0.5, 0.5, 0.5], [1.0, 0.0, 0.5], [-0.1, 0.5, 0.5], [0.5, 1.2, 0.5], [0.999999, 0.999999, 0.999999]]
    S = float(np.log2(ckpt["per_level_scale"]))
    emb, off = T(ckpt["embeddings"]), T(ckpt["offsets"].astype(np.int32))
    out = {}
    for name, mm in (("ref", ref), ("ours", ours), ("fma", fma)):
        y, dy = torch.empty(L, B, C, device=DEV), torch.empty(B, L * 3 * C, device=DEV)
        tx = T(x)
        mm["gridencoder"].grid_encode_forward(tx, emb, off, y, B, 3, C, L, S, ckpt["base_resolution"], dy, gridtype, align, interp)
        y2 = torch.empty(L, B, C, device=DEV)
        mm["gridencoder"].grid_encode_forward(tx, emb, off, y2, B, 3, C, L, S, ckpt["base_resolution"], None, gridtype, align, interp)
        torch.cuda.synchronize()
        assert torch.equal(y, y2)
        out[name] = (y, dy)
    e_fma, e_ref = float((out["ours"][0] - out["fma"][0]).abs().max()), float((out["ours"][0] - out["ref"][0]).abs().max())
    d_fma = rel(out["ours"][1], out["fma"][1])
    REPORT[f"grid_encode[{gridtype},{int(align)},{interp}]"] = dict(max_abs_vs_fma=e_fma, max_abs_vs_nocontract=e_ref, dy_dx_rel_vs_fma=d_fma,
                                                                   bits_vs_fma=mismatch(out["ours"][0], out["fma"][0])[0])
    print(REPORT[f"grid_encode[{gridtype},{int(align)},{interp}]"])
    # measured: bit-identical to the contracting build; vs the no-contraction build 9.5e-5 (linear) / 1.5e-4 (smoothstep: the weight's slope is 1.5x)
    assert e_fma <= 2e-6 and e_ref <= (1e-4 if interp == 0 else 2e-4)
    assert d_fma < 1e-4
    assert not out["ref"][0][:, 4].any() and not out["ours"][0][:, 4].any() and not out["ours"][0][:, 5].any()   # out-of-range inputs encode to zero


def test_grid_encode_half_equals_the_reference_half_kernel(mods, ckpt):
    """kernel_grid<at::Half,3,2>: the `--fp16` / autocast form (grid.py:43-44: embeddings.to(torch.half), half outputs) — configs[4]'s tables."""
    ref, fma, ours = mods
    rng = np.random.default_rng(12)
    B, L, C = 30000, 16, 2
    x = T(rng.random((B, 3)).astype(np.float32))
    emb, off = T(ckpt["embeddings"]).half(), T(ckpt["offsets"].astype(np.int32))
    S = float(np.log2(ckpt["per_level_scale"]))
    out = {}
    for name, mm in (("ref", ref), ("ours", ours), ("fma", fma)):
        y = torch.empty(L, B, C, device=DEV, dtype=torch.half)
        mm["gridencoder"].grid_encode_forward(x, emb, off, y, B, 3, C, L, S, ckpt["base_resolution"], None, 0, False, 0)
        torch.cuda.synchronize()
        out[name] = y
    a, r, f = (out[k].float() for k in ("ours", "ref", "fma"))
    frac_r, frac_f = float((a != r).float().mean()), float((a != f).float().mean())
    REPORT["grid_encode_half"] = dict(frac_differing_vs_nocontract=frac_r, frac_differing_vs_fma=frac_f, max_abs_vs_fma=float((a - f).abs().max()),
                                      max_abs_vs_nocontract=float((a - r).abs().max()))
    print(REPORT["grid_encode_half"])
    # half accumulation of 8 half-rounded corner products: one half ulp (2^-11 relative of O(1) features) when a weight's last float bit moves
    assert min(float((a - f).abs().max()), float((a - r).abs().max())) <= 2e-3 and min(frac_r, frac_f) < 0.02


@pytest.mark.parametrize("D,C,gridtype,align,interp", [(2, 2, 0, False, 0), (2, 4, 1, True, 1), (4, 2, 0, False, 0), (4, 1, 0, True, 1), (5, 2, 0, False, 0),
                                                       (5, 8, 1, False, 1)])
def test_grid_encode_other_input_dims_equal_the_reference_kernel(mods, D, C, gridtype, align, interp):
    """kernel_grid<float, D, C> for D = 2, 4, 5 (gridencoder.cu:386-399), kernel_grid_backward + kernel_input_backward (:430-444): the stand-alone op's
    other input dimensions (csrc/pn_grid_nd.hip) against the reference's own kernels, and the CPU oracle (oracle/grid_nd_oracle.cpp) against them too.
    Forward + dy_dx: <= 2e-6 against the contracting build (same arithmetic, one rounding of `pos`); backward: fp32 atomics in any order, 1e-4."""
    from pienerf_amd.gridencoder.grid import level_table_offsets
    ref, fma, ours = mods
    pls, base, L = 1.5, 4, 7
    cap = {2: 9, 4: 13, 5: 13}[D]
    offsets = level_table_offsets(D, L, pls, base, cap, align)          # low levels dense, upper levels hashed (or tiled) into 2^cap entries
    assert (np.diff(offsets) == 1 << cap).any() and (np.diff(offsets) < 1 << cap).any()
    rng = np.random.default_rng(20 + D)
    emb_np = rng.uniform(-1, 1, (int(offsets[-1]), C)).astype(np.float32)
    B = 6000
    x = rng.uniform(0, 1, (B, D)).astype(np.float32)
    x[0], x[1], x[2], x[3] = 0.0, 1.0, 0.5, 0.999999
    x[4, 0], x[5, D - 1] = -0.1, 1.2                                      # out of range in one coordinate: zeros, no gradient
    grad_np = rng.standard_normal((L, B, C)).astype(np.float32)
    emb, tx, off, grad = T(emb_np), T(x), T(offsets.astype(np.int32)), T(grad_np)
    S = float(np.log2(pls))
    out = {}
    for name, mm in (("ref", ref), ("ours", ours), ("fma", fma)):
        y, dy = torch.empty(L, B, C, device=DEV), torch.empty(B, L * D * C, device=DEV)
        mm["gridencoder"].grid_encode_forward(tx, emb, off, y, B, D, C, L, S, base, dy, gridtype, align, interp)
        y2 = torch.empty(L, B, C, device=DEV)
        mm["gridencoder"].grid_encode_forward(tx, emb, off, y2, B, D, C, L, S, base, None, gridtype, align, interp)
        ge, gi = torch.zeros_like(emb), torch.zeros(B, D, device=DEV)
        mm["gridencoder"].grid_encode_backward(grad, tx, emb, off, ge, B, D, C, L, S, base, dy, gi, gridtype, align, interp)
        tv = torch.zeros_like(emb)
        mm["gridencoder"].grad_total_variation(tx, emb, tv, off, 0.3, B, D, C, L, S, base, gridtype, align)   # kernel_grad_tv<float, D, C> (:506-634)
        torch.cuda.synchronize()
        assert torch.equal(y, y2)
        out[name] = (y, dy, ge, gi, tv)
    key = f"grid_nd[D={D},C={C},{gridtype},{int(align)},{interp}]"
    REPORT[key] = dict(outputs_vs_fma=mismatch(out["ours"][0], out["fma"][0]), outputs_vs_nocontract=mismatch(out["ours"][0], out["ref"][0]),
                       dy_dx_rel_vs_fma=rel(out["ours"][1], out["fma"][1]), grad_embeddings_rel=rel(out["ours"][2], out["ref"][2]),
                       grad_inputs_rel=rel(out["ours"][3], out["fma"][3]), grad_tv_rel=rel(out["ours"][4], out["ref"][4]))
    print(key, REPORT[key])
    assert REPORT[key]["outputs_vs_fma"][1] <= 2e-6 and REPORT[key]["outputs_vs_nocontract"][1] <= 2e-4
    assert REPORT[key]["dy_dx_rel_vs_fma"] < 1e-5 and REPORT[key]["grad_embeddings_rel"] < 1e-4 and REPORT[key]["grad_inputs_rel"] < 1e-4
    assert REPORT[key]["grad_tv_rel"] < 1e-4 and float(out["ref"][4].abs().max()) > 1e-4
    assert not out["ref"][0][:, 4:6].any() and not out["ours"][0][:, 4:6].any() and not out["ours"][3][4:6].any()
    # the CPU oracle, first-hand against the reference kernel (contracting build)
    want_y, want_dy = oracle.grid_nd_forward(x, emb_np, offsets, pls, base, gridtype, align, interp, dy_dx=True)
    y_fma = out["fma"][0].permute(1, 0, 2).reshape(B, L * C).cpu().numpy()
    assert np.abs(want_y - y_fma).max() <= 2e-6 and np.abs(want_dy - out["fma"][1].cpu().numpy()).max() <= 1e-5 * np.abs(want_dy).max()
    gi_o, ge_o = oracle.grid_nd_backward(grad_np.transpose(1, 0, 2).reshape(B, L * C), x, emb_np.shape, offsets, pls, base, want_dy, gridtype, align, interp)
    assert np.abs(ge_o - out["ref"][2].cpu().numpy()).max() <= 1e-4 * np.abs(ge_o).max()
    assert np.abs(gi_o - out["fma"][3].cpu().numpy()).max() <= 1e-4 * np.abs(gi_o).max()
    tv_o = oracle.grid_nd_grad_tv(x, emb_np, offsets, 0.3, pls, base, gridtype, align)
    assert np.abs(tv_o - out["ref"][4].cpu().numpy()).max() <= 1e-4 * np.abs(tv_o).max()


@pytest.mark.parametrize("degree", [1, 2, 3, 4, 5, 6, 7, 8])
def test_sh_encode_equals_the_reference_kernel(mods, degree):
    """kernel_sh (shencoder.cu:27-123 forward, :125-355 dy_dx) and kernel_sh_backward (:358-383).  Degrees 5-8 are evaluated by recurrence in double and
    narrowed once (csrc/pn_sh_bands.h) where the reference sums expanded float polynomials with coefficients up to 20: the gap is the REFERENCE's float
    rounding (a few ulp of its largest term), hence the wider bar there."""
    ref, fma, ours = mods
    rng = np.random.default_rng(4)
    B = 20000
    d = rng.standard_normal((B, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    grad = T(rng.standard_normal((B, degree * degree)).astype(np.float32))
    out = {}
    for name, mm in (("ref", ref), ("ours", ours), ("fma", fma)):
        td = T(d)
        y, dy, gi = torch.empty(B, degree ** 2, device=DEV), torch.empty(B, 3 * degree ** 2, device=DEV), torch.zeros(B, 3, device=DEV)
        mm["shencoder"].sh_encode_forward(td, y, B, 3, degree, dy)
        mm["shencoder"].sh_encode_backward(grad, td, B, 3, degree, dy, gi)
        torch.cuda.synchronize()
        out[name] = (y, dy, gi)
    for i, what in enumerate(("y", "dy_dx", "grad_inputs")):
        e = min(float((out["ours"][i] - out["ref"][i]).abs().max()), float((out["ours"][i] - out["fma"][i]).abs().max()))
        REPORT[f"sh[{degree}].{what}"] = dict(vs_ref=mismatch(out["ours"][i], out["ref"][i]), vs_fma=mismatch(out["ours"][i], out["fma"][i]))
        assert e <= ((1e-6 if i == 0 else 2e-5) if degree <= 4 else (1e-5 if i == 0 else 4e-4)), (what, e)


# ------------------------------------------------------------------------------------------------ the CPU oracle against the reference, first-hand
@pytest.mark.parametrize("num_seek_IP,max_iter_num,n_step,cut", [(1, 1, 1, False), (3, 1, 8, False), (2, 3, 4, False), (3, 5, 8, False), (1, 1, 6, True), (3, 1, 2, True)])
def test_cpu_oracle_march_equals_the_reference_kernel_directly(mods, deformed_ip_state, small_opt, ckpt, num_seek_IP, max_iter_num, n_step, cut):
    """oracle/render_oracle.cpp (the CPU restatement every other test leans on) against kernel_march_rays_quadratic_bending itself — no HIP kernel of this
    repository in between: the same inputs, xyzs / dirs / deltas bit for bit."""
    ref, _, _ = mods
    ip, ck = deformed_ip_state, dict(ckpt)
    if cut:
        blobs = np.repeat(np.random.default_rng(5).random(len(ck["density_bitfield"]) // 64) < 0.04, 64)
        ck["density_bitfield"] = ck["density_bitfield"] | np.where(blobs, 0xFF, 0).astype(np.uint8)
        W = 40
        o, d = oracle.get_rays(scene.orbit_pose(4.0, 10.0, -5.0), scene.orbit_intrinsics(W, W, 50.0), W, W)
        hgs = np.float32(small_opt["hash_grid_size"])
        bbmin, bbmax, res = oracle.render_bbox(ip["p_def"], hgs, cut=True, bound=1.0)
        n_grid = int(res.prod())
        pig = oracle.get_pnts_in_grids(len(ip["p_def"]), n_grid, ip["p_def"], bbmin, bbmax, hgs, res)
        nears, fars = oracle.near_far_from_aabb(o, d, np.concatenate([bbmin, bbmax]), 0.2)
        m = dict(o=o, d=d, hgs=hgs, bbmin=bbmin, bbmax=bbmax, res=res, n_grid=n_grid, pig=pig, nears=nears, fars=fars)
        alive = np.arange(W * W, dtype=np.int32)
        cb = np.array([-0.3, 0.9, -0.9, 0.5, -0.9, 0.9], np.float32)
        dt_gamma, max_steps = 1.0 / 128, 300
    else:
        m = _march_inputs(ip, small_opt, ck, W=96)
        alive = np.nonzero(m["nears"] < 1e30)[0].astype(np.int32)
        alive = np.concatenate([alive, np.arange(0, m["o"].shape[0], 97, dtype=np.int32)])
        cb = np.zeros(6, np.float32)
        dt_gamma, max_steps = 0.0, 1024
    noises = np.random.default_rng(1).random(len(alive)).astype(np.float32) if n_step == 4 else np.zeros(len(alive), np.float32)
    r = _call_march(ref["raymarching"], m, ip, ck, alive, n_step, num_seek_IP, max_iter_num, cut, cb, dt_gamma, max_steps, noises, m["nears"])
    want = oracle.march_rays_quadratic_bending(*m["pig"], len(ip["p_def"]), m["n_grid"], ip["p_def"], ip["p_ori"], ip["F"], ip["dF"], max_iter_num, m["bbmin"],
                                               m["bbmax"], m["hgs"], m["res"], num_seek_IP, np.float32(ip["IP_dx"]), cut, cb, len(alive), n_step, alive,
                                               m["nears"], m["o"], m["d"], 1.0, ck["density_bitfield"], ck["cascade"], ck["grid_size"], m["nears"], m["fars"],
                                               128, False, dt_gamma, max_steps, noises=noises)
    assert int((want[2][:, 0] != 0).sum()) > 100
    for name, a, b in zip(("xyzs", "dirs", "deltas"), r[:3], want):
        a = a.cpu().numpy()
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), f"{name}: {int(np.sum(a != b))} of {a.size} values differ"


def test_cpu_oracle_equals_the_reference_kernels_directly(mods, ckpt):
    """near/far, morton, packbits, composite, hash grid (float, all four option sets) and SH: the CPU oracle against the reference's kernels on the same
    inputs, without this repository's HIP path in between."""
    ref, fma, _ = mods
    W = 96
    o, d = oracle.get_rays(scene.orbit_pose(5.0, 30.0, -20.0), scene.orbit_intrinsics(W, W, 50.0), W, W)
    N = len(o)
    aabb = np.array([-0.6, -0.8, -0.5, 0.7, 0.9, 0.55], np.float32)
    n, f = torch.empty(N, device=DEV), torch.empty(N, device=DEV)
    ref["raymarching"].near_far_from_aabb(T(o), T(d), T(aabb), N, 0.2, n, f)
    wn, wf = oracle.near_far_from_aabb(o, d, aabb, 0.2)
    assert np.array_equal(n.cpu().numpy().view(np.uint32), wn.view(np.uint32)) and np.array_equal(f.cpu().numpy().view(np.uint32), wf.view(np.uint32))
    # morton / packbits
    rng = np.random.default_rng(9)
    coords = rng.integers(0, 1024, (5000, 3)).astype(np.int32)
    grid = (rng.random(128 ** 3).astype(np.float32) * 20)
    idx, back = torch.empty(5000, dtype=torch.int32, device=DEV), torch.empty(5000, 3, dtype=torch.int32, device=DEV)
    bits = torch.zeros(128 ** 3 // 8, dtype=torch.uint8, device=DEV)
    ref["raymarching"].morton3D(T(coords), 5000, idx)
    ref["raymarching"].morton3D_invert(idx, 5000, back)
    ref["raymarching"].packbits(T(grid), 128 ** 3 // 8, 10.0, bits)
    torch.cuda.synchronize()
    assert np.array_equal(idx.cpu().numpy().astype(np.uint32), np.asarray(oracle.morton3D(coords)).astype(np.uint32))
    assert np.array_equal(back.cpu().numpy(), np.asarray(oracle.morton3D_invert(idx.cpu().numpy())).reshape(-1, 3))
    assert np.array_equal(bits.cpu().numpy(), oracle.packbits(grid, 10.0))
    # composite
    Nr, n_alive, n_step = 3000, 1700, 8
    alive = np.sort(rng.choice(Nr, n_alive, replace=False)).astype(np.int32)
    M = n_alive * n_step
    sig = (rng.random(M).astype(np.float32) * 120)
    rgb = rng.random((M, 3)).astype(np.float32)
    deltas = np.stack([np.full(M, 0.0034, np.float32), (rng.random(M) * 0.01 + 0.0034).astype(np.float32)], 1)
    for k in range(0, n_alive, 3):
        deltas[k * n_step + rng.integers(0, n_step):(k + 1) * n_step] = 0
    st = dict(t=(rng.random(Nr).astype(np.float32) + 3), ws=(rng.random(Nr).astype(np.float32) * 0.9), dep=rng.random(Nr).astype(np.float32),
              img=rng.random((Nr, 3)).astype(np.float32))
    g = {k: T(v.copy()) for k, v in st.items()}
    al = T(alive.copy())
    keep = (T(sig), T(rgb), T(deltas))
    ref["raymarching"].composite_rays(n_alive, n_step, 1e-2, al, g["t"], keep[0], keep[1], keep[2], g["ws"], g["dep"], g["img"])
    torch.cuda.synchronize()
    w = {k: v.copy() for k, v in st.items()}
    wal = alive.copy()
    oracle.composite_rays(n_alive, n_step, wal, w["t"], sig, rgb, deltas, w["ws"], w["dep"], w["img"], 1e-2)
    assert np.array_equal(al.cpu().numpy(), wal) and (wal < 0).any() and (wal >= 0).any()
    assert np.array_equal(g["t"].cpu().numpy().view(np.uint32), w["t"].view(np.uint32))   # a running sum of deltas: nothing to approximate
    for k in ("ws", "dep", "img"):   # __expf: the GPU's v_exp_f32 against the oracle's exp2f(x * log2e)
        assert np.abs(g[k].cpu().numpy() - w[k]).max() / np.abs(w[k]).max() < 2e-6, k
    # hash grid: the oracle states pos = fmaf(u, scale, 0.5) as one rounding (what a contracting compiler makes of gridencoder.cu:129)
    B, L, C = 20000, 16, 2
    x = rng.random((B, 3)).astype(np.float32)
    x[:7] = [[0, 0, 0], [1, 1, 1], [0.5, 0.5, 0.5], [1.0, 0.0, 0.5], [-0.1, 0.5, 0.5], [0.5, 1.2, 0.5], [0.999999, 0.999999, 0.999999]]
    S = float(np.log2(ckpt["per_level_scale"]))
    emb, off = T(ckpt["embeddings"]), T(ckpt["offsets"].astype(np.int32))
    for gridtype, align, interp in ((0, False, 0), (1, False, 0), (0, True, 0), (0, False, 1)):
        want = oracle.grid_encode_forward(x, ckpt["embeddings"], ckpt["offsets"], ckpt["per_level_scale"], ckpt["base_resolution"], gridtype, align, interp)
        got = {}
        for name, mm in (("ref", ref), ("fma", fma)):
            y = torch.empty(L, B, C, device=DEV)
            mm["gridencoder"].grid_encode_forward(T(x), emb, off, y, B, 3, C, L, S, ckpt["base_resolution"], None, gridtype, align, interp)
            torch.cuda.synchronize()
            got[name] = y.permute(1, 0, 2).reshape(B, L * C).cpu().numpy()
        e_fma, e_ref = float(np.abs(got["fma"] - want).max()), float(np.abs(got["ref"] - want).max())
        REPORT[f"oracle_grid_encode[{gridtype},{int(align)},{interp}]"] = dict(max_abs_vs_fma=e_fma, max_abs_vs_nocontract=e_ref)
        # (against the no-contraction build the one rounding of `pos` moves a finest-level weight by up to 1.2e-4 x a feature of O(1))
        assert e_fma <= 2e-6 and e_ref <= (1.5e-4 if interp == 0 else 2.5e-4), (gridtype, align, interp, e_fma, e_ref)
    # SH
    dd = rng.standard_normal((B, 3)).astype(np.float32)
    dd /= np.linalg.norm(dd, axis=-1, keepdims=True)
    for degree in (1, 2, 3, 4, 6, 8):
        y = torch.empty(B, degree ** 2, device=DEV)
        ref["shencoder"].sh_encode_forward(T(dd), y, B, 3, degree, None)
        torch.cuda.synchronize()
        assert np.abs(y.cpu().numpy() - oracle.sh_encode_forward(dd, degree)).max() <= (1e-6 if degree <= 4 else 1e-5), degree


# ------------------------------------------------------------------------------------------------ training ops
def test_march_rays_train_equals_the_reference_kernel(mods):
    """kernel_march_rays_train (raymarching.cu:314-483).  The reference hands out ray rows and point ranges with two atomicAdd counters (race order); the
    drop-in writes rows in ray order with prefix-sum ranges (documented difference).  Compared here: the counters, each ray's sample count,
    and each ray's samples bit for bit through its own (offset, count) row."""
    ref, fma, ours = mods
    for bound, dt_gamma, max_steps, W in ((1.0, 0.0, 1024, 64), (2.0, 1.0 / 128, 300, 48)):
        ck = scene.make_checkpoint(bound=bound, seed=1)
        o, d = oracle.get_rays(scene.orbit_pose(3.4 * bound, 25.0, -20.0), scene.orbit_intrinsics(W, W, 50.0), W, W)
        nears, fars = oracle.near_far_from_aabb(o, d, np.array([-bound] * 3 + [bound] * 3, np.float32), 0.2)
        N = len(o)
        M = N * 256
        noise = np.random.default_rng(3).random(N).astype(np.float32)
        out = {}
        for name, mm in (("ref", ref), ("ours", ours)):
            xyzs, dirs, deltas = torch.zeros(M, 3, device=DEV), torch.zeros(M, 3, device=DEV), torch.zeros(M, 2, device=DEV)
            rays, counter = torch.empty(N, 3, dtype=torch.int32, device=DEV), torch.zeros(2, dtype=torch.int32, device=DEV)
            keep = (T(o), T(d), T(ck["density_bitfield"]), T(nears), T(fars), T(noise))
            mm["raymarching"].march_rays_train(keep[0], keep[1], keep[2], bound, dt_gamma, max_steps, N, ck["cascade"], ck["grid_size"], M, keep[3], keep[4], xyzs, dirs,
                                               deltas, rays, counter, keep[5])
            torch.cuda.synchronize()
            out[name] = [t.cpu().numpy() for t in (xyzs, dirs, deltas, rays, counter)]
        r, g = out["ref"], out["ours"]
        assert np.array_equal(r[4], g[4]) and r[4][0] > 500 and r[4][0] < M
        rr, gr = r[3][np.argsort(r[3][:, 0])], g[3][np.argsort(g[3][:, 0])]
        assert np.array_equal(rr[:, 0], gr[:, 0]) and np.array_equal(rr[:, 2], gr[:, 2])
        gather = lambda res, rows: [np.concatenate([a[o_:o_ + n] for _, o_, n in rows]) for a in res[:3]]
        for a, b in zip(gather(g, gr), gather(r, rr)):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.mark.parametrize("T_thresh", [1e-4, 5e-2])
def test_composite_rays_train_equals_the_reference_kernels(mods, T_thresh):
    from test_gpu_training import _ray_batch
    ref, fma, ours = mods
    rng = np.random.default_rng(4)
    N = 5000
    rays, sig, rgb, deltas = _ray_batch(rng, N, 120)
    M = len(sig)
    gws, gim = T(rng.standard_normal(N).astype(np.float32)), T(rng.standard_normal((N, 3)).astype(np.float32))
    out = {}
    for name, mm in (("ref", ref), ("ours", ours)):
        keep = (T(sig), T(rgb), T(deltas), T(rays))
        ws, dep, img = torch.zeros(N, device=DEV), torch.zeros(N, device=DEV), torch.zeros(N, 3, device=DEV)
        mm["raymarching"].composite_rays_train_forward(keep[0], keep[1], keep[2], keep[3], M, N, T_thresh, ws, dep, img)
        gs, gc = torch.zeros(M, device=DEV), torch.zeros(M, 3, device=DEV)
        mm["raymarching"].composite_rays_train_backward(gws, gim, keep[0], keep[1], keep[2], keep[3], ws, img, M, N, T_thresh, gs, gc)
        torch.cuda.synchronize()
        out[name] = (ws, dep, img, gs, gc)
    for i, (what, bar) in enumerate((("weights_sum", 1e-5), ("depth", 1e-5), ("image", 1e-5), ("grad_sigmas", 1e-4), ("grad_rgbs", 1e-5))):
        e = rel(out["ours"][i], out["ref"][i])
        REPORT[f"composite_train[{T_thresh}].{what}"] = e
        assert e < bar, (what, e)


@pytest.mark.parametrize("interp", [0, 1])
def test_grid_encode_backward_and_tv_equal_the_reference_kernels(mods, interp):
    """kernel_grid_backward<float>, kernel_input_backward, kernel_grad_tv (gridencoder.cu:248-369, 506-611).  Both sides scatter with fp32 atomics in
    unordered summation: 1e-4 relative."""
    from pienerf_amd.gridencoder.grid import level_table_offsets
    ref, fma, ours = mods
    pls, base, L, C = 1.6, 8, 6, 2
    offsets = level_table_offsets(3, L, pls, base, 12, False)
    rng = np.random.default_rng(6)
    emb = T(rng.uniform(-1, 1, (int(offsets[-1]), C)).astype(np.float32))
    B = 20000
    x = rng.uniform(0, 1, (B, 3)).astype(np.float32)
    x[:9] = [1.2, 0.5, 0.5]
    tx, off = T(x), T(offsets.astype(np.int32))
    grad = T(rng.standard_normal((L, B, C)).astype(np.float32))      # [L, B, C] as grid.py:75 hands it over
    S = float(np.log2(pls))
    out = {}
    for name, mm in (("ref", ref), ("ours", ours)):
        y, dy = torch.empty(L, B, C, device=DEV), torch.empty(B, L * 3 * C, device=DEV)
        mm["gridencoder"].grid_encode_forward(tx, emb, off, y, B, 3, C, L, S, base, dy, 0, False, interp)
        ge, gi = torch.zeros_like(emb), torch.zeros(B, 3, device=DEV)
        mm["gridencoder"].grid_encode_backward(grad, tx, emb, off, ge, B, 3, C, L, S, base, dy, gi, 0, False, interp)
        ge2 = torch.zeros_like(emb)
        mm["gridencoder"].grid_encode_backward(grad, tx, emb, off, ge2, B, 3, C, L, S, base, None, None, 0, False, interp)
        tv = torch.zeros_like(emb)
        mm["gridencoder"].grad_total_variation(tx, emb, tv, off, 0.3, B, 3, C, L, S, base, 0, False)
        torch.cuda.synchronize()
        out[name] = (y, dy, ge, gi, ge2, tv)
    for i, what in enumerate(("outputs", "dy_dx", "grad_embeddings", "grad_inputs", "grad_embeddings_no_dy_dx", "grad_tv")):
        e = rel(out["ours"][i], out["ref"][i])
        REPORT[f"grid_backward[{interp}].{what}"] = e
        assert e < 1e-4, (what, e)
    assert float(out["ref"][5].abs().max()) > 1e-4 and not out["ours"][3][:9].any()


# ------------------------------------------------------------------------------------------------ whole frames through the reference's kernels
def _frame_through_backend(mm, net, ip, o, rays_o, rays_d, pig_fn):
    """nerf/renderer.py:782-907 (rund_cuda) driven on one set of backend modules; the MLP is torch fp32 (as in the reference).  Returns the image and
    the integer trip record."""
    N = rays_o.shape[0]
    p_def, p_ori, F_IP, dF_IP = (T(ip[k]) for k in ("p_def", "p_ori", "F", "dF"))
    hgs = o["hash_grid_size"]
    bbmin = p_def.min(0).values - 1e-3 * torch.ones(3, device=DEV)
    bbmax = p_def.max(0).values + 1e-3 * torch.ones(3, device=DEV)
    resolution = torch.ceil((bbmax - bbmin) / hgs).to(torch.int32)
    aabb = torch.cat((bbmin, bbmax), 0)
    nears, fars = torch.empty(N, device=DEV), torch.empty(N, device=DEV)
    mm["raymarching"].near_far_from_aabb(rays_o, rays_d, aabb, N, net.min_near, nears, fars)
    n_vtx, n_grid = p_ori.shape[0], int(resolution.prod())
    pig_cnt, pig_bgn, pig_idx = pig_fn(n_vtx, n_grid, p_def, bbmin, bbmax, hgs, resolution)
    weights_sum, depth, image = torch.zeros(N, device=DEV), torch.zeros(N, device=DEV), torch.zeros(N, 3, device=DEV)
    rays_alive, rays_t = torch.arange(N, dtype=torch.int32, device=DEV), nears.clone()
    cut_bounds = torch.tensor(o["cut_bounds"], dtype=torch.float32, device=DEV)
    enc, sig_net, col_net = net.encoder, net.sigma_net, net.color_net
    S = float(np.log2(enc.per_level_scale))
    offs = enc.offsets.to(DEV, torch.int32)
    step, record, samples = 0, [], 0
    while step < o["max_steps"]:
        n_alive = rays_alive.shape[0]
        if n_alive <= 0:
            break
        n_step = max(min(N // n_alive, 8), 1)
        M = n_alive * n_step
        M += 128 - (M % 128)
        xyzs, dirs, deltas = torch.zeros(M, 3, device=DEV), torch.zeros(M, 3, device=DEV), torch.zeros(M, 2, device=DEV)
        noises = torch.zeros(n_alive, device=DEV)
        mm["raymarching"].march_rays_quadratic_bending(pig_cnt, pig_bgn, pig_idx, n_vtx, n_grid, p_def, p_ori, F_IP, dF_IP, o["max_iter_num"], bbmin, bbmax, hgs,
                                                       resolution, o["num_seek_IP"], float(ip["IP_dx"]), False, cut_bounds, n_alive, n_step, rays_alive, rays_t,
                                                       rays_o, rays_d, net.bound, o["dt_gamma"], o["max_steps"], net.cascade, net.grid_size, net.density_bitfield,
                                                       nears, fars, xyzs, dirs, deltas, noises)
        u = ((xyzs + net.bound) / (2 * net.bound)).contiguous()
        feat = torch.empty(16, M, 2, device=DEV)
        mm["gridencoder"].grid_encode_forward(u, enc.embeddings.detach(), offs, feat, M, 3, 2, 16, S, 16, None, 0, False, 0)
        h = feat.permute(1, 0, 2).reshape(M, 32)
        h = torch.relu(h @ sig_net[0].weight.t()) @ sig_net[1].weight.t()
        sigmas = torch.exp(h[:, 0])
        sh = torch.empty(M, 16, device=DEV)
        mm["shencoder"].sh_encode_forward(dirs.contiguous(), sh, M, 3, 4, None)
        hc = torch.cat([sh, h[:, 1:]], -1)
        hc = torch.relu(torch.relu(hc @ col_net[0].weight.t()) @ col_net[1].weight.t()) @ col_net[2].weight.t()
        rgbs = torch.sigmoid(hc)
        mm["raymarching"].composite_rays(n_alive, n_step, o["T_thresh"], rays_alive, rays_t, sigmas.contiguous(), rgbs.contiguous(), deltas, weights_sum, depth, image)
        torch.cuda.synchronize()
        samples += int((deltas[:, 0] != 0).sum())
        rays_alive = rays_alive[rays_alive >= 0]
        record.append((n_alive, n_step))
        step += n_step
    image = image + (1 - weights_sum).unsqueeze(-1)
    return dict(image=image, weights_sum=weights_sum, depth=depth, record=record, samples=samples, survivors=rays_alive)


@pytest.mark.parametrize("num_seek_IP,max_iter_num", [(3, 1), (1, 1), (3, 5)])
def test_frame_loop_on_the_reference_kernels_equals_the_fused_frame(mods, deformed_ip_state, small_opt, ckpt, num_seek_IP, max_iter_num):
    """One whole deformed frame three ways: the reference's rund_cuda loop on the REFERENCE'S kernels, the same loop on the drop-in backends, and the
    fused frame driver of the product (pn_render_deformed).  Trip structure (alive counts, n_step per trip) and sample totals are integer outcomes:
    equal.  Radiance within 1e-4."""
    from pienerf_amd.nerf.network import NeRFNetwork
    from pienerf_amd.nerf.utils import get_pnts_in_grids
    ref, fma, ours = mods
    ip = deformed_ip_state
    o = dict(small_opt, W=96, H=96, num_seek_IP=num_seek_IP, max_iter_num=max_iter_num)
    ro, rd = oracle.get_rays(scene.orbit_pose(o["radius"], 20.0, -10.0), scene.orbit_intrinsics(96, 96, o["fovy"]), 96, 96)
    rays_o, rays_d = T(ro), T(rd)
    net = NeRFNetwork(encoding="hashgrid", bound=1.0, cuda_ray=True).to(DEV).load_checkpoint_dict(ckpt)
    net.p_def, net.p_ori, net.IP_F, net.IP_dF, net.IP_dx = T(ip["p_def"]), T(ip["p_ori"]), T(ip["F"]), T(ip["dF"]), ip["IP_dx"]
    with torch.no_grad():
        a = _frame_through_backend(ref, net, ip, o, rays_o, rays_d, get_pnts_in_grids)
        b = _frame_through_backend(ours, net, ip, o, rays_o, rays_d, get_pnts_in_grids)
        c = net.render_deformed(rays_o[None], rays_d[None], collect_stats=True, **o)
        st = dict(net.last_stats)
    assert a["record"] == b["record"] and a["samples"] == b["samples"] == st["samples"] and len(a["record"]) == st["trips"]
    assert torch.equal(a["survivors"], b["survivors"])
    assert a["samples"] > 5000
    assert float((a["image"] - b["image"]).abs().max()) < 2e-5 and float((a["weights_sum"] - b["weights_sum"]).abs().max()) < 2e-5
    assert float((a["image"] - c["image"][0]).abs().max()) < 1e-4 and float((a["weights_sum"] - c["weights_sum"]).abs().max()) < 1e-4
    assert float((a["depth"] - c["depth_0"][0]).abs().max()) < 1e-4 * max(1.0, float(a["depth"].max()))
    REPORT[f"frame[seek{num_seek_IP},iter{max_iter_num}]"] = dict(trips=len(a["record"]), samples=a["samples"],
                                                                 image_max_abs_ref_vs_dropin_ops=float((a["image"] - b["image"]).abs().max()),
                                                                 image_max_abs_ref_vs_fused=float((a["image"] - c["image"][0]).abs().max()))


def test_full_size_chair_frame_on_the_reference_kernels():
    """configs[1] itself (800x800, sim_dx 0.05, 3 576 IPs, num_seek_IP 3): the frame the bench measures, rendered by the reference's own kernels
    (one lane per ray, every trip synchronised) and by the fused frame driver."""
    from pienerf_amd.harness import SimRenderHarness
    from pienerf_amd.nerf.utils import get_pnts_in_grids
    ref = {e: ref_build.load(e, False) for e in ref_build.EXTS}
    opt = scene.default_opt()
    h = SimRenderHarness(opt, device=DEV, overlap_sim=False)
    h.sim.update_force(h.sim.n_IP // 2, torch.tensor([300.0, 100.0, -200.0], dtype=torch.float64))
    for _ in range(6):
        h.sim.stepforward()
    torch.cuda.synchronize()
    p_def, F, dF = h.sim.get_IP_info()
    ip = dict(p_def=p_def.cpu().numpy(), p_ori=h.model.p_ori.cpu().numpy(), F=F.cpu().numpy(), dF=dF.cpu().numpy(), IP_dx=h.model.IP_dx)
    net = h.model
    net.p_def, net.IP_F, net.IP_dF = p_def, F, dF
    ro, rd = oracle.get_rays(h.pose, h.intrinsics, opt["H"], opt["W"])
    rays_o, rays_d = T(ro), T(rd)
    with torch.no_grad():
        a = _frame_through_backend(ref, net, ip, opt, rays_o, rays_d, get_pnts_in_grids)
        c = net.render_deformed(rays_o[None], rays_d[None], collect_stats=True, **opt)
        st = dict(net.last_stats)
    assert a["samples"] == st["samples"] > 500_000 and len(a["record"]) == st["trips"], (a["samples"], st)
    e_img, e_ws = float((a["image"] - c["image"][0]).abs().max()), float((a["weights_sum"] - c["weights_sum"]).abs().max())
    REPORT["frame_full_size_chair"] = dict(trips=st["trips"], samples=st["samples"], image_max_abs=e_img, weights_sum_max_abs=e_ws, record=a["record"])
    print(REPORT["frame_full_size_chair"])
    assert e_img < 1e-4 and e_ws < 1e-4
    # ... and by the launch sets bench.py's pipelines run (harness._HipBackend: three lanes = the first trip's network / composite / compaction folded into
    # the fused launch on 128 workgroups, march pass 1 in its throughput form; two lanes = the whole frame in the launch on 160 workgroups): the modes that
    # produce `value` against the reference's kernels first-hand, not through the trip-by-trip form
    forms = {"fold_128_three_lanes": dict(fused_from=1, fused_whole=False, fused_fold=True, fused_grid=128, march_throughput=64),
             "whole_160_two_lanes": dict(fused_from=0, fused_whole=True, fused_fold=False, fused_grid=160, march_throughput=64)}
    for name, kw in forms.items():
        with torch.no_grad():
            f = net.render_deformed(rays_o[None], rays_d[None], collect_stats=True, **dict(opt, **kw))
            stf = dict(net.last_stats)
            mode = net.fused_clocks(slot=0)
        assert (mode["mode"], mode["first_trip"]) == ((2, 1) if kw["fused_fold"] else (1, 0)), (name, mode)   # the launch really ran in that form
        assert stf["samples"] == a["samples"] and stf["trips"] == len(a["record"]), (name, stf)
        assert torch.equal(f["image"], c["image"]) and torch.equal(f["depth_0"], c["depth_0"]), name             # every launch form gives the same bits
        e_f = float((a["image"] - f["image"][0]).abs().max())
        REPORT["frame_full_size_chair"][name] = dict(image_max_abs=e_f, samples=stf["samples"], trips=stf["trips"])
        assert e_f < 1e-4, (name, e_f)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "ref_parity_report_fullsize.json"), "w") as f:
        json.dump(REPORT["frame_full_size_chair"], f)
