"""GPU edge cases of the render path: empty and ragged inputs, rays that all miss, a whole frame in --cut mode, and the two
march launches pushed to their extremes (every ray through the 8-lane launch / through the wave-per-ray tail launch) — the
samples must not depend on that split, bit for bit."""
import numpy as np
import pytest
import torch

import oracle
from pienerf_amd import scene
from test_gpu_parity import DEV, T, _march_inputs

pytestmark = pytest.mark.gpu


def _net(ckpt, ip):
    from pienerf_amd.nerf.network import NeRFNetwork
    net = NeRFNetwork(encoding="hashgrid", bound=1.0, cuda_ray=True).to(DEV).load_checkpoint_dict(ckpt)
    net.p_def, net.p_ori, net.IP_F, net.IP_dF, net.IP_dx = T(ip["p_def"]), T(ip["p_ori"]), T(ip["F"]), T(ip["dF"]), ip["IP_dx"]
    return net


@pytest.fixture
def tail_rounds():
    from pienerf_amd._lib import check, lib

    def set_rounds(r):
        check(lib().pn_march_set_tail_rounds(int(r)), "set_tail_rounds")
    yield set_rounds
    set_rounds(0)


def test_empty_inputs_are_no_ops(ckpt):
    from pienerf_amd import raymarching
    from pienerf_amd.gridencoder import GridEncoder
    from pienerf_amd.shencoder import SHEncoder
    z3 = torch.zeros(0, 3, device=DEV)
    n, f = raymarching.near_far_from_aabb(z3, z3, T(np.array([-1, -1, -1, 1, 1, 1], np.float32)), 0.2)
    assert n.numel() == 0 and f.numel() == 0
    out = raymarching.compact_rays(torch.zeros(0, dtype=torch.int32, device=DEV))
    assert out.numel() == 0
    assert raymarching.compact_rays(torch.full((300,), -1, dtype=torch.int32, device=DEV)).numel() == 0  # nothing survives
    enc = GridEncoder(input_dim=3, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=19, desired_resolution=2048).to(DEV)
    assert enc(z3, bound=1.0).shape == (0, 32)
    assert SHEncoder(input_dim=3, degree=4).to(DEV)(z3).shape == (0, 16)


@pytest.mark.parametrize("W,H", [(1, 1), (33, 7), (257, 3)])
def test_ragged_ray_counts(W, H):
    from pienerf_amd import raymarching
    from pienerf_amd.nerf.utils import get_rays
    pose = scene.orbit_pose(4.0, 40.0, -10.0)
    intr = scene.orbit_intrinsics(W, H, 50.0)
    o_ref, d_ref = oracle.get_rays(pose, intr, H, W)
    r = get_rays(T(pose[None]), intr, H, W)
    assert np.array_equal(r["rays_o"][0].cpu().numpy(), o_ref) and np.array_equal(r["rays_d"][0].cpu().numpy(), d_ref)
    aabb = np.array([-0.5, -0.5, -0.5, 0.5, 0.5, 0.5], np.float32)
    n_ref, f_ref = oracle.near_far_from_aabb(o_ref, d_ref, aabb, 0.2)
    n, f = raymarching.near_far_from_aabb(r["rays_o"][0], r["rays_d"][0], T(aabb), 0.2)
    assert np.array_equal(n.cpu().numpy(), n_ref) and np.array_equal(f.cpu().numpy(), f_ref)


def test_frame_whose_rays_all_miss(deformed_ip_state, small_opt, ckpt):
    """Camera looking away from the object: no ray enters the IP box, the loop ends after trip 0 with zero samples."""
    W = 40
    pose = scene.orbit_pose(5.0, 20.0, -15.0)
    pose[:3, :3] = -pose[:3, :3]  # turn the camera around
    o, d = oracle.get_rays(pose, scene.orbit_intrinsics(W, W, 50.0), W, W)
    ref = oracle.render_deformed(o, d, deformed_ip_state, ckpt, small_opt)
    net = _net(ckpt, deformed_ip_state)
    with torch.no_grad():
        out = net.render_deformed(T(o)[None], T(d)[None], collect_stats=True, **small_opt)
    st = dict(net.last_stats)
    assert ref["samples"] == 0 and st["samples"] == 0 and st["alive_at_exit"] == 0 and st["err"] == 0
    assert np.array_equal(out["image"][0].cpu().numpy(), ref["image"])  # pure background
    dep = out["depth"][0].cpu().numpy()
    assert np.array_equal(np.isfinite(dep), np.isfinite(ref["depth"]))   # NaN wherever nears == fars == FLT_MAX (renderer.py:898)
    assert np.isnan(dep).sum() > 0.8 * dep.size                          # most rays miss the IP box altogether
    assert np.array_equal(dep[np.isfinite(dep)], ref["depth"][np.isfinite(dep)])  # a grazing ray without samples has depth 0


@pytest.mark.parametrize("num_seek_IP", [1, 3])
def test_frame_in_cut_mode(deformed_ip_state, small_opt, ckpt, num_seek_IP):
    """--cut through the fused frame driver: bbox = +-bound, samples outside cut_bounds are un-warped background."""
    W = 56
    opt = dict(small_opt, cut=True, cut_bounds=[-0.3, 0.9, -0.9, 0.5, -0.9, 0.9], num_seek_IP=num_seek_IP, max_steps=256)
    pose = scene.orbit_pose(4.0, 10.0, -5.0)
    o, d = oracle.get_rays(pose, scene.orbit_intrinsics(W, W, 50.0), W, W)
    ref = oracle.render_deformed(o, d, deformed_ip_state, ckpt, opt)
    net = _net(ckpt, deformed_ip_state)
    with torch.no_grad():
        out = net.render_deformed(T(o)[None], T(d)[None], collect_stats=True, **opt)
    st = dict(net.last_stats)
    assert ref["samples"] > 1000
    assert st["trips"] == ref["trips"] and st["samples"] == ref["samples"] and st["err"] == 0 and st["alive_at_exit"] == 0
    assert np.abs(out["image"][0].cpu().numpy() - ref["image"]).max() < 1e-4
    assert np.abs(out["weights_sum"].cpu().numpy() - ref["weights_sum"]).max() < 1e-4


def test_cut_frame_with_points_outside_the_grid(deformed_ip_state, small_opt, ckpt):
    """--cut: the spatial hash spans +-bound whatever the body does, and the reference files a point that has left it under its FLAT cell index when that
    still lies in [0, n_grid) (nerf/utils.py:389-407: no per-axis test) — a cell far from the body.  The frame prologue builds candidate lists only near
    the cells that hold points (PnFrameDev::ip_lo / ip_hi): those are the cells the points are FILED in, wrapped ones included, so the frame still equals
    the oracle's (which restates p2g literally): a fifth of the points pushed 0.2-0.5 beyond the +x face (filed in the next row's first cells)."""
    ip = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in deformed_ip_state.items()}
    rng = np.random.default_rng(9)
    n = len(ip["p_def"])
    far = rng.choice(n, n // 5, replace=False)
    ip["p_def"][far, 0] = 1.0 + rng.uniform(0.2, 0.5, len(far)).astype(np.float32)
    # (a flat index outside [0, n_grid) — beyond -z, or +x in the grid's last row — is error flag 2 here, loudly, where the reference drops the point silently)
    W = 64
    opt = dict(small_opt, cut=True, cut_bounds=[-0.9, 0.95, -0.9, 0.9, -0.95, 0.9], num_seek_IP=3, max_steps=256)
    o, d = oracle.get_rays(scene.orbit_pose(4.0, 35.0, -15.0), scene.orbit_intrinsics(W, W, 50.0), W, W)
    ref = oracle.render_deformed(o, d, ip, ckpt, opt)
    net = _net(ckpt, ip)
    with torch.no_grad():
        out = net.render_deformed(T(o)[None], T(d)[None], collect_stats=True, **opt)
    st = dict(net.last_stats)
    assert ref["samples"] > 1000
    assert st["trips"] == ref["trips"] and st["samples"] == ref["samples"] and st["alive_at_exit"] == 0, (st, ref["samples"], ref["trips"])
    assert np.abs(out["image"][0].cpu().numpy() - ref["image"]).max() < 1e-4


def test_frame_in_trex_configuration(deformed_ip_state):
    """The option set of BASELINE config 3 (README.md:134 of the reference) at test size: bound 2 -> two density cascades and the
    4096-resolution hash grid, dt_gamma = 1/128 (step length grows along the ray), --cut with cut_bounds (static background rendered
    un-warped, spatial hash over +-bound), max_steps 300, T_thresh 5e-2, num_seek_IP 1, a 4:3 image."""
    from pienerf_amd.nerf.network import NeRFNetwork
    ck = scene.make_checkpoint(bound=2.0, seed=3)
    assert ck["cascade"] == 2
    W, H = 84, 63
    opt = scene.default_opt(bound=2.0, scale=0.33, dt_gamma=1.0 / 128, max_steps=300, T_thresh=5e-2, num_seek_IP=1, max_iter_num=1, cut=True,
                            cut_bounds=[-0.62, 1.0, -0.82, 0.42, -0.52, 0.28], sim_dx=0.1, W=W, H=H)
    pose = scene.orbit_pose(4.5, 25.0, -10.0)
    o, d = oracle.get_rays(pose, scene.orbit_intrinsics(W, H, 50.0), W, H)
    ref = oracle.render_deformed(o, d, deformed_ip_state, ck, opt)
    ip = deformed_ip_state
    net = NeRFNetwork(encoding="hashgrid", bound=2.0, cuda_ray=True).to(DEV).load_checkpoint_dict(ck)
    net.p_def, net.p_ori, net.IP_F, net.IP_dF, net.IP_dx = T(ip["p_def"]), T(ip["p_ori"]), T(ip["F"]), T(ip["dF"]), ip["IP_dx"]
    with torch.no_grad():
        out = net.render_deformed(T(o)[None], T(d)[None], collect_stats=True, **opt)
    st = dict(net.last_stats)
    assert ref["samples"] > 1000 and ref["trips"] >= 2
    assert st["trips"] == ref["trips"] and st["samples"] == ref["samples"] and st["err"] == 0 and st["alive_at_exit"] == 0
    assert np.abs(out["image"][0].cpu().numpy() - ref["image"]).max() < 1e-4
    assert np.abs(out["weights_sum"].cpu().numpy() - ref["weights_sum"]).max() < 1e-4
    dep, rd = out["depth"][0].cpu().numpy(), ref["depth"]
    assert np.array_equal(np.isfinite(dep), np.isfinite(rd)) and np.abs(dep[np.isfinite(dep)] - rd[np.isfinite(rd)]).max() < 1e-4


@pytest.mark.parametrize("num_seek_IP,max_iter_num,n_step,rounds", [(3, 1, 8, 1), (3, 1, 8, 1000), (2, 4, 64, 1), (1, 1, 64, 1), (3, 2, 200, 2)])
def test_march_split_between_the_two_launches_does_not_matter(deformed_ip_state, small_opt, ckpt, tail_rounds, num_seek_IP, max_iter_num, n_step,
                                                             rounds):
    """rounds = 1: every ray that needs more than one window of 8 points finishes in the wave-per-ray launch (windows of 64);
    rounds = 1000: no ray ever reaches it.  n_step = 64 / 200 marches whole rays in one call (many windows per ray)."""
    from pienerf_amd import raymarching
    ip, ck = deformed_ip_state, ckpt
    m = _march_inputs(ip, small_opt, ck, W=48)
    alive = np.nonzero(m["nears"] < 1e30)[0].astype(np.int32)
    n_alive = len(alive)
    cb = np.zeros(6, np.float32)
    args = (len(ip["p_def"]), m["n_grid"])
    ref = oracle.march_rays_quadratic_bending(*m["pig"], *args, ip["p_def"], ip["p_ori"], ip["F"], ip["dF"], max_iter_num, m["bbmin"], m["bbmax"],
                                              m["hgs"], m["res"], num_seek_IP, np.float32(ip["IP_dx"]), False, cb, n_alive, n_step, alive, m["nears"],
                                              m["o"], m["d"], 1.0, ck["density_bitfield"], ck["cascade"], ck["grid_size"], m["nears"], m["fars"], 128,
                                              False, 0.0, 1024)
    tail_rounds(rounds)
    got = raymarching.march_rays_quadratic_bending(*[T(a) for a in m["pig"]], *args, T(ip["p_def"]), T(ip["p_ori"]), T(ip["F"]), T(ip["dF"]),
                                                   max_iter_num, T(m["bbmin"]), T(m["bbmax"]), float(m["hgs"]), T(m["res"]), num_seek_IP,
                                                   float(ip["IP_dx"]), False, T(cb), n_alive, n_step, T(alive), T(m["nears"]), T(m["o"]), T(m["d"]),
                                                   1.0, T(ck["density_bitfield"]), ck["cascade"], ck["grid_size"], T(m["nears"]), T(m["fars"]), 128,
                                                   False, 0.0, 1024)
    assert (ref[2][:, 0] != 0).sum() > 500
    for name, a, b in zip(("xyzs", "dirs", "deltas"), got, ref):
        a = a.cpu().numpy()
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), f"{name}: {np.sum(a != b)} of {a.size} values differ"


@pytest.mark.parametrize("rounds", [1, 1000])
def test_frame_independent_of_tail_rounds(deformed_ip_state, small_opt, ckpt, tail_rounds, rounds):
    W = 64
    pose = scene.orbit_pose(5.0, 20.0, -15.0)
    o, d = oracle.get_rays(pose, scene.orbit_intrinsics(W, W, 50.0), W, W)
    net = _net(ckpt, deformed_ip_state)
    with torch.no_grad():
        base = net.render_deformed(T(o)[None], T(d)[None], collect_stats=True, **small_opt)
        st0 = dict(net.last_stats)
        img0, d0 = base["image"].clone(), base["depth_0"].clone()
        tail_rounds(rounds)
        alt = net.render_deformed(T(o)[None], T(d)[None], collect_stats=True, **small_opt)
        st1 = dict(net.last_stats)
    assert st0["samples"] == st1["samples"] > 2000 and st0["trips"] == st1["trips"]
    # same samples, but the sample LIST order (the order the network kernel visits them in) differs: per-ray compositing is unchanged
    assert torch.equal(alt["image"], img0) and torch.equal(alt["depth_0"], d0)


@pytest.fixture
def skip_dda():
    from pienerf_amd._lib import check, lib

    def set_dda(on):
        check(lib().pn_march_set_skip_dda(int(on)), "set_skip_dda")
    yield set_dda
    set_dda(-1)


@pytest.mark.parametrize("radius,theta,phi", [(5.0, 20.0, -15.0), (2.2, 75.0, -40.0), (9.0, -60.0, 5.0), (5.0, 0.0, 0.0), (5.0, 90.0, 0.0)])
def test_frame_independent_of_the_skip_pre_pass_form(deformed_ip_state, small_opt, ckpt, skip_dda, radius, theta, phi):
    """Trip 0's skip pre-pass with the DDA start + hop budget (pn_march_window.h: skip_empty_cells; the default) and walking hop by hop (rounds 1-2): the same
    samples (count, trips) and the same pixels bit for bit, from far, from close (rays crossing a binade of t), and along the grid axes (rays nearly
    parallel to cell faces: the cases the DDA refuses or hands on)."""
    W = 96
    pose = scene.orbit_pose(radius, theta, phi)
    o, d = oracle.get_rays(pose, scene.orbit_intrinsics(W, W, 50.0), W, W)
    net = _net(ckpt, deformed_ip_state)
    res = []
    with torch.no_grad():
        for on in (0, 1):
            skip_dda(on)
            out = net.render_deformed(T(o)[None], T(d)[None], collect_stats=True, **small_opt)
            res.append((dict(net.last_stats), out["image"].clone(), out["depth"].clone(), out["depth_0"].clone()))
    (s0, i0, d0, e0), (s1, i1, d1, e1) = res
    assert s0["samples"] == s1["samples"] > 500 and s0["trips"] == s1["trips"] and s0["err"] == s1["err"] == 0
    assert torch.equal(i0, i1) and torch.equal(e0, e1)
    assert torch.equal(torch.nan_to_num(d0, nan=-1.0), torch.nan_to_num(d1, nan=-1.0))


@pytest.mark.parametrize("bound,dt_gamma,blob_frac,radius,theta,phi", [
    (1.0, 0.0, 0.02, 3.0, 25.0, -10.0),        # one cascade, fixed step (the chair's stepping) with --cut
    (1.0, 1.0 / 64, 0.10, 0.7, 80.0, 30.0),    # camera INSIDE the volume, dense background
    (2.0, 1.0 / 128, 0.02, 4.5, 25.0, -10.0),  # the trex option set's geometry
    (2.0, 1.0 / 128, 0.002, 3.0, 0.0, 0.0),    # rays along the axes: crossings on faces, edges and corners of the regions
    (2.0, 0.0, 0.05, 1.5, 90.0, 0.0),          # two cascades with the fixed step: the mip level changes at |x| = 1 only
    (2.0, 1.0 / 16, 0.02, 5.0, -45.0, 35.0),   # large steps: dt reaches dt_max, levels from dt as well as from the position
    (4.0, 1.0 / 128, 0.01, 7.0, 30.0, 20.0),   # three cascades
    (4.0, 1.0 / 256, 0.05, 2.5, -120.0, -5.0),
])
def test_cut_frame_independent_of_the_region_skip(deformed_ip_state, small_opt, skip_dda, bound, dt_gamma, blob_frac, radius, theta, phi):
    """--cut frames with the skip pre-pass crossing the static background's empty regions on the ray's t-sequence (pn_march_window.h: region_dda; the default)
    and visiting them voxel by voxel (pn_march_set_skip_dda(0)): the same trips, the same sample count and the same pixels bit for bit — over one, two and three
    cascades, fixed and growing steps, sparse and dense backgrounds (random 8^3-voxel blocks of the bitfield on every level), a camera inside the volume and
    rays along the grid axes."""
    from pienerf_amd.nerf.network import NeRFNetwork
    ck = scene.make_checkpoint(bound=bound, seed=7)
    blobs = np.repeat(np.random.default_rng(int(1000 * blob_frac) + int(bound)).random(len(ck["density_bitfield"]) // 64) < blob_frac, 64)
    ck["density_bitfield"] = ck["density_bitfield"] | np.where(blobs, 0xFF, 0).astype(np.uint8)
    net = NeRFNetwork(encoding="hashgrid", bound=bound, cuda_ray=True).to(DEV).load_checkpoint_dict(ck)
    ip = deformed_ip_state
    net.p_def, net.p_ori, net.IP_F, net.IP_dF, net.IP_dx = T(ip["p_def"]), T(ip["p_ori"]), T(ip["F"]), T(ip["dF"]), ip["IP_dx"]
    opt = dict(small_opt, bound=bound, dt_gamma=dt_gamma, cut=True, cut_bounds=[-0.62, 1.0, -0.82, 0.42, -0.52, 0.28], max_steps=300 if dt_gamma else 1024,
               T_thresh=5e-2, num_seek_IP=1)
    W = 112
    o, d = oracle.get_rays(scene.orbit_pose(radius, theta, phi), scene.orbit_intrinsics(W, W, 50.0), W, W)
    res = []
    with torch.no_grad():
        visited = []
        net.render_deformed(T(o)[None], T(d)[None], collect_stats=True, **opt)   # (creates the workspace the counters live in)
        for on in (0, 1):
            skip_dda(on)
            net.march_counters(1)
            out = net.render_deformed(T(o)[None], T(d)[None], collect_stats=True, **opt)
            visited.append(net.march_counters(1, read=True)["iterations"])
            net.march_counters(0)
            res.append((dict(net.last_stats), out["image"].clone(), out["depth"].clone(), out["depth_0"].clone()))
    (s0, i0, d0, e0), (s1, i1, d1, e1) = res
    assert s0["samples"] == s1["samples"] > 500 and s0["trips"] == s1["trips"] and s0["err"] == s1["err"] == 0, (s0, s1)
    assert torch.equal(i0, i1) and torch.equal(e0, e1)
    assert torch.equal(torch.nan_to_num(d0, nan=-1.0), torch.nan_to_num(d1, nan=-1.0))
    print(f"bound {bound}, dt_gamma {dt_gamma:.4f}, background {blob_frac}: visited points {visited[0]} voxel by voxel, {visited[1]} with the region skip")
    assert visited[1] <= visited[0] and (blob_frac > 0.02 or visited[1] < visited[0])   # a sparse background: the region path was taken, not silently switched off


@pytest.mark.parametrize("rounds", [1, 5, 64])
@pytest.mark.parametrize("num_seek_IP,max_iter_num", [(3, 1), (2, 4), (1, 1)])
def test_frame_independent_of_the_first_trip_form(deformed_ip_state, small_opt, ckpt, rounds, num_seek_IP, max_iter_num):
    """pn_render_opts.throughput: the first trip's pass 1 with ONE lane per ray (every evaluated point a visited one) for `rounds` rounds before a ray
    goes on in the wave-per-ray windows, against the latency form (windows of 8 lattice elements): the same samples, the same pixels bit for bit."""
    W = 96
    opt = dict(small_opt, num_seek_IP=num_seek_IP, max_iter_num=max_iter_num)
    o, d = oracle.get_rays(scene.orbit_pose(3.0, 35.0, -25.0), scene.orbit_intrinsics(W, W, 50.0), W, W)
    net = _net(ckpt, deformed_ip_state)
    res = []
    with torch.no_grad():
        for thr, trips in ((0, 0), (rounds, 0), (rounds, 3)):  # the latency form; the first trip; the first three trips (throughput_trips)
            out = net.render_deformed(T(o)[None], T(d)[None], collect_stats=True, **dict(opt, march_throughput=thr, march_throughput_trips=trips))
            res.append((dict(net.last_stats), out["image"].clone(), out["depth"].clone(), out["depth_0"].clone()))
    (s0, i0, d0, e0) = res[0]
    assert s0["trips"] >= 3
    for s1, i1, d1, e1 in res[1:]:
        assert s0["samples"] == s1["samples"] > 1000 and s0["trips"] == s1["trips"] and s0["err"] == s1["err"] == 0
        assert torch.equal(i0, i1) and torch.equal(e0, e1)
        assert torch.equal(torch.nan_to_num(d0, nan=-1.0), torch.nan_to_num(d1, nan=-1.0))


@pytest.mark.parametrize("W,H", [(96, 96), (112, 72), (100, 96), (96, 98)])
def test_frame_independent_of_the_ray_tile_order(deformed_ip_state, small_opt, ckpt, W, H):
    """pn_render_opts.ray_tile_w: the alive list of a whole image starts in 16 x 4 pixel tiles instead of arange(N) (renderer.py:828).  Every ray's
    samples, composite and pixel are its own, so the frame is the same bit for bit — latency and throughput form, with trips enough that the
    compacted lists of later trips inherit the order.  Widths that are not multiples of 16 / heights that are not multiples of 4 keep the row-major order (same frame, trivially)."""
    o, d = oracle.get_rays(scene.orbit_pose(3.0, 35.0, -25.0), scene.orbit_intrinsics(W, H, 50.0), H, W)
    net = _net(ckpt, deformed_ip_state)
    opt = {k: v for k, v in small_opt.items() if k not in ("W", "H")}
    res = []
    with torch.no_grad():
        for tile_w, thr in ((0, 0), (W, 0), (W, 64), (0, 64)):
            out = net.render_deformed(T(o)[None], T(d)[None], collect_stats=True, **dict(opt, ray_tile_w=tile_w, march_throughput=thr))
            res.append((dict(net.last_stats), out["image"].clone(), out["depth"].clone(), out["depth_0"].clone(), net.trip_records()))
    (s0, i0, d0, e0, r0) = res[0]
    assert s0["trips"] >= 3 and s0["samples"] > 1000
    for s1, i1, d1, e1, r1 in res[1:]:
        assert s0["samples"] == s1["samples"] and s0["trips"] == s1["trips"] and s0["err"] == s1["err"] == 0
        assert [tuple(r[:4]) for r in r0] == [tuple(r[:4]) for r in r1]  # (n_alive, n_step, step_base, n_samples) per trip
        assert torch.equal(i0, i1) and torch.equal(e0, e1)
        assert torch.equal(torch.nan_to_num(d0, nan=-1.0), torch.nan_to_num(d1, nan=-1.0))


_FRAME_HASH_SCRIPT = r"""
import hashlib, sys, numpy as np, torch
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + "/tests")
import oracle
from conftest import make_oracle_sim
from pienerf_amd import scene
from pienerf_amd.nerf.network import NeRFNetwork
opt = scene.default_opt(sim_dx=0.1, sim_iters=4, W=48, H=48)
cloud = scene.make_chair_points(sub_res=30, hgs=opt["hash_grid_size"])
ck = scene.make_checkpoint(bound=1.0, seed=0)
s = make_oracle_sim(cloud, opt)
p_ori, _, _ = s.get_IP_info()
s.update_force(s.n_IP // 2, np.array([300.0, 100.0, -200.0]))
for _ in range(12):
    s.stepforward()
p_def, F, dF = s.get_IP_info()
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
net = NeRFNetwork(encoding="hashgrid", bound=1.0, cuda_ray=True).cuda().load_checkpoint_dict(ck)
net.p_def, net.p_ori, net.IP_F, net.IP_dF, net.IP_dx = T(p_def), T(p_ori), T(F), T(dF), s.dx * 1.05
W = 160
o, d = oracle.get_rays(scene.orbit_pose(3.0, 20.0, -15.0), scene.orbit_intrinsics(W, W, 50.0), W, W)
with torch.no_grad():
    out = net.render_deformed(T(o)[None], T(d)[None], collect_stats=True, **opt)
st = dict(net.last_stats)
h = hashlib.sha1(out["image"].cpu().numpy().tobytes() + out["depth_0"].cpu().numpy().tobytes()).hexdigest()
print("FRAME", h, st["samples"], st["trips"], st["err"])
"""


def test_composite_with_chunk_loops_equals_one_workgroup_per_chunk():
    """The fused composite / compaction with a grid far smaller than the number of 256-ray chunks (PN_CC_GRID=4: 25 rounds per workgroup on a
    160x160 frame, every round adding the words behind the workgroup's previous chunk to its prefix), on the frame's first trip too
    (PN_CC_TRIP0=1), against the default (two launches on trip 0, one workgroup per chunk afterwards): the same frame, bit for bit.  The
    knobs are read once per process, hence two processes."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    res = []
    for extra in ({}, {"PN_CC_GRID": "4", "PN_CC_TRIP0": "1"}):
        env = dict(os.environ, **extra)
        r = subprocess.run([sys.executable, "-c", _FRAME_HASH_SCRIPT % dict(root=ROOT)], env=env, capture_output=True, text=True, timeout=600)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("FRAME")]
        assert r.returncode == 0 and line, r.stderr[-2000:]
        res.append(line[0].split()[1:])
    assert res[0] == res[1] and int(res[0][1]) > 5000 and res[0][3] == "0"


def test_calc_elastic_on_adversarial_deformation_gradients():
    """pn_sim_calc_elastic (k_elastic: cyclic-Jacobi SVD with rcp/rsq + Newton instead of IEEE div/sqrt, det-+1 contract of wp.svd3, volume
    projection) on deformation gradients that decide R = U V^T: inverted (det < 0), rank 2, rank 1, zero, repeated singular values, pure
    rotations, 1e-12- and 1e+8-scaled, plus random ones — against the CPU oracle (1e-9) and, where F is non-singular, against the polar
    rotation of numpy.linalg.svd (cuda_utils.py:83-121)."""
    import ctypes as C
    from pienerf_amd._lib import check, lib, ptr, stream_ptr
    rng = np.random.default_rng(0)

    def rot(axis, ang):
        axis = np.asarray(axis, float) / np.linalg.norm(axis)
        K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
        return np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K
    Q1, Q2 = rot([1, 2, 3], 0.7), rot([-2, 1, 0.5], 2.1)
    mats = [np.eye(3) + 0.3 * rng.standard_normal((3, 3)) for _ in range(300)]
    mats += [m @ np.diag([1, 1, -1]) for m in mats[:80]]                                       # inverted elements
    mats += [Q1 @ np.diag(s) @ Q2.T for s in ([2.0, 2.0, 0.5], [1.5, 1.5, 1.5], [3.0, 1.0, 1.0], [1.0, 1.0, -1.0], [2.0, 2.0, -2.0])]  # repeated sigma
    mats += [Q1, Q2, Q1 @ Q2, np.eye(3), -np.eye(3)]                                            # rotations, identity, full inversion
    mats += [Q1 @ np.diag([2.0, 0.7, 0.0]) @ Q2.T, Q1 @ np.diag([1.3, 0.0, 0.0]) @ Q2.T, np.outer([1, 2, 3], [0.5, -1, 2.0]), np.zeros((3, 3))]  # rank 2 / 1 / 0
    mats += [1e-12 * (np.eye(3) + 0.2 * rng.standard_normal((3, 3))), 1e8 * (np.eye(3) + 0.2 * rng.standard_normal((3, 3))),
             Q1 @ np.diag([2.0, 0.5, 1e-9]) @ Q2.T, Q1 @ np.diag([1.0, 1.0 + 1e-13, 1.0 - 1e-13]) @ Q2.T]
    Fs = np.stack(mats)
    n = len(Fs)
    # one kernel per IP slot whose affine DOF rows are the identity: F[r][c] = sum_i dNx[v, i, c, 1 + r]
    topo = np.tile(np.arange(8, dtype=np.int32), (n, 1))
    dof = np.zeros((8, 10, 3))
    for j in range(3):
        dof[:, 1 + j, j] = 1.0
    dNx = np.zeros((n, 8, 3, 10))
    part = rng.uniform(0.05, 1.0, (n, 8))
    part /= part.sum(1, keepdims=True)                                                         # F split over the 8 neighbours (sums back to F)
    for r in range(3):
        for c in range(3):
            dNx[:, :, c, 1 + r] = Fs[:, r, c][:, None] * part
    RF_ref, VF_ref, FF_ref = oracle.calc_elastic(topo, dNx, dof.reshape(-1, 3))
    RF, VF, FF = (torch.empty(n, 3, 3, dtype=torch.float64, device=DEV) for _ in range(3))
    topo_d, dNx_d, dof_d = T(topo), T(dNx), T(dof.reshape(-1))   # kept alive until the kernel has run
    check(lib().pn_sim_calc_elastic(n, ptr(topo_d), ptr(dNx_d), ptr(dof_d), ptr(RF), ptr(VF), ptr(FF), stream_ptr()), "calc_elastic")
    torch.cuda.synchronize()
    RF, VF, FF = RF.cpu().numpy(), VF.cpu().numpy(), FF.cpu().numpy()
    # R is finite for every input; V F is finite exactly where the oracle's (= the reference's arithmetic) is: for F = 0 volume_invariant_project
    # divides 0 by |grad C|^2 = 0 (func_utils.py:21-40) and the reference itself produces NaN there
    assert np.all(np.isfinite(RF)) and np.array_equal(np.isfinite(VF), np.isfinite(VF_ref))
    assert np.isfinite(VF).all(axis=(1, 2)).sum() >= n - 1
    scale = np.maximum(1.0, np.abs(Fs).max(axis=(1, 2)))[:, None, None]
    assert np.abs(FF - Fs).max() / 1.0 < 1e-9 * scale.max() and np.abs((FF - Fs) / scale).max() < 1e-12   # U diag(sigma) V^T reassembles F
    dets = np.linalg.det(Fs)
    well = np.abs(dets) > 1e-6 * scale[:, 0, 0] ** 3                                           # non-singular, sigma gaps irrelevant for R
    # R is a proper rotation for EVERY input (the det-+1 contract), also the inverted / rank-deficient ones
    assert np.abs(np.einsum("nij,nkj->nik", RF, RF) - np.eye(3)).max() < 1e-9 and np.abs(np.linalg.det(RF) - 1).max() < 1e-9
    for i in np.flatnonzero(well):
        U, s, Vt = np.linalg.svd(Fs[i])
        if np.linalg.det(U @ Vt) < 0:
            U[:, -1] *= -1
        gap = s[1] - s[2] if dets[i] < 0 else 1.0                                              # an inverted F with sigma_2 = sigma_3 has no unique R
        if gap > 1e-6 * s[0]:
            assert np.abs(RF[i] - U @ Vt).max() < 1e-8, (i, np.abs(RF[i] - U @ Vt).max())
    # against the oracle wherever the answer is unique (R for well-conditioned F; V F always up to the same uniqueness)
    uniq = well.copy()
    for i in np.flatnonzero(well):
        s = np.linalg.svd(Fs[i], compute_uv=False)
        if dets[i] < 0 and s[1] - s[2] < 1e-6 * s[0]:
            uniq[i] = False
    assert np.abs(RF[uniq] - RF_ref[uniq]).max() < 1e-9
    assert np.abs((VF[uniq] - VF_ref[uniq]) / scale[uniq]).max() < 1e-9
    assert uniq.sum() > 350 and (~well).sum() >= 5
