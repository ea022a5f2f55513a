"""BASELINE configs[1] at its full size (800x800 chair, sim_dx 0.05: 640 000 rays, 3 576 IPs, 139 kernels) on the GPU box.  The CPU
oracle needs ~0.7 s per full frame on 128 cores but minutes on a small host, so parity is checked (a) against the oracle on a strided
subset of the frame's rays — rays are independent, so the subset rendered alone must reproduce the full frame's pixels — and (b) through
size-independent properties: reproducibility, agreement of the launch forms, compositing invariants, bookkeeping."""
import numpy as np
import pytest
import torch

import oracle
from conftest import make_oracle_sim, rel_err
from pienerf_amd import scene
from test_gpu_parity import DEV, T

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def full():
    from pienerf_amd.harness import SimRenderHarness
    opt = scene.default_opt()
    cloud = scene.make_chair_points(hgs=opt["hash_grid_size"])
    ckpt = scene.make_checkpoint(bound=opt["bound"], seed=0)
    h = SimRenderHarness(opt, cloud=cloud, ckpt=ckpt, device=DEV)
    h.sim.update_force(h.sim.n_IP // 2, np.array([300.0, 100.0, -200.0]))
    for _ in range(15):
        h.sim.stepforward()
    with torch.no_grad():
        out = h.step(simulate=True, collect_stats=True)   # renders the state after 15 substeps, then advances to 16
        torch.cuda.synchronize()
    return dict(h=h, opt=opt, cloud=cloud, ckpt=ckpt, out={k: v.clone() for k, v in out.items()}, stats=dict(h.model.last_stats),
                ip=dict(p_def=h.model.p_def.cpu().numpy(), p_ori=h.model.p_ori.cpu().numpy(), F=h.model.IP_F.cpu().numpy(),
                        dF=h.model.IP_dF.cpu().numpy(), IP_dx=h.model.IP_dx))


def test_full_frame_bookkeeping_and_compositing_invariants(full):
    st, out, opt = full["stats"], full["out"], full["opt"]
    N = opt["W"] * opt["H"]
    assert st["err"] == 0 and st["alive_at_exit"] == 0 and 3 <= st["trips"] <= 8
    assert 5e5 < st["samples"] < 3e6
    img = out["image"].reshape(N, 3)
    assert torch.isfinite(img).all() and float(img.min()) >= 0.0 and float(img.max()) <= 1.0 + 1e-5
    depth = out["depth"].reshape(N)
    miss = torch.isnan(depth)                       # nears == fars == FLT_MAX (renderer.py:898): the ray misses the IP box
    assert 0.3 < float(miss.float().mean()) < 0.95
    assert bool((img[miss] == 1.0).all())           # untouched rays are exactly the white background
    d0 = out["depth_0"].reshape(N)
    assert bool((d0[miss] == 0).all()) and float(d0[~miss].max()) > 3.0   # camera at radius 5, object around the origin


def test_trip_records_first_trip_lists_later_trips_are_dense(full):
    """pn_frame_trip_records: trip 0 (every ray looks for its first sample) compacts a sample list through the segmented append lists; every
    later trip runs the network over all n_alive x n_step slots and only counts what was really emitted (frame driver)."""
    h, opt = full["h"], full["opt"]
    N = opt["W"] * opt["H"]
    with torch.no_grad():
        h.step(simulate=False, collect_stats=True)
    st = dict(h.model.last_stats)
    rec = h.model.trip_records()
    assert len(rec) == st["trips"]
    n_alive, n_step, step_base, n_list, n_emitted, n_tail = rec[0]
    assert (n_alive, n_step, step_base, n_emitted) == (N, 1, 0, -1) and 0 < n_list < N and 0 < n_tail < N
    samples, base = n_list, 1
    for (na, ns, sb, nl, ne, nt), prev in zip(rec[1:], rec[:-1]):
        assert 0 < na <= prev[0] and ns == max(min(N // na, 8), 1) and sb == base      # renderer.py:841, :891
        assert nl == na * ns and 0 <= ne <= nl and 0 <= nt <= na                         # dense: identity list; emitted <= slots
        samples += ne
        base += ns
    assert samples == st["samples"]
    assert rec[1][0] <= n_list                                                          # only rays that found a sample can be alive next


def test_full_frame_is_reproducible_and_launch_forms_agree(full):
    """Same state -> same bits, eager twice; the captured graph and the pipelined lanes reproduce the eager images of a fresh harness."""
    from pienerf_amd.harness import SimRenderHarness
    h = full["h"]
    with torch.no_grad():
        a = h.step(simulate=False)["image"].clone()
        b = h.step(simulate=False)["image"].clone()
    assert torch.equal(a, b)
    opt, cloud, ckpt = full["opt"], full["cloud"], full["ckpt"]
    eager = SimRenderHarness(opt, cloud=cloud, ckpt=ckpt, device=DEV)
    pipe = SimRenderHarness(opt, cloud=cloud, ckpt=ckpt, device=DEV).capture_pipelined(lanes=3, n_trips=8)
    with torch.no_grad():
        want = [eager.step()["image"].clone() for _ in range(5)]
        eager.synchronize()
        got = []
        for f in range(5):
            got += [(i, r["image"].copy()) for i, r in pipe.step_pipelined()]
        got += [(i, r["image"].copy()) for i, r in pipe.drain_pipeline()]
    assert [g[0] for g in got] == list(range(5))
    for f in range(5):  # initialisation and every kernel are order-deterministic: bit-identical (compared through the pinned host copy)
        assert np.array_equal(got[f][1], want[f][0].cpu().numpy()), f
    assert (want[0] - want[3]).abs().max() > 1e-4   # gravity moved the chair between the frames


def test_full_frame_independent_of_the_skip_pre_pass_form(full):
    """All 640 000 rays with the skip pre-pass's DDA start + hop budget (the default) and walking hop by hop: the same trip records (alive rays, list
    lengths, emitted samples per trip) and the same pixels bit for bit."""
    from pienerf_amd._lib import check, lib
    h = full["h"]
    res = []
    try:
        with torch.no_grad():
            for on in (0, 1):
                check(lib().pn_march_set_skip_dda(on), "set_skip_dda")
                out = h.step(simulate=False, collect_stats=True)
                res.append((dict(h.model.last_stats), h.model.trip_records(), {k: out[k].clone() for k in ("image", "depth", "depth_0")}))
    finally:
        check(lib().pn_march_set_skip_dda(-1), "set_skip_dda")
    (s0, r0, o0), (s1, r1, o1) = res
    assert s0["samples"] == s1["samples"] and s0["trips"] == s1["trips"] and s0["err"] == s1["err"] == 0
    assert [r[:5] for r in r0] == [r[:5] for r in r1]   # (the tail-list length of trip 0 may differ: where the pre-pass stops is not an output)
    for k in o0:
        assert torch.equal(torch.nan_to_num(o0[k], nan=-1.0), torch.nan_to_num(o1[k], nan=-1.0)), k


def test_full_frame_independent_of_the_first_trip_form(full):
    """All 640 000 rays with the first trip's pass 1 in its throughput form (one lane per ray, 64 rounds: what the pipelined harness renders with)
    and in the latency form: the same trip records and the same pixels bit for bit."""
    h = full["h"]
    m = h.model
    res = []
    with torch.no_grad():
        for thr, trips in ((0, 0), (64, 0), (64, 3)):  # latency form; first trip (the chair's pipelined form); three trips (throughput_trips)
            rays = {k: full["out"][k] for k in ("rays_o", "rays_d")}
            out = m.render_deformed(rays["rays_o"], rays["rays_d"], staged=True, bg_color=None, perturb=False, collect_stats=True,
                                    **dict(h.render_kwargs(), march_throughput=thr, march_throughput_trips=trips))
            res.append((dict(m.last_stats), m.trip_records(), {k: out[k].clone() for k in ("image", "depth", "depth_0")}))
    s0, r0, o0 = res[0]
    for s1, r1, o1 in res[1:]:
        assert s0["samples"] == s1["samples"] > 5e5 and s0["trips"] == s1["trips"] and s0["err"] == s1["err"] == 0
        assert [r[:5] for r in r0] == [r[:5] for r in r1]
        for k in o0:
            assert torch.equal(torch.nan_to_num(o0[k], nan=-1.0), torch.nan_to_num(o1[k], nan=-1.0)), k


def test_strided_subset_of_the_full_frame_matches_the_oracle(full):
    """Every 199th ray of the 800x800 frame, rendered by the CPU oracle from the same IP state, against the full frame's pixels;
    and the same subset rendered alone on the GPU reproduces the full frame's pixels bit for bit (rays are independent)."""
    opt, out, ip = full["opt"], full["out"], full["ip"]
    N = opt["W"] * opt["H"]
    sel = np.arange(7, N, 199)
    o = out["rays_o"].reshape(N, 3)[sel].cpu().numpy()
    d = out["rays_d"].reshape(N, 3)[sel].cpu().numpy()
    ref = oracle.render_deformed(o, d, ip, full["ckpt"], opt)
    img = out["image"].reshape(N, 3)[sel].cpu().numpy()
    assert ref["samples"] > 2000
    assert np.abs(img - ref["image"]).max() < 1e-4
    dep, rd = out["depth"].reshape(N)[sel].cpu().numpy(), ref["depth"]
    assert np.array_equal(np.isfinite(dep), np.isfinite(rd)) and np.abs(dep[np.isfinite(dep)] - rd[np.isfinite(rd)]).max() < 1e-4
    m = full["h"].model
    with torch.no_grad():
        sub = m.render_deformed(T(o)[None], T(d)[None], staged=True, bg_color=None, perturb=False, **full["h"].render_kwargs())
    assert np.array_equal(sub["image"].reshape(-1, 3).cpu().numpy(), img)
    ws = sub["weights_sum"].cpu().numpy()
    assert ws.min() >= 0.0 and ws.max() <= 1.0 + 1e-5 and np.abs(ws - ref["weights_sum"]).max() < 1e-4


def test_full_trex_frame_independent_of_the_skip_pre_pass_form():
    """BASELINE configs[2] as bench.py builds it (1008 x 756, bound 2 with two cascades, --cut, dt_gamma 1 / 128, static background in 2 % of the density
    blocks, deformed by 12 substeps under the bench's force): the frame with the skip pre-pass crossing the static background's empty regions on the
    ray's t-sequence (pn_march_window.h: region_dda, the default) and visiting them voxel by voxel (pn_march_set_skip_dda(0)) — the same trip records,
    the same number of marched samples, the same pixels bit for bit; from three camera poses (rays along different axes, grazing the volume's faces)."""
    from pienerf_amd._lib import check, lib
    from pienerf_amd.harness import SimRenderHarness
    opt = scene.trex_opt(radius=4.5)
    cloud = scene.make_chair_points(hgs=opt["hash_grid_size"], bound=opt["bound"])
    ckpt = scene.make_checkpoint(bound=2.0, seed=3)
    blobs = np.repeat(np.random.default_rng(5).random(len(ckpt["density_bitfield"]) // 64) < 0.02, 64)
    ckpt["density_bitfield"] = ckpt["density_bitfield"] | np.where(blobs, 0xFF, 0).astype(np.uint8)
    h = SimRenderHarness(opt, cloud=cloud, ckpt=ckpt, device=DEV, overlap_sim=False)
    h.sim.update_force(h.sim.n_IP // 2, np.array([250.0, 120.0, -180.0]))
    for _ in range(12):
        h.sim.stepforward()
    try:
        for pose in (scene.orbit_pose(4.5, 25.0, -10.0), scene.orbit_pose(3.0, 90.0, 0.0), scene.orbit_pose(6.0, -45.0, 35.0)):
            h.pose = pose
            res = []
            with torch.no_grad():
                for on in (0, 1):
                    check(lib().pn_march_set_skip_dda(on), "set_skip_dda")
                    out = h.step(simulate=False, collect_stats=True)
                    torch.cuda.synchronize()
                    res.append((dict(h.model.last_stats), h.model.trip_records(), {k: out[k].clone() for k in ("image", "depth", "depth_0")}))
            (s0, r0, o0), (s1, r1, o1) = res
            assert s0["samples"] == s1["samples"] > 100000 and s0["trips"] == s1["trips"] and s0["err"] == s1["err"] == 0, (s0, s1)
            assert [r[:5] for r in r0] == [r[:5] for r in r1]
            for k in o0:
                assert torch.equal(torch.nan_to_num(o0[k], nan=-1.0), torch.nan_to_num(o1[k], nan=-1.0)), k
    finally:
        check(lib().pn_march_set_skip_dda(-1), "set_skip_dda")


def test_full_size_substeps_match_the_oracle(full):
    """Three substeps of the 139-kernel / 3 576-IP system against the fp64 oracle (fresh simulators, same load)."""
    from pienerf_amd.harness import SimRenderHarness
    opt, cloud = full["opt"], full["cloud"]
    h = SimRenderHarness(opt, cloud=cloud, ckpt=full["ckpt"], device=DEV)
    ref = make_oracle_sim(cloud, opt)
    assert (h.sim.n_k, h.sim.n_IP) == (ref.n_k, ref.n_IP) == (139, 3576)
    f = np.array([300.0, 100.0, -200.0])
    h.sim.update_force(h.sim.n_IP // 2, f)
    ref.update_force(ref.n_IP // 2, f)
    for step in range(3):
        h.sim.stepforward()
        ref.stepforward()
        disp = h.sim.dof.cpu().numpy().reshape(-1, 3) - ref.dof_rest
        assert rel_err(disp, ref.dof - ref.dof_rest) < 1e-6, step
    p1, F1, dF1 = (t.cpu().numpy() for t in h.sim.get_IP_info())
    p2, F2, dF2 = ref.get_IP_info()
    assert np.abs(p1 - p2).max() < 1e-5 and np.abs(F1 - F2).max() < 1e-4


def test_stress_configuration_exactly_as_baseline_states_it():
    """BASELINE configs[4] in ONE run, every clause of it: the sub_res = 180 cloud (268 k points feeding the same 0.05 simulation grid), fp16 tables +
    fp16-MFMA network (the autocast path: gridencoder/grid.py:43-44), 4096-ray batches (max_ray_batch: renderer.py:565-576), max_iter_num = 5
    Newton iterations, num_seek_IP = 3, 800x800.
      (a) the frame in ray batches (pn_render_opts.ray_batch = 4096: 157 batches with their own trip schedules inside the same launches) equals
          the batches rendered one after the other — image and depth bit for bit, marched samples and trips as integers — and equals the frame
          rendered in one piece bit for bit (compositing does not depend on the schedule);
      (b) the captured, pipelined staged form (harness.capture_staged) returns those same frames;
      (c) a strided ray subset against oracle.render_deformed under oracle.half_precision(): sigma may differ by half an ulp of its half logit on
          isolated samples, which can move a ray's T_thresh exit by one sample — tolerance: 1e-2 worst pixel, 5e-4 mean (the bars of
          tests/test_gpu_half.py at test size); against the fp32 oracle the frame must differ by more than 1e-4 (fp16 really in effect)."""
    from pienerf_amd.harness import SimRenderHarness
    from test_gpu_parity import _batches_one_after_the_other
    opt = scene.default_opt(max_iter_num=5, num_seek_IP=3, fp16=True, max_ray_batch=4096)
    cloud = scene.make_chair_points(sub_res=180, hgs=opt["hash_grid_size"])
    assert len(cloud["pos"]) > 250000
    ckpt = scene.make_checkpoint(bound=opt["bound"], seed=0)
    force = np.array([400.0, -150.0, 250.0])
    whole = SimRenderHarness(opt, cloud=cloud, ckpt=ckpt, device=DEV, overlap_sim=False)
    staged = SimRenderHarness(dict(opt, ray_batch=4096), cloud=cloud, ckpt=ckpt, device=DEV, overlap_sim=False)
    assert whole.sim.n_IP > 3000
    for h in (whole, staged):
        h.sim.update_force(h.sim.n_IP // 2, force)
        for _ in range(12):
            h.sim.stepforward()
    with torch.no_grad():
        out = whole.step(simulate=True, collect_stats=True)
        st_w = dict(whole.model.last_stats)
        got = staged.step(simulate=True, collect_stats=True)
        st_s = dict(staged.model.last_stats)
        torch.cuda.synchronize()
        assert st_w["err"] == 0 and st_w["alive_at_exit"] == 0 and st_w["samples"] > 500000
        assert st_s["err"] == 0 and st_s["alive_at_exit"] == 0
        m = whole.model
        disp = (m.p_def - m.p_ori).abs().max().item()
        assert 1e-3 < disp < 0.3                                     # visibly deformed, not blown up
        img, dep, total, trips = _batches_one_after_the_other(whole, out, 4096)
    N = img.shape[0]
    assert N == 640000 and -(-N // 4096) == 157
    # (a)
    assert st_s["samples"] == total and st_s["trips"] == trips, (st_s, total, trips)
    assert 0.95 * st_w["samples"] < total < 1.1 * st_w["samples"] and total != st_w["samples"] and trips >= st_w["trips"]
    assert torch.equal(got["image"].reshape(-1, 3), img) and torch.equal(got["depth_0"].reshape(-1), dep)
    assert torch.equal(got["image"], out["image"]) and torch.equal(got["depth_0"], out["depth_0"])
    # (b)
    with torch.no_grad():
        eager = [staged.to_host(staged.step()) for _ in range(3)]
        staged.synchronize()
    pipe = SimRenderHarness(opt, cloud=cloud, ckpt=ckpt, device=DEV)
    pipe.sim.update_force(pipe.sim.n_IP // 2, force)
    for _ in range(13):
        pipe.sim.stepforward()
    pipe.synchronize()
    pipe.capture_staged(lanes=2, depth=2, n_trips=None)
    assert pipe._pipe_backend.kw["ray_batch"] == 4096 and "ray_batch" not in pipe.opt   # an option of this pipeline's renders, not of the harness
    res = []
    for _ in range(3):
        res += [(i, r["image"].copy()) for i, r in pipe.step_pipelined()]
    res += [(i, r["image"].copy()) for i, r in pipe.drain_pipeline()]
    assert [i for i, _ in res] == [0, 1, 2]
    for f in range(3):
        assert np.array_equal(res[f][1], eager[f]["image"]), f
    # (c)
    sel = np.arange(97, N, 211)
    o, d = out["rays_o"][0], out["rays_d"][0]
    ip = dict(p_def=m.p_def.cpu().numpy(), p_ori=m.p_ori.cpu().numpy(), F=m.IP_F.cpu().numpy(), dF=m.IP_dF.cpu().numpy(), IP_dx=m.IP_dx)
    with oracle.half_precision():
        ref = oracle.render_deformed(o.cpu().numpy()[sel], d.cpu().numpy()[sel], ip, ckpt, opt)
    ref32 = oracle.render_deformed(o.cpu().numpy()[sel], d.cpu().numpy()[sel], ip, ckpt, opt)
    assert ref["samples"] > 2000
    px = got["image"].reshape(-1, 3)[sel].cpu().numpy()
    err = np.abs(px - ref["image"])
    print(f"configs[4] subset vs half oracle: max {err.max():.2e}, mean {err.mean():.2e}; vs fp32 oracle: max {np.abs(px - ref32['image']).max():.2e}")
    assert err.max() < 1e-2 and err.mean() < 5e-4
    assert np.abs(px - ref32["image"]).max() > 1e-4 and np.abs(px - ref32["image"]).mean() < 2e-3


def test_trex_configuration_full_size():
    """BASELINE configs[2] at its full size: 1008x756 (762 048 rays), bound 2 (two density cascades, 4096-resolution hash grid),
    dt_gamma 1/128, --cut with the README's cut_bounds (samples outside are rendered un-warped as static background), max_steps 300,
    T_thresh 5e-2, num_seek_IP 1, max_iter_num 1, sim_dx 0.05 (README.md:134 of the reference; synthetic assets, orbit pose).  Parity through
    the oracle on a strided ray subset, the subset rendered alone == the full frame's pixels bit for bit, and compositing invariants."""
    from pienerf_amd.harness import SimRenderHarness
    W, H = 1008, 756
    opt = scene.default_opt(bound=2.0, scale=0.33, dt_gamma=1.0 / 128, max_steps=300, T_thresh=5e-2, num_seek_IP=1, max_iter_num=1, cut=True,
                            cut_bounds=[-0.62, 1.0, -0.82, 0.42, -0.52, 0.28], sim_dx=0.05, W=W, H=H, radius=4.5)
    cloud = scene.make_chair_points(hgs=opt["hash_grid_size"], bound=opt["bound"])
    ckpt = scene.make_checkpoint(bound=2.0, seed=3)
    assert ckpt["cascade"] == 2
    h = SimRenderHarness(opt, cloud=cloud, ckpt=ckpt, device=DEV)
    h.sim.update_force(h.sim.n_IP // 2, np.array([250.0, 120.0, -180.0]))
    for _ in range(10):
        h.sim.stepforward()
    pose = scene.orbit_pose(4.5, 25.0, -10.0)
    with torch.no_grad():
        out = h.step(pose=pose, intrinsics=scene.orbit_intrinsics(W, H, 50.0), W=W, H=H, simulate=True, collect_stats=True)
        torch.cuda.synchronize()
    st = dict(h.model.last_stats)
    N = W * H
    assert out["image"].shape == (1, H, W, 3) and st["err"] == 0 and st["alive_at_exit"] == 0 and st["samples"] > 100000 and st["trips"] >= 2
    m = h.model
    ip = dict(p_def=m.p_def.cpu().numpy(), p_ori=m.p_ori.cpu().numpy(), F=m.IP_F.cpu().numpy(), dF=m.IP_dF.cpu().numpy(), IP_dx=m.IP_dx)
    assert 1e-3 < np.abs(ip["p_def"] - ip["p_ori"]).max() < 0.3
    sel = np.arange(11, N, 251)
    o, d = out["rays_o"].reshape(N, 3)[sel].cpu().numpy(), out["rays_d"].reshape(N, 3)[sel].cpu().numpy()
    ref = oracle.render_deformed(o, d, ip, ckpt, opt)
    img = out["image"].reshape(N, 3)[sel].cpu().numpy()
    assert ref["samples"] > 500
    assert np.abs(img - ref["image"]).max() < 1e-4
    with torch.no_grad():
        sub = m.render_deformed(T(o)[None], T(d)[None], staged=True, bg_color=None, perturb=False, frame_slot=1, **h.render_kwargs())
    assert np.array_equal(sub["image"].reshape(-1, 3).cpu().numpy(), img)
    ws = sub["weights_sum"].cpu().numpy()
    assert ws.min() >= 0.0 and ws.max() <= 1.0 + 1e-5 and np.abs(ws - ref["weights_sum"]).max() < 1e-4
    full_img = out["image"].reshape(N, 3)
    assert float(full_img.min()) >= 0.0 and float(full_img.max()) <= 1.0 + 1e-5
    # two trips with one lane per ray (pn_render_opts.throughput / throughput_trips: what the pipelined harness picks when a later trip still has
    # >= 128 k alive rays, as with bench.py's static background) — the same trip records and pixels as the latency form above, bit for bit
    rec0 = m.trip_records()
    assert len(rec0) >= 2 and rec0[1][0] > 10000
    with torch.no_grad():
        thr = m.render_deformed(out["rays_o"], out["rays_d"], staged=True, bg_color=None, perturb=False, collect_stats=True,
                                **dict(h.render_kwargs(), march_throughput=64, march_throughput_trips=2))
    assert dict(m.last_stats)["samples"] == st["samples"] and [r[:5] for r in m.trip_records()] == [r[:5] for r in rec0]
    for k in ("image", "depth", "depth_0"):
        assert torch.equal(torch.nan_to_num(thr[k].reshape(-1), nan=-1.0), torch.nan_to_num(out[k].reshape(-1), nan=-1.0)), k


def test_full_kernel_grid_scene():
    """The largest simulator the chair options allow: a solid block filling the box -> all 7^3 = 343 GMLS kernels (10 290 DOFs, the SURVEY §8
    upper bound), 39 k integration points (11x the chair), 314 k cloud points.  No oracle at this size (its dense CPU initialisation takes
    minutes); size-independent properties instead: finite bounded motion, no device error flags, staged ray batches == the one-shot frame
    bit for bit, two harnesses built from scratch agree bit for bit."""
    from pienerf_amd.harness import SimRenderHarness
    boxes = np.array([(-0.85, 0.85, -0.85, 0.85, -0.85, 0.85)])
    opt = scene.default_opt(W=320, H=320)
    cloud = scene.make_chair_points(sub_res=80, hgs=opt["hash_grid_size"], boxes=boxes)
    ck = scene.make_checkpoint(bound=1.0, seed=0, solid=lambda p, margin=0.0: scene.chair_solid(p, margin, boxes=boxes))
    outs = []
    for _ in range(2):
        h = SimRenderHarness(opt, cloud=cloud, ckpt=ck, device=DEV)
        assert h.sim.n_k == 343 and h.sim.n_IP > 35000
        h.sim.update_force(h.sim.n_IP // 2, np.array([3000.0, 1000.0, -2000.0]))
        with torch.no_grad():
            for _ in range(4):
                out = h.step(collect_stats=True)
            torch.cuda.synchronize()
        st = dict(h.model.last_stats)
        assert st["err"] == 0 and st["alive_at_exit"] == 0 and st["samples"] > 100000
        assert bool(torch.isfinite(h.sim.dof).all())
        disp = (h.model.p_def - h.model.p_ori).abs().max().item()
        assert 1e-4 < disp < 0.2
        outs.append((out["image"].clone(), h.sim.dof.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    with torch.no_grad():
        o, d = out["rays_o"][0], out["rays_d"][0]
        full = h.step(simulate=False)
        img = torch.empty_like(full["image"].reshape(-1, 3))
        for head in range(0, o.shape[0], 30000):
            r = h.model.render_deformed(o[None, head:head + 30000], d[None, head:head + 30000], frame_slot=1, **h.render_kwargs())
            img[head:head + 30000] = r["image"][0]
    assert torch.equal(img, full["image"].reshape(-1, 3))
    assert 0.0 <= float(img.min()) and float(img.max()) <= 1.0 + 1e-5 and float(img.min()) < 0.9
