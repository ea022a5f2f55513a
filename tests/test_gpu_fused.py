"""The later loop trips as ONE persistent launch (pn_render_opts.fused_from, csrc/pn_trips_fused.h) against the trip-by-trip launches of the same
frame (fused_from = -1): the reference's loop (nerf/renderer.py:836-891) couples the rays of a trip only through n_step = max(min(N // n_alive, 8), 1),
which is 8 for the rest of the frame once n_alive <= N / 8 — from there every ray loops { march 8; network; composite } on its own.  Same samples, same
trip records, same pixels BIT FOR BIT, in every build of the kernel (num_seek_IP 1 / 2 / 3, one or several Newton iterations, fp32 / fp16 network), in
--cut mode with two cascades, when max_steps ends the frame, when the launch does not apply at its first trip, inside captured graphs and pipelines."""
import numpy as np
import pytest
import torch

import oracle
from pienerf_amd import scene
from test_gpu_edges import _net
from test_gpu_parity import DEV, T

pytestmark = pytest.mark.gpu


def _both_forms(net, o, d, opt, amp=False, **kw):
    res = []
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16, enabled=amp):
        for fused_from in (-1, 0):   # trip by trip / the whole frame in the fused launch where that applies, else from the first trip with n_step == 8
            out = net.render_deformed(T(o)[None], T(d)[None], collect_stats=True, **dict(opt, fused_from=fused_from, fused_whole=True, **kw))
            res.append((dict(net.last_stats), net.trip_records(max_trips=140), {k: out[k].clone() for k in ("image", "depth", "depth_0", "weights_sum")},
                        net.fused_clocks()["first_trip"]))
    return res


def _assert_same(a, b):
    (sa, ra, oa, fa), (sb, rb, ob, fb) = a, b
    assert fa == -1 and fb >= 0, (fa, fb)        # the second render did switch to the fused launch (0: for the whole frame)
    assert sa == sb and sa["err"] == 0 and sa["alive_at_exit"] == 0, (sa, sb)
    # (n_alive, n_step, step_base, n_samples, n_emitted, rays through the 64-lane windows) per trip; the last one is a diagnostic of the FORM of a frame's
    # first trip (windows of 8 then 64 / one lane per ray then 64: how many rays outlast the first form)
    assert ra[0][:5] == rb[0][:5] and ra[1:] == rb[1:], (ra, rb)
    for k in ("image", "depth_0", "weights_sum"):
        assert torch.equal(oa[k], ob[k]), k
    assert torch.equal(torch.nan_to_num(oa["depth"], nan=-1.0), torch.nan_to_num(ob["depth"], nan=-1.0))


@pytest.mark.parametrize("num_seek_IP,max_iter_num,fp16", [(3, 1, False), (2, 1, False), (1, 1, False), (3, 5, False), (2, 3, True), (3, 1, True), (1, 2, False)])
def test_fused_trips_equal_the_trip_by_trip_frame(deformed_ip_state, small_opt, ckpt, num_seek_IP, max_iter_num, fp16):
    W = 128
    opt = dict(small_opt, W=W, H=W, num_seek_IP=num_seek_IP, max_iter_num=max_iter_num)
    o, d = oracle.get_rays(scene.orbit_pose(5.0, 35.0, -25.0), scene.orbit_intrinsics(W, W, 50.0), W, W)
    net = _net(ckpt, deformed_ip_state)
    a, b = _both_forms(net, o, d, opt, amp=fp16)
    assert a[0]["trips"] >= 4 and a[0]["samples"] > 3000, a[0]
    assert a[1][1][1] == 8  # the second trip marches 8 samples per ray: the fused launch starts there
    _assert_same(a, b)
    assert b[3] == 0        # ... and, few enough rays meeting the object, it took the whole frame, first trip included
    # the first trip as per-trip launches, the rest fused (fused_from = 0 without fused_whole): the same frame again
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16, enabled=fp16):
        out = net.render_deformed(T(o)[None], T(d)[None], collect_stats=True, **dict(opt, fused_from=0))
    assert net.fused_clocks()["first_trip"] == 1 and dict(net.last_stats) == a[0] and net.trip_records(max_trips=140) == a[1]
    for k in ("image", "depth_0", "weights_sum"):
        assert torch.equal(out[k], a[2][k]), k
    # the first trip's march as its own launches, its network / composite / compaction inside the fused launch (fused_fold): the same frame again, trip
    # records included (the rays the first trip handed to the tail pass are counted by the march itself)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16, enabled=fp16):
        out = net.render_deformed(T(o)[None], T(d)[None], collect_stats=True, **dict(opt, fused_from=0, fused_fold=True))
    assert net.fused_clocks()["mode"] == 2 and dict(net.last_stats) == a[0] and net.trip_records(max_trips=140) == a[1]
    for k in ("image", "depth_0", "weights_sum"):
        assert torch.equal(out[k], a[2][k]), k


@pytest.mark.parametrize("pose", [(5.0, 20.0, -15.0), (2.2, 75.0, -40.0), (9.0, -60.0, 5.0)])
def test_fused_trips_from_near_and_far(deformed_ip_state, small_opt, ckpt, pose):
    W = 96
    o, d = oracle.get_rays(scene.orbit_pose(*pose), scene.orbit_intrinsics(W, W, 50.0), W, W)
    net = _net(ckpt, deformed_ip_state)
    a, b = _both_forms(net, o, d, dict(small_opt, W=W, H=W), march_throughput=64)
    assert a[0]["samples"] > 300
    if a[1][1][1] == 8:
        _assert_same(a, b)
    else:   # close up more than an eighth of the rays hit: the launch steps aside until n_step reaches 8 (next test)
        assert a[0] == b[0] and a[1] == b[1] and torch.equal(a[2]["image"], b[2]["image"])


def test_fused_launch_steps_aside_while_n_step_is_below_8(deformed_ip_state, small_opt, ckpt):
    """A close-up: more than N / 8 rays alive after the first trip, so trip 1 has n_step < 8 and the launch does not apply there.  The blocking driver
    runs that trip as per-trip launches and tries again; a fixed-trip (async) render is left unfinished and finished by render_continue."""
    W = 64
    opt = dict(small_opt, W=W, H=W)
    o, d = oracle.get_rays(scene.orbit_pose(1.6, 35.0, -25.0), scene.orbit_intrinsics(W, W, 50.0), W, W)
    net = _net(ckpt, deformed_ip_state)
    a, b = _both_forms(net, o, d, opt)
    assert a[1][1][1] < 8 and a[0]["trips"] >= 3, a[1]
    assert b[3] >= 2                                 # the fused launch took over at a later trip
    assert a[0] == b[0] and a[1] == b[1]
    for k in ("image", "depth_0", "weights_sum"):
        assert torch.equal(a[2][k], b[2][k]), k
    with torch.no_grad():   # ... and the folded first trip steps aside the same way: its march has run, the per-trip launches go on behind it
        out = net.render_deformed(T(o)[None], T(d)[None], collect_stats=True, **dict(opt, fused_from=0, fused_fold=True))
    assert dict(net.last_stats) == a[0] and net.trip_records(max_trips=140) == a[1] and torch.equal(out["image"], a[2]["image"])
    # async: the fused launch finds too many rays with something to march (fused_whole: the whole frame) / n_step < 8 behind one per-trip trip and
    # leaves the frame as it is
    for whole, fold, trips_done in ((True, False, 0), (False, False, 1), (False, True, 0)):
        with torch.no_grad():
            out = net.render_deformed(T(o)[None], T(d)[None], async_trips=8, **dict(opt, fused_from=0, fused_whole=whole, fused_fold=fold))
            st = net.render_status()
            assert st["alive_at_exit"] > 0 and st["trips"] == trips_done, st
            net.render_continue(0, T(o)[None], T(d)[None], out, **dict(opt, fused_from=0, fused_whole=whole, fused_fold=fold))
            assert net.last_stats["alive_at_exit"] == 0 and net.last_stats["samples"] == a[0]["samples"]
        assert torch.equal(out["image"], a[2]["image"]) and torch.equal(out["weights_sum"], a[2]["weights_sum"])


def test_fused_trips_end_at_max_steps(deformed_ip_state, small_opt, ckpt):
    """renderer.py:836: `while step < max_steps` — with max_steps = 20 the loop ends behind the trip that takes step to 25 whatever is alive."""
    W = 96
    o, d = oracle.get_rays(scene.orbit_pose(5.0, 35.0, -25.0), scene.orbit_intrinsics(W, W, 50.0), W, W)
    ck = dict(ckpt)
    net = _net(ck, deformed_ip_state)
    net.density_scale = 0.05    # thin medium: rays stay alive for many trips
    opt = dict(small_opt, W=W, H=W, max_steps=20)
    a, b = _both_forms(net, o, d, opt)
    recs = a[1]
    assert recs[-1][2] + recs[-1][1] >= 20 and len(recs) == 4, recs   # trips of 1 + 8 + 8 + 8 steps
    _assert_same(a, b)


def test_fused_trips_in_cut_mode_with_two_cascades(deformed_ip_state):
    """The trex option set (README.md:134): bound 2, dt_gamma 1/128, --cut, max_steps 300, T_thresh 5e-2, num_seek_IP 1, static background samples."""
    from pienerf_amd.nerf.network import NeRFNetwork
    ck = scene.make_checkpoint(bound=2.0, seed=3)
    blobs = np.repeat(np.random.default_rng(5).random(len(ck["density_bitfield"]) // 64) < 0.02, 64)
    ck["density_bitfield"] = ck["density_bitfield"] | np.where(blobs, 0xFF, 0).astype(np.uint8)
    W, H = 112, 84
    opt = scene.default_opt(bound=2.0, scale=0.33, dt_gamma=1.0 / 128, max_steps=300, T_thresh=5e-2, num_seek_IP=1, max_iter_num=1, cut=True,
                            cut_bounds=[-0.62, 1.0, -0.82, 0.42, -0.52, 0.28], sim_dx=0.1, W=W, H=H)
    o, d = oracle.get_rays(scene.orbit_pose(4.5, 25.0, -10.0), scene.orbit_intrinsics(W, H, 50.0), W, H)
    ip = deformed_ip_state
    net = NeRFNetwork(encoding="hashgrid", bound=2.0, cuda_ray=True).to(DEV).load_checkpoint_dict(ck)
    net.p_def, net.p_ori, net.IP_F, net.IP_dF, net.IP_dx = T(ip["p_def"]), T(ip["p_ori"]), T(ip["F"]), T(ip["dF"]), ip["IP_dx"]
    a, b = _both_forms(net, o, d, opt)
    assert a[0]["samples"] > 1000 and a[0]["trips"] >= 2
    assert b[3] >= 1
    assert a[0] == b[0] and a[1] == b[1]
    for k in ("image", "depth_0", "weights_sum"):
        assert torch.equal(a[2][k], b[2][k]), k


def test_fused_trips_in_graphs_and_pipelines(small_cloud, small_opt, ckpt):
    """The pipelined harness picks fused_from off a blocking frame; its frames equal a harness that never fuses, bit for bit, and no frame is continued."""
    from pienerf_amd.harness import SimRenderHarness
    opt = dict(small_opt, W=128, H=128)
    poses = [scene.orbit_pose(5.0, 30.0 + 4.0 * f, -20.0) for f in range(7)]
    res = {}
    for name, kw in (("classic", dict(fused_from=-1)), ("fused", dict())):   # two render lanes: the pipeline picks the whole-frame launch
        h = SimRenderHarness(opt, cloud=small_cloud, ckpt=ckpt, device=DEV)
        h.sim.update_force(h.sim.n_IP // 2, np.array([300.0, 100.0, -200.0]))
        h.capture_pipelined(lanes=2, depth=2, n_trips=None, render_kw=kw)
        got = []
        for p in poses:
            got += [(i, {k: r[k].copy() for k in ("image", "depth_0")}) for i, r in h.step_pipelined(pose=p)]
        got += [(i, {k: r[k].copy() for k in ("image", "depth_0")}) for i, r in h.drain_pipeline()]
        res[name] = (got, h._pipe_backend.kw["fused_from"], h._pipe_backend.continued)
    assert res["fused"][1] == 0 and res["classic"][1] == -1   # the pipeline's frames are whole-frame launches
    assert res["fused"][2] == 0
    assert [i for i, _ in res["fused"][0]] == list(range(7))
    for (i, a), (_, b) in zip(res["classic"][0], res["fused"][0]):
        assert np.array_equal(a["image"], b["image"]) and np.array_equal(a["depth_0"], b["depth_0"]), i
