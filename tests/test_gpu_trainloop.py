"""NeRFRenderer.run_cuda's training branch, update_extra_state / mark_untrained_grid and a short training run on
the HIP ops (pienerf_amd/training.py): a fresh network is fitted to images rendered from the synthetic chair checkpoint."""
import numpy as np
import pytest
import torch

import oracle
from oracle import training as otr
from pienerf_amd import raymarching, scene
from pienerf_amd.nerf.network import NeRFNetwork
from pienerf_amd.training import RayImageSet, Trainer
from test_gpu_parity import DEV, T

pytestmark = pytest.mark.gpu


def _teacher():
    ck = scene.make_checkpoint(bound=1.0, seed=0, shaped=True)
    return ck, NeRFNetwork(encoding="hashgrid", bound=1.0, cuda_ray=True).to(DEV).load_checkpoint_dict(ck)


def test_run_cuda_training_branch_matches_oracle_composition():
    ck, net = _teacher()
    opt = scene.default_opt(W=40, H=40)
    o, d = oracle.get_rays(scene.orbit_pose(4.0, 40.0, -20.0), scene.orbit_intrinsics(40, 40, 50.0), 40, 40)
    nears, fars = oracle.near_far_from_aabb(o, d, np.array([-1, -1, -1, 1, 1, 1], np.float32), 0.2)
    xyzs, dirs, deltas, rays = otr.march_rays_train(o, d, 1.0, ck["density_bitfield"], 1, 128, nears, fars, None, -1, None, 128, False, 0.0, 1024)
    sig, rgb = oracle.nerf_forward(xyzs, dirs, ck, 1.0)
    ws, depth, image = otr.composite_rays_train_forward(sig, rgb, deltas, rays, 1e-2)
    image = image + (1 - ws)[:, None]
    net.train()
    out = net.run_cuda(T(o)[None], T(d)[None], dt_gamma=0, perturb=False, max_steps=1024, T_thresh=1e-2)
    assert out["image"].requires_grad and net.local_step == 1
    assert int(net.step_counter[0, 0]) == int(rays[:, 2].sum()) and int(net.step_counter[0, 1]) == len(o)
    assert np.abs(out["image"][0].detach().cpu().numpy() - image).max() < 1e-4
    assert np.abs(out["weights_sum"].detach().cpu().numpy() - ws).max() < 1e-4
    # directional derivative of a scalar loss along a random direction in weight space vs autograd
    w = net.color_net[2].weight
    target = torch.rand_like(out["image"])
    loss = ((out["image"] - target) ** 2).mean()
    loss.backward()
    assert net.encoder.embeddings.grad is not None and net.encoder.embeddings.grad.abs().max() > 0
    direction = torch.randn_like(w)
    analytic = float((w.grad * direction).sum())
    eps = 1e-2
    vals = []
    for sgn in (1, -1):
        with torch.no_grad():
            w.add_(sgn * eps * direction)
        vals.append(float(((net.run_cuda(T(o)[None], T(d)[None], perturb=False, T_thresh=1e-2)["image"].detach() - target) ** 2).mean()))
        with torch.no_grad():
            w.sub_(sgn * eps * direction)
    numeric = (vals[0] - vals[1]) / (2 * eps)
    assert abs(numeric - analytic) < 0.05 * abs(analytic) + 1e-6, (numeric, analytic)


def test_update_extra_state_and_mark_untrained_grid():
    ck, net = _teacher()
    net.reset_extra_state()
    assert not net.density_grid.any() and net.mean_count == 0
    poses = np.stack([scene.orbit_pose(2.5, a, -20.0) for a in (0.0, 40.0)])    # two nearby narrow cameras: part of the volume is never seen
    n_unseen = net.mark_untrained_grid(poses, scene.orbit_intrinsics(64, 64, 25.0))
    assert 0 < n_unseen < 128 ** 3 and int((net.density_grid == -1).sum()) == n_unseen
    torch.manual_seed(0)
    net.local_step = 3
    net.step_counter[:3, 0] = torch.tensor([100, 200, 600], dtype=torch.int32)
    net.update_extra_state()
    grid = net.density_grid.cpu().numpy()
    assert (grid[0] == -1).sum() == n_unseen                       # unseen cells stay unseen
    assert net.iter_density == 1 and net.mean_count == 300 and net.local_step == 0
    seen = grid[0] >= 0
    # the grid now holds density_scale * sigma at jittered cell centres: the oracle's sigma at the exact centres agrees in the bulk
    idx = np.flatnonzero(seen)[::997]
    coords = oracle.morton3D_invert(idx.astype(np.int32))
    centres = (2 * coords.astype(np.float32) / 127 - 1) * np.float32(1 - 1 / 128)
    sig, _ = oracle.nerf_forward(centres, np.tile(np.float32([0, 0, 1]), (len(idx), 1)), ck, 1.0)
    inside = sig > 30
    assert inside.sum() > 20 and (grid[0][idx][inside] > 5).mean() > 0.9
    # bitfield == packbits(grid > min(mean, density_thresh)), bit for bit
    thresh = min(net.mean_density, net.density_thresh)
    assert np.array_equal(net.density_bitfield.cpu().numpy(), oracle.packbits(grid, thresh))
    # the EMA-max rule: a second update never lowers a seen cell below decay * old
    old = net.density_grid.clone()
    net.update_extra_state(decay=0.95)
    new = net.density_grid
    ok = old >= 0
    assert bool((new[ok] >= 0.95 * old[ok] - 1e-6).all()) and bool((new[~ok] == -1).all())
    # partial-update branch (iter_density >= 16)
    net.iter_density = 16
    net.update_extra_state()
    assert net.iter_density == 17 and bool((net.density_grid[~ok] == -1).all())


@pytest.mark.parametrize("fp16", [False, True])
def test_training_fits_the_teacher_images(fp16):
    """fp16 (trainer.py:20,84,629-642: ``--fp16``): the same run under autocast with a GradScaler — half tables and features, the half scatter-add of the
    grid's backward (gridencoder.cu:324-331), half nn.Linear."""
    ck, teacher = _teacher()
    Wd = 64
    intr = scene.orbit_intrinsics(Wd, Wd, 50.0)
    poses = np.stack([scene.orbit_pose(4.0, a, e) for a in (0.0, 60.0, 120.0, 180.0, 240.0, 300.0) for e in (-20.0, -50.0)]).astype(np.float32)
    from pienerf_amd.nerf.utils import get_rays
    images = []
    with torch.no_grad():
        for p in poses:
            r = get_rays(T(p[None]), intr, Wd, Wd)
            out = teacher.run_cuda(r["rays_o"], r["rays_d"], bg_color=1, max_steps=1024, T_thresh=1e-2)
            images.append(out["image"].view(Wd, Wd, 3))
    images = torch.stack(images)
    assert float(images.std()) > 0.05
    torch.manual_seed(1)
    student = NeRFNetwork(encoding="hashgrid", bound=1.0, cuda_ray=True, density_thresh=10).to(DEV)
    data = RayImageSet(T(poses), intr, images, generator=torch.Generator().manual_seed(2))
    tr = Trainer(student, dict(dt_gamma=0, max_steps=512, T_thresh=1e-2), lr=1e-2, iters=400, num_rays=2048, fp16=fp16)
    psnr0, _ = tr.evaluate(data, 0)
    losses = tr.train(data, 400)
    psnr1, out = tr.evaluate(data, 0)
    assert np.isfinite(losses).all()
    assert np.mean(losses[-20:]) < 0.25 * np.mean(losses[:5]), (losses[:5], losses[-20:])
    assert psnr1 > psnr0 + 5 and psnr1 > 20, (psnr0, psnr1)
    assert student.mean_count > 0 and student.iter_density == 25
    # the trained bitfield keeps the object and drops most of the empty space
    occ = np.unpackbits(student.density_bitfield.cpu().numpy()).mean()
    assert 0.001 < occ < 0.6


def test_training_state_with_two_cascades():
    """bound = 2 (the trex option set): two density cascades, 4096-resolution hash grid — density-grid sweep, partial update, one training step
    with gradients through march_rays_train / composite_rays_train at dt_gamma = 1/128, and the eval render afterwards."""
    ck = scene.make_checkpoint(bound=2.0, seed=3, shaped=True)
    net = NeRFNetwork(encoding="hashgrid", bound=2.0, cuda_ray=True, density_thresh=10).to(DEV).load_checkpoint_dict(ck)
    assert net.cascade == 2 and net.density_grid.shape == (2, 128 ** 3) and net.density_bitfield.shape[0] == 2 * 128 ** 3 // 8
    net.reset_extra_state()
    torch.manual_seed(0)
    net.update_extra_state()
    grid = net.density_grid.cpu().numpy()
    assert (grid >= 0).all() and (grid[0] > 5).sum() > 1000 and (grid[1] > 5).sum() > 100      # the object shows in both cascades
    assert np.array_equal(net.density_bitfield.cpu().numpy(), oracle.packbits(grid, min(net.mean_density, net.density_thresh)))
    net.iter_density = 16
    net.update_extra_state()
    W = 48
    o, d = oracle.get_rays(scene.orbit_pose(4.5, 25.0, -10.0), scene.orbit_intrinsics(W, W, 50.0), W, W)
    net.train()
    out = net.run_cuda(T(o)[None], T(d)[None], dt_gamma=1.0 / 128, perturb=True, max_steps=300, T_thresh=5e-2)
    out["image"].sum().backward()
    g = net.encoder.embeddings.grad
    assert g is not None and bool(torch.isfinite(g).all()) and float(g.abs().max()) > 0
    assert int(net.step_counter[0, 0]) > 500 and int(net.step_counter[0, 1]) == W * W
    net.eval()
    img = net.run_cuda(T(o)[None], T(d)[None], dt_gamma=1.0 / 128, max_steps=300, T_thresh=5e-2)["image"]
    assert bool(torch.isfinite(img).all()) and float(img.min()) < 0.9


def test_grid_state_kernels_match_oracle_bit_for_bit():
    """pn_grid_state.hip against the numpy restatement (oracle/training.py): the set of unseen cells of mark_untrained_grid, the jittered
    cell samples of the full sweep, and the EMA-max / mean / bitfield update — integer and index work bit-exact, the mean to 1e-6."""
    import ctypes as C
    from pienerf_amd._lib import check, lib, ptr, stream_ptr
    H, bound, cascade = 128, 2.0, 2
    poses = np.stack([scene.orbit_pose(3.2, a, e) for a, e in ((0.0, -20.0), (35.0, -10.0), (200.0, -50.0))]).astype(np.float32)
    intr = scene.orbit_intrinsics(64, 48, 30.0)
    net = NeRFNetwork(encoding="hashgrid", bound=bound, cuda_ray=True).to(DEV)
    net.reset_extra_state()
    n = net.mark_untrained_grid(poses, intr)
    want = otr.mark_untrained_grid(poses, intr, cascade, H, bound)
    got = (net.density_grid == -1).cpu().numpy()
    assert n == int(want.sum()) and 0 < n < cascade * H ** 3 and np.array_equal(got, want)
    # full-sweep samples
    noise = torch.rand(cascade * H ** 3, 3, device=DEV)
    pts = torch.empty(cascade * H ** 3, 3, device=DEV)
    check(lib().pn_density_cells_full(cascade, H, bound, ptr(noise), ptr(pts), stream_ptr()), "cells")
    ref = otr.density_cells_full(cascade, H, bound, noise.cpu().numpy())
    assert np.array_equal(pts.cpu().numpy().view(np.uint32), ref.view(np.uint32))
    assert np.abs(ref[:H ** 3]).max() <= 1.0 and 1.5 < np.abs(ref[H ** 3:]).max() <= 2.0       # cascade 0 spans +-1, cascade 1 +-2
    # update: EMA-max, mean, bitfield
    rng = np.random.default_rng(0)
    g0 = rng.uniform(-0.5, 30, (cascade, H ** 3)).astype(np.float32)
    g0[g0 < 0] = -1
    t0 = rng.uniform(-0.5, 30, (cascade, H ** 3)).astype(np.float32)
    t0[t0 < 0] = -1
    grid, tmp = T(g0), T(t0)
    bits = torch.zeros(cascade * H ** 3 // 8, dtype=torch.uint8, device=DEV)
    partial = torch.empty((cascade * H ** 3 + 255) // 256, dtype=torch.float64, device=DEV)
    mt = torch.empty(2, device=DEV)
    check(lib().pn_density_grid_update(cascade * H ** 3, ptr(grid), ptr(tmp), 0.95, 10.0, ptr(bits), ptr(partial), ptr(mt), stream_ptr()), "update")
    g_ref, mean_ref, bits_ref = otr.density_grid_update(g0, t0, 0.95, 10.0)
    assert np.array_equal(grid.cpu().numpy(), g_ref) and abs(float(mt[0]) - mean_ref) < 1e-6 * mean_ref
    assert float(mt[1]) == min(float(mt[0]), 10.0) and np.array_equal(bits.cpu().numpy(), bits_ref)
    # partial sweep: indices are morton codes of the drawn cells / of occupied cells, samples lie inside their cells
    N = H ** 3 // 4
    draws = torch.randint(0, H, (N, 3), device=DEV, dtype=torch.int32)
    scratch = torch.empty(int(lib().pn_density_partial_scratch_ints(H)), dtype=torch.int32, device=DEV)
    idx = torch.empty(2 * N, dtype=torch.int32, device=DEV)
    pts = torch.empty(2 * N, 3, device=DEV)
    tmpc = torch.zeros(H ** 3, device=DEV)
    check(lib().pn_density_cells_partial(1, H, bound, N, ptr(draws), ptr(torch.rand(N, device=DEV)), ptr(torch.rand(2 * N, 3, device=DEV)), ptr(grid[1]),
                                         ptr(tmpc), ptr(scratch), ptr(idx), ptr(pts), stream_ptr()), "partial")
    idx_h, pts_h = idx.cpu().numpy(), pts.cpu().numpy()
    assert bool((tmpc == -1).all())
    assert np.array_equal(idx_h[:N], oracle.morton3D(draws.cpu().numpy()))
    occ = g_ref[1] > 0
    assert occ[idx_h[N:]].all() and len(np.unique(idx_h[N:])) > N // 4                 # draws from the occupied cells, well spread
    centres, half = otr._cell_centres(1, H, bound)
    assert np.abs(pts_h - centres[idx_h]).max() <= half * (1 + 1e-6)


def test_run_cuda_eval_fused_equals_op_loop_and_oracle():
    """NeRFRenderer.run_cuda in eval(): the device-driven frame (pn_render_static) reproduces the op-by-op loop bit for bit (same kernels on the
    same samples) and the CPU oracle's static render within 1e-4; it also runs captured in a HIP graph (fixed trip count, no host sync)."""
    ck, net = _teacher()
    W = 72
    o, d = oracle.get_rays(scene.orbit_pose(4.0, 30.0, -25.0), scene.orbit_intrinsics(W, W, 45.0), W, W)
    opt = dict(dt_gamma=0.0, max_steps=1024, T_thresh=1e-2)
    with torch.no_grad():
        a = net.run_cuda(T(o)[None], T(d)[None], **opt)
        sa = dict(net.last_stats)
        b = net.run_cuda_ops(T(o)[None], T(d)[None], **opt)
        sb = dict(net.last_stats)
    assert sa["trips"] == sb["trips"] and sa["samples"] == sb["samples"] and sa["alive_at_exit"] == 0
    assert torch.equal(a["image"], b["image"]) and torch.equal(a["weights_sum"], b["weights_sum"])
    da, db = a["depth"], b["depth"]
    assert torch.equal(torch.isnan(da), torch.isnan(db)) and torch.equal(da[~torch.isnan(da)], db[~torch.isnan(db)])
    ref = oracle.render_static(o, d, ck, dict(opt))
    assert ref["samples"] == sa["samples"] and np.abs(a["image"][0].cpu().numpy() - ref["image"]).max() < 1e-4
    # captured: fixed number of trips, completion checked afterwards
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    ro, rd = T(o)[None], T(d)[None]
    with torch.no_grad(), torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
        c = net.run_cuda(ro, rd, async_trips=sa["trips"] + 2, **opt)
    g.replay()
    torch.cuda.synchronize()
    st = net.render_status()
    assert st["alive_at_exit"] == 0 and st["samples"] == sa["samples"] and torch.equal(c["image"], a["image"])
