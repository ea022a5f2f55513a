"""SURVEY §8(f) rank 3, training side: march_rays_train, composite_rays_train fwd/bwd, grid_encode dy_dx / backward /
grad_total_variation, sh_encode dy_dx / backward — the HIP path (through the C ABI and the autograd Functions that mirror the
reference's) against the oracle's restatements.  Bars: ray rows, point ranges and sample coordinates bit-exact; composited values and
gradients within 1e-4 relative (atomic summation order and __expf differ from the CPU)."""
import numpy as np
import pytest
import torch

import oracle
from oracle import training as otr
from pienerf_amd import raymarching, scene
from pienerf_amd.gridencoder import GridEncoder
from pienerf_amd.gridencoder.grid import grid_encode, level_table_offsets
from pienerf_amd.shencoder.sphere_harmonics import sh_encode
from test_gpu_parity import DEV, T

pytestmark = pytest.mark.gpu


def _rays(bound, W):
    o, d = oracle.get_rays(scene.orbit_pose(3.4 * bound, 25.0, -20.0), scene.orbit_intrinsics(W, W, 50.0), W, W)
    nears, fars = oracle.near_far_from_aabb(o, d, np.array([-bound] * 3 + [bound] * 3, np.float32), 0.2)
    return o, d, nears, fars


def rel(a, b):
    return float(np.abs(np.asarray(a, np.float64) - b).max() / max(1e-30, np.abs(b).max()))


@pytest.mark.parametrize("bound,dt_gamma,max_steps,W", [(1.0, 0.0, 1024, 64), (2.0, 1.0 / 128, 300, 48), (1.0, 0.0, 128, 1)])
def test_march_rays_train_bit_exact(bound, dt_gamma, max_steps, W):
    ck = scene.make_checkpoint(bound=bound, seed=1)
    o, d, nears, fars = _rays(bound, W)
    N = len(o)
    cref = np.zeros(2, np.int32)
    ref = otr.march_rays_train(o, d, bound, ck["density_bitfield"], ck["cascade"], ck["grid_size"], nears, fars, cref, -1, None, 128, False, dt_gamma,
                               max_steps)
    counter = torch.zeros(2, dtype=torch.int32, device=DEV)
    got = raymarching.march_rays_train(T(o), T(d), bound, T(ck["density_bitfield"]), ck["cascade"], ck["grid_size"], T(nears), T(fars), counter, -1, False,
                                       128, False, dt_gamma, max_steps)
    assert np.array_equal(counter.cpu().numpy(), cref) and cref[1] == N and (W == 1 or cref[0] > 500)
    for a, b in zip(got, ref):
        assert a.shape == b.shape and np.array_equal(a.cpu().numpy(), b)
    # the counter accumulates across calls exactly like the reference's atomicAdd target (raymarching.cu:404-405)
    raymarching.march_rays_train(T(o), T(d), bound, T(ck["density_bitfield"]), ck["cascade"], ck["grid_size"], T(nears), T(fars), counter, -1, False, 128,
                                 True, dt_gamma, max_steps)
    assert np.array_equal(counter.cpu().numpy(), 2 * cref)


def test_march_rays_train_point_budget_and_noise():
    ck = scene.make_checkpoint(bound=1.0, seed=1)
    o, d, nears, fars = _rays(1.0, 56)
    noise = np.random.default_rng(3).random(len(o)).astype(np.float32)
    full = otr.march_rays_train(o, d, 1.0, ck["density_bitfield"], 1, 128, nears, fars, None, -1, noise, -1, False, 0.0, 512)
    budget = int(full[3][:, 2].sum()) // 3
    cref = np.zeros(2, np.int32)
    ref = otr.march_rays_train(o, d, 1.0, ck["density_bitfield"], 1, 128, nears, fars, cref, budget, noise, 128, False, 0.0, 512)
    # the C ABI takes the noise vector; the Python wrapper draws it with torch.rand, so call the library directly here
    from pienerf_amd._lib import check, lib, ptr, stream_ptr
    M = ref[0].shape[0]
    xyzs, dirs, deltas = (torch.zeros(M, k, device=DEV) for k in (3, 3, 2))
    rays = torch.empty(len(o), 3, dtype=torch.int32, device=DEV)
    counter = torch.zeros(2, dtype=torch.int32, device=DEV)
    to, td, tg, tn, tf, tz = T(o), T(d), T(ck["density_bitfield"]), T(nears), T(fars), T(noise)  # keep the device copies alive across the launch
    check(lib().pn_march_rays_train(ptr(to), ptr(td), ptr(tg), 1.0, 0.0, 512, len(o), 1, 128, M, ptr(tn), ptr(tf), ptr(xyzs), ptr(dirs), ptr(deltas),
                                    ptr(rays), ptr(counter), ptr(tz), stream_ptr()))
    assert counter[0].item() == cref[0] > M                     # demand exceeds the budget: later rays are dropped, not truncated
    for a, b in zip((xyzs, dirs, deltas, rays), ref):
        assert np.array_equal(a.cpu().numpy(), b)
    # perturb=True through the wrapper: same ray rows up to the first sample shift, all starts inside one dt of the unperturbed ones
    got = raymarching.march_rays_train(T(o), T(d), 1.0, T(ck["density_bitfield"]), 1, 128, T(nears), T(fars), None, -1, True, -1, False, 0.0, 512)
    assert got[3].shape == (len(o), 3) and abs(int(got[3][:, 2].sum()) - int(full[3][:, 2].sum())) < 0.02 * full[3][:, 2].sum()


def _ray_batch(rng, N, max_len):
    lens = rng.integers(1, max_len, N)
    lens[::7] = 0
    offs = np.cumsum(lens) - lens
    M = int(lens.sum())
    rays = np.stack([rng.permutation(N), offs, lens], 1).astype(np.int32)
    sig = rng.uniform(0, 40, M).astype(np.float32)
    rgb = rng.uniform(0, 1, (M, 3)).astype(np.float32)
    deltas = np.stack([rng.uniform(0.002, 0.02, M), rng.uniform(0.002, 0.05, M)], 1).astype(np.float32)
    return rays, sig, rgb, deltas


@pytest.mark.parametrize("T_thresh", [1e-4, 5e-2])
def test_composite_rays_train_forward_backward(T_thresh):
    rng = np.random.default_rng(4)
    N = 5000
    rays, sig, rgb, deltas = _ray_batch(rng, N, 120)
    ws_r, depth_r, image_r = otr.composite_rays_train_forward(sig, rgb, deltas, rays, T_thresh)
    ts, tc = T(sig).requires_grad_(True), T(rgb).requires_grad_(True)
    ws, depth, image = raymarching.composite_rays_train(ts, tc, T(deltas), T(rays), T_thresh)
    assert rel(ws.detach().cpu().numpy(), ws_r) < 1e-5 and rel(depth.detach().cpu().numpy(), depth_r) < 1e-5
    assert rel(image.detach().cpu().numpy(), image_r) < 1e-5
    gws, gim = rng.standard_normal(N).astype(np.float32), rng.standard_normal((N, 3)).astype(np.float32)
    gs_r, gc_r = otr.composite_rays_train_backward(gws, gim, sig, rgb, deltas, rays, ws_r, image_r, T_thresh)
    # depth gets a gradient too; the reference (and this build) does not propagate it (raymarching.py:271)
    ((ws * T(gws)).sum() + (image * T(gim)).sum() + depth.sum()).backward()
    assert rel(tc.grad.cpu().numpy(), gc_r) < 1e-5
    assert rel(ts.grad.cpu().numpy(), gs_r) < 1e-4
    # rays beyond the point budget: zero outputs, untouched gradients
    M2 = int(rays[N // 2, 1])
    ws2, _, im2 = raymarching.composite_rays_train(T(sig[:M2]), T(rgb[:M2]), T(deltas[:M2]), T(rays), T_thresh)
    ref2 = otr.composite_rays_train_forward(sig[:M2], rgb[:M2], deltas[:M2], rays, T_thresh)
    assert rel(ws2.cpu().numpy(), ref2[0]) < 1e-5 and not ws2.cpu().numpy()[rays[N // 2:, 0]].any()


@pytest.mark.parametrize("interp,log2_T", [(0, 12), (1, 12), (0, 19)])
def test_grid_encode_backward_and_dy_dx(interp, log2_T):
    pls, base, L = 1.6, 8, 6
    offsets = level_table_offsets(3, L, pls, base, log2_T, False)
    rng = np.random.default_rng(6)
    emb = rng.uniform(-1, 1, (int(offsets[-1]), 2)).astype(np.float32)
    B = 20000
    x = rng.uniform(0, 1, (B, 3)).astype(np.float32)
    x[:9] = [1.2, 0.5, 0.5]
    x[9] = [0.0, 0.0, 1.0]
    grad = rng.standard_normal((B, L * 2)).astype(np.float32)
    dy_dx_r = otr.grid_encode_dy_dx(x, emb, offsets, pls, base, 0, False, interp)
    gi_r, ge_r = otr.grid_encode_backward(grad, x, emb.shape, offsets, pls, base, dy_dx_r, 0, False, interp)
    tx, te = T(x).requires_grad_(True), T(emb).requires_grad_(True)
    y = grid_encode(tx, te, T(offsets), pls, base, True, 0, False, interp)
    assert rel(y.detach().cpu().numpy(), oracle.grid_encode_forward(x, emb, offsets, pls, base, 0, False, interp)) < 1e-5
    y.backward(T(grad))
    assert rel(te.grad.cpu().numpy(), ge_r) < 1e-4            # atomic summation order differs from the oracle's sample order
    assert rel(tx.grad.cpu().numpy(), gi_r) < 1e-4
    assert not tx.grad[:9].any()
    # without calc_grad_inputs: same table gradient, no input gradient
    te2 = T(emb).requires_grad_(True)
    grid_encode(T(x), te2, T(offsets), pls, base, False, 0, False, interp).backward(T(grad))
    assert rel(te2.grad.cpu().numpy(), ge_r) < 1e-4


@pytest.mark.parametrize("interp,log2_T,C", [(0, 12, 2), (1, 12, 4), (0, 19, 2)])
def test_grid_encode_backward_under_autocast(interp, log2_T, C):
    """kernel_grid_backward<at::Half> (gridencoder.cu:248-341; the __half2 atomicAdd of :324-331): under autocast the table is half (grid.py:43-44),
    every contribution (__half)(w * grad) is accumulated in half, in the order the atomics meet.  Against the oracle's backward on the same
    half-rounded gradients: each table entry within the rounding a sum of its contributions' halves can have (contributions x half epsilon x the
    largest partial sum), and nothing lost (the fp32 sums agree to 1e-2 of the largest entry); rows no sample touches stay exactly zero."""
    pls, base, L = 1.6, 8, 6
    offsets = level_table_offsets(3, L, pls, base, log2_T, False)
    rng = np.random.default_rng(16)
    emb = rng.uniform(-1, 1, (int(offsets[-1]), C)).astype(np.float32)
    B = 20000
    x = rng.uniform(0, 1, (B, 3)).astype(np.float32)
    x[:9] = [1.2, 0.5, 0.5]
    grad = (rng.standard_normal((B, L * C)) * 1e-2).astype(np.float16)
    _, ge_r = otr.grid_encode_backward(grad.astype(np.float32), x, emb.shape, offsets, pls, base, None, 0, False, interp)
    _, ge_abs = otr.grid_encode_backward(np.abs(grad.astype(np.float32)), x, emb.shape, offsets, pls, base, None, 0, False, interp)
    te = T(emb).requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.float16):
        y = grid_encode(T(x), te, T(offsets), pls, base, False, 0, False, interp)
    assert y.dtype == torch.float16 and y.requires_grad
    y.backward(torch.from_numpy(grad).to(DEV))
    got = te.grad.cpu().numpy()
    assert te.grad.dtype == torch.float32 and np.isfinite(got).all()
    # a sum of n halves, every term and every partial sum rounded to half (2^-11 relative), partial sums bounded by the sum of magnitudes; the order is the
    # atomics' (it changes from run to run): |error| <= 2^-6 x sum |terms| covers the coarse levels' hundreds of contributions per entry with room
    assert (np.abs(got - ge_r) <= 2.0 ** -6 * ge_abs + 1e-6).all(), float((np.abs(got - ge_r) - 2.0 ** -6 * ge_abs).max())
    assert rel(got, ge_r) < 1e-2
    assert not got[ge_abs == 0].any()
    with pytest.raises(RuntimeError), torch.autocast("cuda", dtype=torch.float16):
        grid_encode(T(x).requires_grad_(True), te, T(offsets), pls, base, True, 0, False, interp)


def test_grid_encoder_module_gradients_on_the_chair_tables(ckpt):
    """The real table geometry (16 levels, 2^19 entries, 6.1 M rows): autograd through GridEncoder against the oracle."""
    enc = GridEncoder(input_dim=3, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=19, desired_resolution=2048).to(DEV)
    with torch.no_grad():
        enc.embeddings.copy_(T(ckpt["embeddings"]))
    rng = np.random.default_rng(9)
    B = 4096
    p = rng.uniform(-1, 1, (B, 3)).astype(np.float32)
    grad = rng.standard_normal((B, 32)).astype(np.float32)
    tp = T(p).requires_grad_(True)
    enc(tp, bound=1).backward(T(grad))
    u = (p + np.float32(1)) / np.float32(2)
    off, pls = ckpt["offsets"], ckpt["per_level_scale"]
    dy_dx = otr.grid_encode_dy_dx(u, ckpt["embeddings"], off, pls, 16)
    gi, ge = otr.grid_encode_backward(grad, u, ckpt["embeddings"].shape, off, pls, 16, dy_dx)
    assert rel(enc.embeddings.grad.cpu().numpy(), ge) < 1e-4
    assert rel(tp.grad.cpu().numpy(), gi / 2) < 1e-4           # d u / d p = 1 / (2 bound)
    # total variation on top of the accumulated gradient
    pts = rng.uniform(-1, 1, (30000, 3)).astype(np.float32)
    want = otr.grad_total_variation((pts + np.float32(1)) / np.float32(2), ckpt["embeddings"], ge.copy(), off, pls, 16, weight=1e-3)
    enc.grad_total_variation(1e-3, T(pts), bound=1)
    tv = want - ge
    assert np.abs(tv).max() > 1e-6 and np.abs(enc.embeddings.grad.cpu().numpy() - want).max() < 1e-4 * np.abs(want).max()
    enc.embeddings.grad = None
    with pytest.raises(ValueError):
        enc.grad_total_variation(1e-3, T(pts))


def test_grad_total_variation_small_tables():
    pls, base, L = 2.0, 4, 4
    offsets = level_table_offsets(3, L, pls, base, 10, False)       # levels 2,3 hashed with heavy collisions
    rng = np.random.default_rng(7)
    emb = rng.uniform(-1, 1, (int(offsets[-1]), 2)).astype(np.float32)
    x = rng.uniform(0, 1, (5000, 3)).astype(np.float32)
    x[0], x[1], x[2] = [0, 0, 0], [1, 1, 1], [-0.1, 0.5, 0.5]
    want = otr.grad_total_variation(x, emb, np.zeros_like(emb), offsets, pls, base, weight=0.3)
    from pienerf_amd._lib import check, lib, ptr, stream_ptr
    g = torch.zeros(emb.shape, device=DEV)
    off_host = torch.from_numpy(offsets)
    tx, te = T(x), T(emb)
    check(lib().pn_grad_total_variation(ptr(tx), ptr(te), ptr(g), off_host.data_ptr(), 0.3, len(x), 3, 2, L, float(np.float32(np.log2(pls))), base, 0, 0,
                                        stream_ptr()))
    assert rel(g.cpu().numpy(), want) < 1e-4


@pytest.mark.parametrize("degree", [1, 2, 3, 4])
def test_sh_encode_dy_dx_and_backward(degree):
    rng = np.random.default_rng(8)
    d = rng.standard_normal((10000, 3))
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    grad = rng.standard_normal((len(d), degree * degree)).astype(np.float32)
    td = T(d).requires_grad_(True)
    y = sh_encode(td, degree, True)
    assert rel(y.detach().cpu().numpy(), oracle.sh_encode_forward(d, degree)) < 1e-6
    y.backward(T(grad))
    dy_dx = otr.sh_encode_dy_dx(d, degree)
    assert rel(td.grad.cpu().numpy(), otr.sh_encode_backward(grad, dy_dx, degree)) < 1e-5
    # no input gradient requested: backward returns None for the directions
    td2 = T(d).requires_grad_(True)
    assert sh_encode(td2.detach(), degree, False).requires_grad is False


def test_training_ops_reject_cpu_tensors():
    with pytest.raises(RuntimeError):
        raymarching.composite_rays_train(torch.zeros(4), torch.zeros(4, 3), torch.zeros(4, 2), torch.zeros(1, 3, dtype=torch.int32))
    with pytest.raises(RuntimeError):
        grid_encode(torch.zeros(4, 3), torch.zeros(64, 2), torch.tensor([0, 64], dtype=torch.int32), 2.0, 4)
