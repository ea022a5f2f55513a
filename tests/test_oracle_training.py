"""Pins the training-side restatements of the CPU oracle by independent routes: torch autograd in float64 on
closed-form expressions, finite differences, and the inference march."""
import numpy as np
import pytest
import torch

import oracle
from oracle import training as otr
from pienerf_amd import scene


def _rays(bound, W, seed_pose=(3.4, 25.0, -20.0)):
    o, d = oracle.get_rays(scene.orbit_pose(seed_pose[0] * bound, seed_pose[1], seed_pose[2]), scene.orbit_intrinsics(W, W, 50.0), W, W)
    aabb = np.array([-bound] * 3 + [bound] * 3, np.float32)
    nears, fars = oracle.near_far_from_aabb(o, d, aabb, 0.2)
    return o, d, nears, fars


@pytest.mark.parametrize("bound,dt_gamma,max_steps", [(1.0, 0.0, 256), (2.0, 1.0 / 128, 128)])
def test_march_rays_train_equals_inference_march(bound, dt_gamma, max_steps):
    """The training march of one ray is the inference march of that ray run to the end: same points, same deltas; `rays` rows are
    (n, exclusive prefix, count) and the counters hold (total points, N)."""
    ck = scene.make_checkpoint(bound=bound, seed=2)
    o, d, nears, fars = _rays(bound, 28)
    N = len(o)
    counter = np.zeros(2, np.int32)
    xyzs, dirs, deltas, rays = otr.march_rays_train(o, d, bound, ck["density_bitfield"], ck["cascade"], ck["grid_size"], nears, fars, counter, -1, None, 128,
                                                    False, dt_gamma, max_steps)
    cnt = rays[:, 2]
    assert np.array_equal(rays[:, 0], np.arange(N)) and np.array_equal(rays[:, 1], np.cumsum(cnt) - cnt)
    total = int(cnt.sum())
    assert total > 200 and counter[0] == total and counter[1] == N
    assert xyzs.shape[0] == total + 128 - total % 128 and not deltas[total:].any()
    alive = np.arange(N, dtype=np.int32)
    ix, idr, idl = oracle.march_rays(N, max_steps, alive, nears, o, d, bound, ck["density_bitfield"], ck["cascade"], ck["grid_size"], nears, fars, -1, None,
                                     dt_gamma, max_steps)
    ix, idl = ix.reshape(N, max_steps, 3), idl.reshape(N, max_steps, 2)
    assert np.array_equal((idl[..., 0] != 0).sum(1), cnt)
    for n in np.flatnonzero(cnt)[::7]:
        s = slice(rays[n, 1], rays[n, 1] + cnt[n])
        assert np.array_equal(xyzs[s], ix[n, :cnt[n]]) and np.array_equal(deltas[s], idl[n, :cnt[n]])
        assert np.array_equal(dirs[s], np.broadcast_to(d[n], (cnt[n], 3)))


def test_march_rays_train_budget_drops_late_rays_and_noise_shifts_start():
    ck = scene.make_checkpoint(bound=1.0, seed=2)
    o, d, nears, fars = _rays(1.0, 24)
    full = otr.march_rays_train(o, d, 1.0, ck["density_bitfield"], 1, 128, nears, fars, None, -1, None, -1, False, 0.0, 256)
    total = int(full[3][:, 2].sum())
    budget = total // 2
    counter = np.zeros(2, np.int32)
    xyzs, dirs, deltas, rays = otr.march_rays_train(o, d, 1.0, ck["density_bitfield"], 1, 128, nears, fars, counter, budget, None, 128, False, 0.0, 256)
    M = budget + 128 - budget % 128
    assert xyzs.shape[0] == M and counter[0] == total          # the counter still reports the demand (raymarching.py:225, renderer.py:326-331)
    assert np.array_equal(rays, full[3])
    fits = rays[:, 1] + rays[:, 2] <= M
    for n in np.flatnonzero(rays[:, 2] > 0):
        if fits[n]:
            s = slice(rays[n, 1], rays[n, 1] + rays[n, 2])
            assert np.array_equal(xyzs[s], full[0][s])
    last = np.flatnonzero(fits & (rays[:, 2] > 0))[-1]
    assert not deltas[rays[last, 1] + rays[last, 2]:].any()     # nothing written past the last ray that fits
    # noise: t0 = near + dt * noise (raymarching.cu:351)
    noise = np.full(len(o), 0.5, np.float32)
    shifted = otr.march_rays_train(o, d, 1.0, ck["density_bitfield"], 1, 128, nears, fars, None, -1, noise, -1, True, 0.0, 256)
    n = int(np.argmax(full[3][:, 2]))
    a, b = full[0][full[3][n, 1]], shifted[0][shifted[3][n, 1]]
    assert 0 < np.linalg.norm(a - b) < 2 * np.sqrt(3) / 256


def _random_ray_batch(rng, N, max_len, with_empty=True):
    lens = rng.integers(1, max_len, N)
    if with_empty:
        lens[::5] = 0
    offs = np.cumsum(lens) - lens
    M = int(lens.sum())
    rays = np.stack([rng.permutation(N), offs, lens], 1).astype(np.int32)  # ray ids are a permutation: row n != pixel index
    sig = rng.uniform(0, 40, M).astype(np.float32)
    rgb = rng.uniform(0, 1, (M, 3)).astype(np.float32)
    deltas = np.stack([rng.uniform(0.002, 0.02, M), rng.uniform(0.002, 0.05, M)], 1).astype(np.float32)
    return rays, sig, rgb, deltas


def _composite_torch(sig, rgb, deltas, rays, T_thresh, N):
    """Closed form in float64 with torch autograd; the early break is applied as a mask computed without gradient."""
    ws, depth, image = [torch.zeros((), dtype=torch.float64)] * N, [torch.zeros((), dtype=torch.float64)] * N, [torch.zeros(3, dtype=torch.float64)] * N
    for idx, off, n in rays.tolist():
        if n == 0:
            continue
        s, c, dl = sig[off:off + n], rgb[off:off + n], deltas[off:off + n]
        alpha = 1 - torch.exp(-s * dl[:, 0])
        T_after = torch.cumprod(1 - alpha, 0)
        T_before = torch.cat([torch.ones(1, dtype=torch.float64), T_after[:-1]])
        stop = (T_after.detach() < T_thresh).nonzero()
        last = int(stop[0]) if len(stop) else n - 1     # the sample where T drops below the threshold is still accumulated
        w = (alpha * T_before)[:last + 1]
        t = torch.cumsum(dl[:, 1], 0)[:last + 1]
        ws[idx], depth[idx], image[idx] = w.sum(), (w * t).sum(), (w[:, None] * c[:last + 1]).sum(0)
    return torch.stack(ws), torch.stack(depth), torch.stack(image)


@pytest.mark.parametrize("T_thresh", [1e-4, 5e-2])
def test_composite_train_forward_backward_vs_autograd(T_thresh):
    rng = np.random.default_rng(4)
    N = 60
    rays, sig, rgb, deltas = _random_ray_batch(rng, N, 50)
    ws, depth, image = otr.composite_rays_train_forward(sig, rgb, deltas, rays, T_thresh)
    ts, tc = torch.tensor(sig, dtype=torch.float64, requires_grad=True), torch.tensor(rgb, dtype=torch.float64, requires_grad=True)
    tws, tdepth, timage = _composite_torch(ts, tc, torch.tensor(deltas, dtype=torch.float64), rays, T_thresh, N)
    assert np.allclose(ws, tws.detach().numpy(), atol=2e-6) and np.allclose(depth, tdepth.detach().numpy(), atol=2e-6)
    assert np.allclose(image, timage.detach().numpy(), atol=2e-6)
    empty = rays[rays[:, 2] == 0, 0]
    assert not ws[empty].any() and not image[empty].any()
    # backward: the reference's closed form (raymarching.cu:664-670) treats the ray as if it were NOT truncated — it is the exact
    # gradient when no early break happens, so compare on a batch that never reaches the threshold ...
    gws, gim = rng.standard_normal(N).astype(np.float32), rng.standard_normal((N, 3)).astype(np.float32)
    gs, gc = otr.composite_rays_train_backward(gws, gim, sig, rgb, deltas, rays, ws, image, T_thresh)
    loss = (tws * torch.tensor(gws, dtype=torch.float64)).sum() + (timage * torch.tensor(gim, dtype=torch.float64)).sum()
    loss.backward()
    # rgbs: exact everywhere (grad_image * weight)
    assert np.allclose(gc, tc.grad.numpy(), atol=2e-6)
    # sigmas: exact on rays that ran to their last sample
    for idx, off, n in rays.tolist():
        if n == 0:
            continue
        alpha = 1 - np.exp(-sig[off:off + n].astype(np.float64) * deltas[off:off + n, 0])
        if np.prod(1 - alpha) >= T_thresh:
            assert np.allclose(gs[off:off + n], ts.grad.numpy()[off:off + n], rtol=1e-4, atol=1e-5)
    # slots after an early break keep the zero the wrapper filled in
    for idx, off, n in rays.tolist():
        if n:
            alpha = 1 - np.exp(-sig[off:off + n].astype(np.float64) * deltas[off:off + n, 0])
            stop = np.flatnonzero(np.cumprod(1 - alpha) < T_thresh)
            if len(stop) and stop[0] + 1 < n:
                assert not gs[off + stop[0] + 1:off + n].any() and not gc[off + stop[0] + 1:off + n].any()


def test_composite_train_skips_rays_beyond_the_point_budget():
    rng = np.random.default_rng(5)
    rays, sig, rgb, deltas = _random_ray_batch(rng, 20, 30, with_empty=False)
    M = int(rays[10, 1] + 3)                                   # ray 10 and later do not fit
    ws, depth, image = otr.composite_rays_train_forward(sig[:M], rgb[:M], deltas[:M], rays, 1e-4)
    assert not ws[rays[10:, 0]].any() and (ws[rays[:10, 0]] > 0).all()
    gs, gc = otr.composite_rays_train_backward(np.ones(20, np.float32), np.ones((20, 3), np.float32), sig[:M], rgb[:M], deltas[:M], rays, ws, image, 1e-4)
    assert not gs[rays[10, 1]:].any() and gs[:rays[10, 1]].any()


def _torch_grid(x, emb, offsets, pls, base, interp):
    """Differentiable float64 trilinear / hash interpolation (index computed without gradient)."""
    L = len(offsets) - 1
    outs = []
    S = np.float32(np.log2(pls))
    for l in range(L):
        scale = float(np.float32(np.exp2(np.float32(l) * S) * np.float32(base) - np.float32(1.0)))
        res = int(np.ceil(scale)) + 1
        hs = int(offsets[l + 1] - offsets[l])
        pos = x * scale + 0.5
        g = torch.floor(pos.detach()).to(torch.int64)
        w = pos - g
        if interp == 1:
            w = w * w * (3 - 2 * w)
        acc = 0
        for c in range(8):
            bit = torch.tensor([(c >> k) & 1 for k in range(3)])
            gc = g + bit
            wc = torch.prod(torch.where(bit.bool(), w, 1 - w), dim=1)
            if (res + 1) ** 3 <= hs:
                idx = gc[:, 0] + gc[:, 1] * (res + 1) + gc[:, 2] * (res + 1) ** 2
            else:
                idx = (gc[:, 0] ^ ((gc[:, 1] * 2654435761) & 0xFFFFFFFF) ^ ((gc[:, 2] * 805459861) & 0xFFFFFFFF)) & 0xFFFFFFFF
            idx = idx % hs + int(offsets[l])
            acc = acc + wc[:, None] * emb[idx]
        outs.append(acc)
    return torch.cat(outs, 1)


@pytest.mark.parametrize("interp", [0, 1])
def test_grid_backward_and_dy_dx_vs_autograd(interp):
    from pienerf_amd.gridencoder.grid import level_table_offsets
    pls, base, L = 1.6, 8, 6
    offsets = level_table_offsets(3, L, pls, base, 12, False)          # small tables: levels 3+ are hashed with collisions
    rng = np.random.default_rng(6)
    emb = rng.uniform(-1, 1, (int(offsets[-1]), 2)).astype(np.float32)
    B = 300
    x = rng.uniform(0.02, 0.98, (B, 3)).astype(np.float32)
    x[:5] = [1.2, 0.5, 0.5]                                            # out of range: zero output, zero gradient
    grad = rng.standard_normal((B, L * 2)).astype(np.float32)
    tx = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    te = torch.tensor(emb, dtype=torch.float64, requires_grad=True)
    inside = torch.tensor((x >= 0).all(1) & (x <= 1).all(1))
    y = _torch_grid(tx, te, offsets, pls, base, interp) * inside[:, None]
    assert np.allclose(oracle.grid_encode_forward(x, emb, offsets, pls, base, 0, False, interp), y.detach().numpy(), atol=3e-5)
    (y * torch.tensor(grad, dtype=torch.float64)).sum().backward()
    dy_dx = otr.grid_encode_dy_dx(x, emb, offsets, pls, base, 0, False, interp)
    gi, ge = otr.grid_encode_backward(grad, x, emb.shape, offsets, pls, base, dy_dx, 0, False, interp)
    assert np.abs(ge - te.grad.numpy()).max() < 1e-4 * max(1.0, np.abs(te.grad.numpy()).max())
    # fp32 positions at scale ~100 carry ~1e-5 relative error in the interpolation weights; dy_dx multiplies table differences by scale
    assert np.abs(gi - tx.grad.numpy()).max() < 2e-3 * np.abs(tx.grad.numpy()).max()
    assert not gi[:5].any()
    gi2, ge2 = otr.grid_encode_backward(grad, x, emb.shape, offsets, pls, base, None, 0, False, interp)
    assert gi2 is None and np.array_equal(ge2, ge)


def test_grad_total_variation_vs_closed_form():
    """TV gradient of a fully dense level by hand: for the cell g of each sample, grad[g] += w/6 * sum_nbrs(e_g - e_nbr) / sqrt(sum (e_g - e_nbr)^2 + 1e-9)."""
    from pienerf_amd.gridencoder.grid import level_table_offsets
    pls, base, L = 2.0, 4, 2
    offsets = level_table_offsets(3, L, pls, base, 19, False)          # both levels dense
    rng = np.random.default_rng(7)
    emb = rng.uniform(-1, 1, (int(offsets[-1]), 2)).astype(np.float32)
    x = rng.uniform(0, 1, (200, 3)).astype(np.float32)
    x[0] = [0.0, 0.0, 0.0]                                             # cell 0: no left neighbours
    x[1] = [1.0, 1.0, 1.0]
    x[2] = [-0.1, 0.5, 0.5]                                            # out of range: ignored
    got = otr.grad_total_variation(x, emb, np.zeros_like(emb), offsets, pls, base, weight=0.3)
    want = np.zeros(emb.shape, np.float64)
    scales, ress = oracle.grid_level_params(L, pls, base)
    for l in range(L):
        scale, res = np.float32(scales[l]), int(ress[l])
        stride = res + 1
        for p in x:
            if (p < 0).any() or (p > 1).any():
                continue
            g = np.floor(np.float32(p * scale + np.float32(0.5))).astype(np.int64)
            cell = lambda q: int(offsets[l]) + int(q[0] + q[1] * stride + q[2] * stride * stride)
            r, idl = np.zeros(2), np.zeros(2)
            for d in range(3):
                for step, ok in ((1, g[d] < res), (-1, g[d] > 0)):
                    if ok:
                        q = g.copy()
                        q[d] += step
                        gv = emb[cell(g)].astype(np.float64) - emb[cell(q)]
                        r += gv
                        idl += gv * gv
            want[cell(g)] += 0.3 / 6 * r / np.sqrt(idl + 1e-9)
    assert np.abs(got - want).max() < 1e-5 and np.abs(want).max() > 0.01


def test_sh_dy_dx_and_backward_vs_autograd():
    rng = np.random.default_rng(8)
    d = rng.standard_normal((200, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d = d.astype(np.float32)
    t = torch.tensor(d, dtype=torch.float64, requires_grad=True)
    x, y, z = t[:, 0], t[:, 1], t[:, 2]
    pi = np.pi
    c = lambda v: v / np.sqrt(pi)
    sh = torch.stack([  # the closed forms quoted in shencoder.cu:50-68's comments, written out independently
        torch.full_like(x, c(0.5)), -c(np.sqrt(3) / 2) * y, c(np.sqrt(3) / 2) * z, -c(np.sqrt(3) / 2) * x,
        c(np.sqrt(15) / 2) * x * y, -c(np.sqrt(15) / 2) * y * z, c(np.sqrt(5) / 4) * (3 * z * z - 1), -c(np.sqrt(15) / 2) * x * z, c(np.sqrt(15) / 4) * (x * x - y * y),
        c(np.sqrt(70) / 8) * y * (-3 * x * x + y * y), c(np.sqrt(105) / 2) * x * y * z, c(np.sqrt(42) / 8) * y * (1 - 5 * z * z), c(np.sqrt(7) / 4) * z * (5 * z * z - 3),
        c(np.sqrt(42) / 8) * x * (1 - 5 * z * z), c(np.sqrt(105) / 4) * z * (x * x - y * y), c(np.sqrt(70) / 8) * x * (-x * x + 3 * y * y)], 1)
    assert np.allclose(oracle.sh_encode_forward(d, 4), sh.detach().numpy(), atol=2e-6)
    grad = rng.standard_normal((200, 16)).astype(np.float32)
    (sh * torch.tensor(grad, dtype=torch.float64)).sum().backward()
    dy_dx = otr.sh_encode_dy_dx(d, 4)
    J = torch.autograd.functional.jacobian(lambda v: torch.stack([  # one sample: [16, 3]
        torch.full_like(v[0], c(0.5)), -c(np.sqrt(3) / 2) * v[1], c(np.sqrt(3) / 2) * v[2], -c(np.sqrt(3) / 2) * v[0]]), torch.tensor(d[0], dtype=torch.float64))
    assert np.allclose(dy_dx[0].reshape(3, 16)[:, :4], J.numpy().T, atol=1e-6)
    gi = otr.sh_encode_backward(grad, dy_dx, 4)
    assert np.allclose(gi, t.grad.numpy(), atol=2e-5)
    for deg in (1, 2, 3):
        dd = otr.sh_encode_dy_dx(d, deg)
        assert np.array_equal(dd.reshape(-1, 3, deg * deg), dy_dx.reshape(-1, 3, 16)[:, :, :deg * deg])
