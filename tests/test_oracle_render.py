"""Pins the render half of the CPU oracle with independent maths (the reference ships no tests or golden vectors —
SURVEY.md §4/§8c — so each check below re-derives the expected value by a different route)."""
import numpy as np
import pytest

import oracle
from pienerf_amd import scene


def test_morton_matches_bit_interleave():
    rng = np.random.default_rng(0)
    xyz = rng.integers(0, 1024, size=(500, 3))
    want = np.zeros(500, np.uint32)
    for b in range(10):
        for c in range(3):
            want |= (((xyz[:, c] >> b) & 1) << (3 * b + c)).astype(np.uint32)
    assert np.array_equal(scene.morton3D(xyz[:, 0], xyz[:, 1], xyz[:, 2]), want)


def test_near_far_slab_test():
    rng = np.random.default_rng(1)
    N = 4000
    o = rng.uniform(-3, 3, (N, 3)).astype(np.float32)
    d = rng.standard_normal((N, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    aabb = np.array([-0.7, -0.9, -0.6, 0.8, 0.9, 0.5], np.float32)
    near, far = oracle.near_far_from_aabb(o, d, aabb, 0.2)
    # textbook slab test in float64
    t0 = (aabb[None, :3] - o.astype(np.float64)) / d
    t1 = (aabb[None, 3:] - o.astype(np.float64)) / d
    tn, tf = np.minimum(t0, t1).max(1), np.maximum(t0, t1).min(1)
    hit = tn <= tf
    FMAX = np.finfo(np.float32).max
    decided = np.abs(tn - tf) > 1e-4  # grazing rays may flip in fp32
    assert np.array_equal((near != FMAX)[decided], hit[decided])
    ok = hit & decided
    assert np.allclose(near[ok], np.maximum(tn[ok], 0.2), rtol=1e-4, atol=1e-5)
    assert np.allclose(far[ok], tf[ok], rtol=1e-4, atol=1e-5)
    assert np.all(far[~hit & decided] == FMAX)


def _numpy_grid_reference(x, emb, offsets, pls, base):
    """Independent 20-line trilinear / hash reference in float64 (instant-ngp indexing)."""
    L = len(offsets) - 1
    out = np.zeros((len(x), L, 2))
    S = np.float32(np.log2(pls))
    for l in range(L):
        scale = np.float32(np.exp2(np.float32(l) * S) * np.float32(base) - np.float32(1.0))
        res = int(np.ceil(scale)) + 1
        hs = int(offsets[l + 1] - offsets[l])
        pos = x.astype(np.float64) * float(scale) + 0.5
        g = np.floor(pos).astype(np.int64)
        w = pos - g
        for c in range(8):
            bit = np.array([(c >> k) & 1 for k in range(3)])
            gc = g + bit
            wc = np.prod(np.where(bit, w, 1 - w), axis=1)
            if (res + 1) ** 3 <= hs:
                idx = gc[:, 0] + gc[:, 1] * (res + 1) + gc[:, 2] * (res + 1) ** 2
            else:
                idx = (gc[:, 0] ^ (gc[:, 1] * 2654435761 & 0xFFFFFFFF) ^ (gc[:, 2] * 805459861 & 0xFFFFFFFF)) & 0xFFFFFFFF
            out[:, l] += wc[:, None] * emb[offsets[l] + idx % hs]
    return out.reshape(len(x), -1)


def test_grid_encode_vs_numpy(ckpt):
    rng = np.random.default_rng(2)
    x = rng.random((300, 3)).astype(np.float32) * 0.98 + 0.01
    got = oracle.grid_encode_forward(x, ckpt["embeddings"], ckpt["offsets"], ckpt["per_level_scale"], ckpt["base_resolution"])
    want = _numpy_grid_reference(x, ckpt["embeddings"], ckpt["offsets"], ckpt["per_level_scale"], ckpt["base_resolution"])
    # fine levels: pos ~ 2000 so one fp32 ulp of pos moves a weight by 1.2e-4; the weights multiply values |v| <= 0.5
    assert np.abs(got - want).max() < 2e-4
    assert np.abs(got[:, :10] - want[:, :10]).max() < 1e-5  # coarse levels are tight
    assert np.allclose(got[:, 0], 0.5, atol=1e-6)  # the constant channel of the synthetic checkpoint
    oob = oracle.grid_encode_forward(np.array([[1.5, 0.5, 0.5], [0.5, -0.01, 0.5]], np.float32), ckpt["embeddings"], ckpt["offsets"],
                                     ckpt["per_level_scale"], ckpt["base_resolution"])
    assert np.all(oob == 0)


def test_grid_offsets_and_level_params(ckpt):
    off = ckpt["offsets"]
    # chair geometry quoted in SURVEY.md §8a R10
    assert off[-1] == 6119864 and list(np.diff(off)[:5]) == [4920, 13824, 32768, 85184, 216000] and np.all(np.diff(off)[5:] == 524288)
    scales, res = oracle.grid_level_params(16, ckpt["per_level_scale"], 16)
    assert scales[0] == 15.0 and res[0] == 16
    assert abs(scales[15] - 2047.0) < 0.01 and res[15] in (2048, 2049)


def test_sh_vs_scipy():
    import warnings
    from scipy.special import sph_harm
    warnings.filterwarnings('ignore', category=DeprecationWarning)
    rng = np.random.default_rng(3)
    d = rng.standard_normal((200, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d = d.astype(np.float32).astype(np.float64)   # the basis is evaluated AT the float inputs
    got = oracle.sh_encode_forward(d.astype(np.float32), 8)
    assert np.array_equal(got[:, :16], oracle.sh_encode_forward(d.astype(np.float32), 4))
    r = np.linalg.norm(d, axis=1)                   # (unit to float rounding: the polynomials are homogeneous per band only on the sphere)
    theta = np.arctan2(d[:, 1], d[:, 0])  # azimuth
    phi = np.arccos(np.clip(d[:, 2] / r, -1, 1))  # polar
    k = 0
    for l in range(8):
        for m in range(-l, l + 1):
            Y = sph_harm(abs(m), l, theta, phi)
            if m < 0:
                real = np.sqrt(2) * Y.imag
            elif m == 0:
                real = Y.real
            else:
                real = np.sqrt(2) * Y.real
            # the reference's basis (shencoder.cu:50-68) is the real SH basis that KEEPS the Condon-Shortley phase (Y_1,-1 = -c y)
            assert np.abs(got[:, k] - real).max() < (2e-6 if l < 4 else 1e-5), (l, m)
            k += 1


def test_sh_gradients_are_the_derivatives_of_the_basis():
    """orc_sh_encode_dy_dx (shencoder.cu:125-355) for degree 1..8 against central differences of the forward in double-stepped float inputs, off the
    unit sphere too (the reference differentiates its polynomials as polynomials in x, y, z)."""
    rng = np.random.default_rng(8)
    d = rng.uniform(-1, 1, (300, 3)).astype(np.float32)
    for degree in (2, 4, 5, 8):
        C2 = degree * degree
        from oracle import training as otr
        g = otr.sh_encode_dy_dx(d, degree).reshape(-1, 3, C2)
        h = np.float32(2.0 ** -7)
        for ax in range(3):
            e = np.zeros(3, np.float32)
            e[ax] = h
            num = (oracle.sh_encode_forward(d + e, degree).astype(np.float64) - oracle.sh_encode_forward(d - e, degree)) / (2.0 * float(h))
            # central differences of a degree-7 polynomial with coefficients ~10: O(h^2) truncation ~ 1e-2 at worst on the top band
            assert np.abs(num - g[:, ax]).max() < (2e-3 if degree <= 4 else 6e-2), (degree, ax, np.abs(num - g[:, ax]).max())


def test_composite_closed_form():
    rng = np.random.default_rng(4)
    n_alive, n_step, N = 50, 8, 80
    alive = np.arange(n_alive, dtype=np.int32) + 5
    sig = rng.random(n_alive * n_step).astype(np.float32) * 20
    rgb = rng.random((n_alive * n_step, 3)).astype(np.float32)
    deltas = np.full((n_alive * n_step, 2), 0.01, np.float32)
    t0 = np.full(N, 2.0, np.float32)
    ws, dep, img = np.zeros(N, np.float32), np.zeros(N, np.float32), np.zeros((N, 3), np.float32)
    al = alive.copy()
    oracle.composite_rays(n_alive, n_step, al, t0, sig, rgb, deltas, ws, dep, img, T_thresh=0.0)
    alpha = 1 - np.exp(-sig.reshape(n_alive, n_step).astype(np.float64) * 0.01)
    T = np.cumprod(np.concatenate([np.ones((n_alive, 1)), 1 - alpha[:, :-1]], 1), 1)
    w = alpha * T
    assert np.allclose(ws[alive], 1 - np.prod(1 - alpha, 1), atol=1e-5)
    assert np.allclose(img[alive], (w[..., None] * rgb.reshape(n_alive, n_step, 3)).sum(1), atol=1e-5)
    tt = 2.0 + 0.01 * np.arange(1, n_step + 1)
    assert np.allclose(dep[alive], (w * tt).sum(1), atol=1e-4)
    assert np.all(al == alive) and np.allclose(t0[alive], 2.0 + 0.01 * n_step, atol=1e-5)  # all rays survive, t advanced
    assert np.all(ws[:5] == 0)


def test_composite_termination_rules():
    # deltas == 0 ends the ray; T < T_thresh is checked AFTER accumulating the sample (raymarching.cu:867,903)
    alive = np.array([0, 1, 2], np.int32)
    n_step = 4
    sig = np.array([1000, 1, 1, 1, 5, 5, 5, 5, 5, 5, 0, 0], np.float32)
    deltas = np.tile(np.array([[0.01, 0.01]], np.float32), (12, 1))
    deltas[10:] = 0  # ray 2 emitted only two samples
    rgb = np.ones((12, 3), np.float32)
    t = np.ones(3, np.float32)
    ws, dep, img = np.zeros(3, np.float32), np.zeros(3, np.float32), np.zeros((3, 3), np.float32)
    oracle.composite_rays(3, n_step, alive, t, sig, rgb, deltas, ws, dep, img, T_thresh=1e-2)
    assert list(alive) == [-1, 1, -1]
    a0, a1 = 1 - np.exp(-10.0), 1 - np.exp(-0.01)
    assert abs(ws[0] - (a0 + (1 - a0) * a1)) < 1e-6  # opaque first sample, then exactly one more sample is accumulated
    assert abs(t[1] - 1.04) < 1e-6 and t[0] == 1 and t[2] == 1


def test_compaction_is_stable_filter():
    rng = np.random.default_rng(5)
    for n in (0, 1, 7, 1000):
        a = rng.integers(0, 10 ** 6, n).astype(np.int32)
        a[rng.random(n) < 0.5] = -1
        assert np.array_equal(oracle.compact_rays(a), a[a >= 0])


def test_pnts_in_grids_counting_sort():
    rng = np.random.default_rng(6)
    p = rng.uniform(-0.5, 0.5, (700, 3)).astype(np.float32)
    hgs = np.float32(0.06)
    bbmin, bbmax, res = oracle.render_bbox(p, hgs)
    n_grid = int(res.prod())
    cnt, bgn, idx = oracle.get_pnts_in_grids(len(p), n_grid, p, bbmin, bbmax, hgs, res)
    g = np.floor((p - bbmin) / hgs).astype(np.int64)
    gid = g[:, 2] * res[1] * res[0] + g[:, 1] * res[0] + g[:, 0]
    assert np.array_equal(cnt, np.bincount(gid, minlength=n_grid))
    assert np.array_equal(bgn, np.cumsum(cnt) - cnt)
    assert np.array_equal(np.sort(idx), np.arange(len(p)))
    assert np.array_equal(gid[idx], np.sort(gid, kind="stable"))       # grouped by cell ...
    assert np.array_equal(idx, np.argsort(gid, kind="stable"))          # ... ascending id inside a cell


def _quadratic_map(F, dF, q):
    """phi(q) with the kernel's own flat-index arithmetic (raymarching.cu:940-951,1278-1296): A(q) = F + D(q),
    D[m] = sum_b dF[b*9+m] q_b, phi = mul31(F, q) + 0.5 mul31(D, q) where mul31(M, v)[r] = sum_c M[c*3+r] v_c."""
    D = sum(dF[b * 9:(b + 1) * 9] * q[b] for b in range(3))
    m31 = lambda M, v: np.array([M[0] * v[0] + M[3] * v[1] + M[6] * v[2], M[1] * v[0] + M[4] * v[1] + M[7] * v[2], M[2] * v[0] + M[5] * v[1] + M[8] * v[2]])
    return m31(F, q) + 0.5 * m31(D, q)


def test_newton_warp_round_trip():
    rng = np.random.default_rng(7)
    for trial in range(20):
        F = (np.eye(3) + 0.2 * rng.standard_normal((3, 3))).T.reshape(9)
        dF = 0.5 * rng.standard_normal(27)
        p_ori, p_def = rng.uniform(-0.5, 0.5, 3), rng.uniform(-0.5, 0.5, 3)
        q = rng.uniform(-0.04, 0.04, 3)
        x = p_def + _quadratic_map(F, dF, q)
        p5, rej = oracle.warp_point(x, p_ori, p_def, F, dF, 8, 0.0525)
        assert not rej and np.abs(p5 - (p_ori + q)).max() < 2e-6  # converged Newton recovers the rest point
        p1, _ = oracle.warp_point(x, p_ori, p_def, F, dF, 1, 0.0525)
        lin = p_ori + np.linalg.solve(np.array(F).reshape(3, 3).T, x - p_def)  # one iteration from q=0 is the linear inverse
        assert np.abs(p1 - lin).max() < 2e-6
    far, rej = oracle.warp_point(p_def + np.array([0.2, 0, 0]), p_ori, p_def, np.eye(3).reshape(9), np.zeros(27), 1, 0.0525)
    assert rej  # outside the IP's trust region -> rejected (raymarching.cu:1316-1319)
    same, rej = oracle.warp_point(p_def, p_ori, p_def, np.zeros(9), np.zeros(27), 3, 0.0525)
    assert not rej and np.allclose(same, p_ori)  # singular A: A_inv stays 0, dq = 0, exits on the 1e-12 test (quirk R7q-ii)


def _rest_state(n_side=8, dx=0.05):
    g = (np.arange(n_side) - n_side / 2 + 0.5) * dx
    p = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3).astype(np.float32)
    n = len(p)
    return dict(p_def=p, p_ori=p.copy(), F=np.tile(np.eye(3, dtype=np.float32).reshape(1, 9), (n, 1)), dF=np.zeros((n, 27), np.float32), IP_dx=dx * 1.05)


@pytest.mark.parametrize("num_seek_IP", [1, 2, 3])
def test_march_at_rest_is_plain_ray_marching(num_seek_IP):
    """With p_def = p_ori, F = I, dF = 0 the warp is the identity, so emitted points lie on the ray, inside occupied voxels,
    dt = dt_min apart, and deltas[1] telescopes to the distance marched."""
    ip = _rest_state()
    H = 128
    bits = np.full(H ** 3 // 8, 0xFF, np.uint8)  # fully occupied grid: every found sample is emitted
    hgs = np.float32(0.06)
    bbmin, bbmax, res = oracle.render_bbox(ip["p_def"], hgs)
    n_grid = int(res.prod())
    pig = oracle.get_pnts_in_grids(len(ip["p_def"]), n_grid, ip["p_def"], bbmin, bbmax, hgs, res)
    rng = np.random.default_rng(8)
    N = 64
    o = np.tile(np.array([[0.01, 0.02, 3.0]], np.float32), (N, 1))
    tgt = rng.uniform(-0.12, 0.12, (N, 3)).astype(np.float32)
    d = tgt - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    nears, fars = oracle.near_far_from_aabb(o, d, np.concatenate([bbmin, bbmax]), 0.2)
    alive = np.arange(N, dtype=np.int32)
    n_step = 8
    xyz, dirs, deltas = oracle.march_rays_quadratic_bending(*pig, len(ip["p_def"]), n_grid, ip["p_def"], ip["p_ori"], ip["F"], ip["dF"], 2, bbmin, bbmax,
                                                            hgs, res, num_seek_IP, np.float32(ip["IP_dx"]), False, np.zeros(6, np.float32), N, n_step,
                                                            alive, nears, o, d, 1.0, bits, 1, H, nears, fars, 128)
    assert not oracle.march_rays_quadratic_bending.last_oob
    xyz, deltas, dirs = xyz[:N * n_step].reshape(N, n_step, 3), deltas[:N * n_step].reshape(N, n_step, 2), dirs[:N * n_step].reshape(N, n_step, 3)
    assert np.all(deltas[..., 0] > 0), "every ray crosses the IP block and the grid is full"
    dt_min = np.float32(2 * np.sqrt(3) / 1024)
    assert np.allclose(deltas[..., 0], dt_min, rtol=1e-6)
    t = nears[:, None] + np.cumsum(deltas[..., 1], axis=1) - deltas[..., 0]  # sample parameter
    on_ray = o[:, None, :] + t[..., None] * d[:, None, :]
    assert np.abs(xyz - on_ray).max() < 5e-6  # identity warp (Newton on F = I)
    assert np.array_equal(dirs, np.broadcast_to(d[:, None, :], dirs.shape))


def test_march_respects_density_bitfield_and_skips(ckpt, deformed_ip_state, small_opt):
    """Emitted rest-space points fall in occupied voxels of the morton-ordered bitfield; nothing is emitted for rays that miss."""
    ip = deformed_ip_state
    W = 40
    o, d = oracle.get_rays(scene.orbit_pose(5.0, 25.0, -15.0), scene.orbit_intrinsics(W, W, 50.0), W, W)
    hgs = np.float32(small_opt["hash_grid_size"])
    bbmin, bbmax, res = oracle.render_bbox(ip["p_def"], hgs)
    n_grid = int(res.prod())
    pig = oracle.get_pnts_in_grids(len(ip["p_def"]), n_grid, ip["p_def"], bbmin, bbmax, hgs, res)
    nears, fars = oracle.near_far_from_aabb(o, d, np.concatenate([bbmin, bbmax]), 0.2)
    alive = np.arange(W * W, dtype=np.int32)
    xyz, _, deltas = oracle.march_rays_quadratic_bending(*pig, len(ip["p_def"]), n_grid, ip["p_def"], ip["p_ori"], ip["F"], ip["dF"], 1, bbmin, bbmax, hgs,
                                                         res, 3, np.float32(ip["IP_dx"]), False, np.zeros(6, np.float32), W * W, 4, alive, nears, o, d,
                                                         1.0, ckpt["density_bitfield"], 1, 128, nears, fars, 128)
    em = deltas[:, 0] != 0
    assert em.sum() > 100
    p = xyz[em]
    n = np.clip((0.5 * (p.astype(np.float64) + 1) * 128).astype(np.int64), 0, 127)
    m = scene.morton3D(n[:, 0], n[:, 1], n[:, 2]).astype(np.int64)
    assert np.all((ckpt["density_bitfield"][m // 8] >> (m % 8)) & 1)
    miss = np.repeat(nears > 1e30, 4)
    assert not em[:W * W * 4][miss].any()


def test_render_frame_properties(ckpt, deformed_ip_state, small_opt):
    ip = deformed_ip_state
    W = 48
    o, d = oracle.get_rays(scene.orbit_pose(5.0, 20.0, -15.0), scene.orbit_intrinsics(W, W, 50.0), W, W)
    r1 = oracle.render_deformed(o, d, ip, ckpt, small_opt)
    r2 = oracle.render_deformed(o, d, ip, ckpt, small_opt)
    assert np.array_equal(r1["image"], r2["image"])  # deterministic (OpenMP only partitions independent rays)
    ws = r1["weights_sum"]
    assert np.all(ws >= 0) and np.all(ws <= 1 + 1e-5)
    assert np.all(r1["image"] >= 0) and np.all(r1["image"] <= 1 + 1e-5)
    assert np.all(r1["image"][ws == 0] == 1.0)  # background = 1 where nothing was hit (renderer.py:803-804,896)
    assert (ws > 0.5).sum() > 20 and r1["trips"] >= 3 and r1["samples"] > 1000
    # T_thresh: a hit ray stops once transmittance < 1e-2, so alpha saturates just above 0.99
    assert np.percentile(ws[ws > 0.5], 50) > 0.9
    # depth is NaN exactly where the ray misses the IP bbox (renderer.py:898)
    bbmin, bbmax, _ = oracle.render_bbox(ip["p_def"], np.float32(small_opt["hash_grid_size"]))
    nears, _ = oracle.near_far_from_aabb(o, d, np.concatenate([bbmin, bbmax]), 0.2)
    assert np.array_equal(np.isnan(r1["depth"]), nears > 1e30)
    # num_seek_IP = 1 takes the find_closest_IP path (own cell first): still a picture of the same object
    r3 = oracle.render_deformed(o, d, ip, ckpt, dict(small_opt, num_seek_IP=1))
    both = (r3["weights_sum"] > 0.5) & (ws > 0.5)
    assert both.sum() > 0.6 * (ws > 0.5).sum()


def test_get_rays_matches_reference_formulation():
    W, H = 30, 20
    pose = scene.orbit_pose(5.0, 33.0, -12.0)
    intr = scene.orbit_intrinsics(W, H, 50.0)
    o, d = oracle.get_rays(pose, intr, H, W)
    j, i = np.meshgrid(np.arange(H) + 0.5, np.arange(W) + 0.5, indexing="ij")  # row-major pixels, centres at +0.5 (nerf/utils.py:72-74)
    dirs = np.stack([(i - intr[2]) / intr[0], (j - intr[3]) / intr[1], np.ones_like(i)], -1).reshape(-1, 3)
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    assert np.abs(d - dirs @ pose[:3, :3].T).max() < 1e-6
    assert np.all(o == pose[:3, 3][None, :])
    assert abs(np.linalg.norm(pose[:3, 3]) - 5.0) < 1e-5 and abs(intr[0] - H / (2 * np.tan(np.radians(25.0)))) < 1e-9
    # default OrbitCamera: at +z looking down -z, y up (nerf/gui.py:13-44)
    p0 = scene.orbit_pose(5.0)
    assert np.allclose(p0[:3, 3], [0, 0, 5]) and np.allclose(p0[:3, :3], np.diag([1, -1, -1]))


# ----------------------------------------------------------------------------- static inference ops
def test_packbits_and_morton_invert_vs_numpy():
    rng = np.random.default_rng(11)
    grid = rng.random((2, 32 ** 3)).astype(np.float32)
    grid[1, :9] = 0.25  # ties: strict > (raymarching.cu:291)
    bits = oracle.packbits(grid, 0.25)
    assert np.array_equal(bits, np.packbits((grid > 0.25).reshape(-1), bitorder="little"))
    c = rng.integers(0, 1024, size=(3000, 3)).astype(np.int32)
    idx = oracle.morton3D(c)
    assert np.array_equal(idx.astype(np.uint32), scene.morton3D(c[:, 0], c[:, 1], c[:, 2]))
    assert np.array_equal(oracle.morton3D_invert(idx), c)


@pytest.mark.parametrize("dt_gamma,cascade", [(0.0, 1), (1.0 / 128, 2)])
def test_static_march_properties(dt_gamma, cascade):
    """Independent route: emitted points lie on the (clamped) ray inside occupied voxels of the level the kernel picks, dt follows
    clamp(t*dt_gamma), deltas[1] telescopes to the distance marched, and rays end with zero rows."""
    bound = float(2 ** (cascade - 1))
    ck = scene.make_checkpoint(bound=bound, seed=3)
    assert ck["cascade"] == cascade
    H, max_steps, n_step, W = ck["grid_size"], 512, 16, 36
    o, d = oracle.get_rays(scene.orbit_pose(3.2 * bound, 20.0, -25.0), scene.orbit_intrinsics(W, W, 50.0), W, W)
    aabb = np.array([-bound] * 3 + [bound] * 3, np.float32)
    nears, fars = oracle.near_far_from_aabb(o, d, aabb, 0.2)
    alive = np.arange(W * W, dtype=np.int32)[::2].copy()
    xyz, dirs, deltas = oracle.march_rays(len(alive), n_step, alive, nears, o, d, bound, ck["density_bitfield"], cascade, H, nears, fars, 128, None,
                                          dt_gamma, max_steps)
    M = len(alive) * n_step
    assert xyz.shape[0] % 128 == 0 and xyz.shape[0] > M and not xyz[M:].any() and not deltas[M:].any()
    xyz, dirs, deltas = xyz[:M].reshape(-1, n_step, 3), dirs[:M].reshape(-1, n_step, 3), deltas[:M].reshape(-1, n_step, 2)
    em = deltas[..., 0] != 0
    assert em.sum() > 200
    # emitted rows are a prefix of each ray's slots
    assert np.all(em[:, :-1] >= em[:, 1:])
    dt_min, dt_max = 2 * np.sqrt(3) / max_steps, 2 * np.sqrt(3) * 2 ** (cascade - 1) / H
    # sample parameter: t_k = near + sum_{j<=k} deltas[j,1] - deltas[k,0]
    t = nears[alive][:, None] + np.cumsum(deltas[..., 1].astype(np.float64), axis=1) - deltas[..., 0]
    want_dt = np.clip(t * dt_gamma, dt_min, dt_max)
    assert np.allclose(deltas[..., 0][em], want_dt[em], rtol=1e-5)
    on_ray = np.clip(o[alive][:, None, :] + t[..., None] * d[alive][:, None, :], -bound, bound)
    assert np.abs(xyz - on_ray)[em].max() < 2e-5
    assert np.array_equal(dirs[em], np.broadcast_to(d[alive][:, None, :], dirs.shape)[em])
    # occupancy at the level max(mip_from_pos, mip_from_dt) (raymarching.cu:38-57)
    p = xyz[em].astype(np.float64)
    mx = np.abs(p).max(1)
    lvl_pos = np.clip(np.ceil(np.log2(np.maximum(mx, 1e-30))), 0, cascade - 1).astype(np.int64)
    lvl_dt = np.clip(np.ceil(np.log2(np.maximum(deltas[..., 0][em].astype(np.float64) * H * 0.5, 1e-30))), 0, cascade - 1).astype(np.int64)
    lvl = np.maximum(lvl_pos, lvl_dt)
    mb = np.minimum(2.0 ** lvl, bound)
    n = np.clip((0.5 * (p / mb[:, None] + 1) * H).astype(np.int64), 0, H - 1)
    m = lvl * H ** 3 + scene.morton3D(n[:, 0], n[:, 1], n[:, 2]).astype(np.int64)
    occ = (ck["density_bitfield"][m // 8] >> (m % 8)) & 1
    assert occ.mean() > 0.999  # a point within an ulp of a level / voxel boundary may round the other way in float64
    # rays that miss the box emit nothing
    assert not em[nears[alive] > 1e30].any()


def test_render_static_properties():
    ck = scene.make_checkpoint(bound=1.0, seed=0, shaped=True)
    opt = scene.default_opt(W=40, H=40)
    o, d = oracle.get_rays(scene.orbit_pose(4.0, 40.0, -20.0), scene.orbit_intrinsics(40, 40, 50.0), 40, 40)
    r1, r2 = oracle.render_static(o, d, ck, opt), oracle.render_static(o, d, ck, opt)
    assert np.array_equal(r1["image"], r2["image"])
    ws = r1["weights_sum"]
    assert np.all(ws >= 0) and np.all(ws <= 1 + 1e-5) and r1["trips"] >= 2 and r1["samples"] > 500
    assert np.all(r1["image"][ws == 0] == 1.0) and (ws > 0.9).sum() > 20
    nears, _ = oracle.near_far_from_aabb(o, d, np.array([-1, -1, -1, 1, 1, 1], np.float32), 0.2)
    assert np.array_equal(np.isnan(r1["depth"]), nears > 1e30)  # 0/0 exactly where the ray misses the +-bound box (renderer.py:384)


def test_sph_from_ray_against_the_closed_form():
    """oracle.sph_from_ray (raymarching.cu:165-202) against a float64 evaluation of the same geometry: the exit point lies on the sphere, theta is the
    polar angle from +y, phi the azimuth in the x-z plane, both scaled to [-1, 1]."""
    import oracle
    rng = np.random.default_rng(3)
    o = rng.uniform(-0.8, 0.8, (500, 3)).astype(np.float32)
    d = rng.normal(size=(500, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    R = 2.0
    c = oracle.sph_from_ray(o, d, R)
    o64, d64 = o.astype(np.float64), d.astype(np.float64)
    A, B, Cq = (d64 * d64).sum(1), (o64 * d64).sum(1), (o64 * o64).sum(1) - R * R
    t = (-B + np.sqrt(B * B - A * Cq)) / A
    p = o64 + t[:, None] * d64
    assert np.abs(np.linalg.norm(p, axis=1) - R).max() < 1e-9
    theta, phi = np.arctan2(np.hypot(p[:, 0], p[:, 2]), p[:, 1]), np.arctan2(p[:, 2], p[:, 0])
    assert np.abs(c[:, 0] - (2 * theta / np.pi - 1)).max() < 5e-6 and np.abs(c[:, 1] - phi / np.pi).max() < 5e-6
    assert c.min() >= -1.0 and c.max() <= 1.0


def test_grid_nd_oracle_restates_the_d3_oracle_and_interpolates():
    """oracle/grid_nd_oracle.cpp (kernel_grid<float, D, C> for D = 2..5, gridencoder.cu:386-399) at D = 3 equals render_oracle.cpp's D = 3 code bit for
    bit (forward, dy_dx, backward, both grid types, align_corners, smoothstep); for D = 2, 4, 5 a constant table encodes to that constant (the 2^D
    weights sum to one) with zero input gradient, out-of-range inputs to zero, and the backward scatters exactly the incoming gradient's mass."""
    from oracle import training as otr
    from pienerf_amd.gridencoder.grid import level_table_offsets
    rng = np.random.default_rng(11)
    pls = 1.5
    for gt in (0, 1):
        for al in (False, True):
            for ip in (0, 1):
                off = level_table_offsets(3, 6, pls, 8, 12, al)
                emb = rng.uniform(-.5, .5, (int(off[-1]), 2)).astype(np.float32)
                x = rng.uniform(-0.05, 1.05, (300, 3)).astype(np.float32)
                a = oracle.grid_encode_forward(x, emb, off, pls, 8, gt, al, ip)
                b, dd = oracle.grid_nd_forward(x, emb, off, pls, 8, gt, al, ip, dy_dx=True)
                d0 = otr.grid_encode_dy_dx(x, emb, off, pls, 8, gt, al, ip)
                g = rng.standard_normal(a.shape).astype(np.float32)
                gi0, ge0 = otr.grid_encode_backward(g, x, emb.shape, off, pls, 8, d0, gt, al, ip)
                gi1, ge1 = oracle.grid_nd_backward(g, x, emb.shape, off, pls, 8, dd, gt, al, ip)
                assert np.array_equal(a, b) and np.array_equal(dd, d0) and np.array_equal(gi0, gi1) and np.array_equal(ge0, ge1), (gt, al, ip)
                if ip == 0:
                    assert np.array_equal(otr.grad_total_variation(x, emb, np.zeros_like(emb), off, pls, 8, 0.3, gt, al), oracle.grid_nd_grad_tv(x, emb, off, 0.3, pls, 8, gt, al))
    for D in (2, 4, 5):
        off = level_table_offsets(D, 5, pls, 4, 11, False)
        emb = np.full((int(off[-1]), 2), 0.75, np.float32)
        x = rng.uniform(0, 1, (200, D)).astype(np.float32)
        x[0, 0] = 1.5
        y, dd = oracle.grid_nd_forward(x, emb, off, pls, 4, 0, False, 0, dy_dx=True)
        assert not y[0].any() and np.abs(y[1:] - 0.75).max() < 1e-6 and np.abs(dd).max() < 1e-4
        g = np.ones_like(y)
        _, ge = oracle.grid_nd_backward(g, x, emb.shape, off, pls, 4, None, 0, False, 0)
        assert abs(ge.sum() - 199 * 5 * 2) < 1e-2
