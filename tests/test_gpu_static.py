"""SURVEY §8(f) rank 3, inference side: the undeformed march, packbits, morton3D(_invert) and NeRFRenderer.run_cuda against the oracle's
restatements (bit-exact for the integer / march work, 1e-4 for the image)."""
import numpy as np
import pytest
import torch

import oracle
from pienerf_amd import raymarching, scene
from test_gpu_parity import DEV, T

pytestmark = pytest.mark.gpu


def test_morton_and_packbits_bit_exact():
    rng = np.random.default_rng(2)
    c = rng.integers(0, 128, size=(10007, 3)).astype(np.int32)
    idx = raymarching.morton3D(torch.from_numpy(c).to(DEV))
    assert np.array_equal(idx.cpu().numpy(), oracle.morton3D(c))
    back = raymarching.morton3D_invert(idx)
    assert np.array_equal(back.cpu().numpy(), c) and np.array_equal(oracle.morton3D_invert(idx.cpu().numpy()), c)
    # the bit-loop definition: bit 3k+a of the index is bit k of coordinate a
    ref = np.zeros(len(c), np.int64)
    for k in range(7):
        for a in range(3):
            ref |= ((c[:, a].astype(np.int64) >> k) & 1) << (3 * k + a)
    assert np.array_equal(idx.cpu().numpy().astype(np.int64), ref)
    grid = rng.random((2, 64 ** 3)).astype(np.float32)
    grid[0, :17] = 0.5                                               # ties are NOT above the threshold (strict >)
    bits = raymarching.packbits(torch.from_numpy(grid).to(DEV), 0.5)
    assert bits.dtype == torch.uint8 and bits.shape[0] == 2 * 64 ** 3 // 8
    assert np.array_equal(bits.cpu().numpy(), oracle.packbits(grid, 0.5))
    assert np.array_equal(np.unpackbits(bits.cpu().numpy(), bitorder="little").astype(bool), (grid > 0.5).reshape(-1))
    into = torch.zeros_like(bits)
    assert raymarching.packbits(torch.from_numpy(grid).to(DEV), 0.5, into) is into and torch.equal(into, bits)


@pytest.mark.parametrize("bound,dt_gamma,n_step,max_steps", [(1.0, 0.0, 1, 1024), (1.0, 0.0, 8, 1024), (2.0, 1.0 / 128, 8, 300), (1.0, 1.0 / 64, 64, 512)])
def test_march_rays_bit_exact(bound, dt_gamma, n_step, max_steps):
    ck = scene.make_checkpoint(bound=bound, seed=1)
    W = 72
    o, d = oracle.get_rays(scene.orbit_pose(3.5 * bound, 30.0, -15.0), scene.orbit_intrinsics(W, W, 50.0), W, W)
    aabb = np.array([-bound] * 3 + [bound] * 3, np.float32)
    nears, fars = oracle.near_far_from_aabb(o, d, aabb, 0.2)
    alive = np.arange(0, W * W, 3, dtype=np.int32)                   # a strided subset: slot n != ray id
    rays_t = nears.copy()
    ref = oracle.march_rays(len(alive), n_step, alive, rays_t, o, d, bound, ck["density_bitfield"], ck["cascade"], ck["grid_size"], nears, fars,
                            128, None, dt_gamma, max_steps)
    got = raymarching.march_rays(len(alive), n_step, T(alive), T(rays_t), T(o), T(d), bound, T(ck["density_bitfield"]), ck["cascade"],
                                 ck["grid_size"], T(nears), T(fars), 128, False, dt_gamma, max_steps)
    assert (ref[2][:, 0] != 0).sum() > 300
    for a, b in zip(got, ref):
        assert a.shape[0] % 128 == 0 and np.array_equal(a.cpu().numpy(), b)


def test_static_render_matches_oracle():
    from pienerf_amd.nerf.network import NeRFNetwork
    ck = scene.make_checkpoint(bound=1.0, seed=0, shaped=True)
    opt = scene.default_opt(W=80, H=60)
    o, d = oracle.get_rays(scene.orbit_pose(4.0, 40.0, -20.0), scene.orbit_intrinsics(80, 60, 50.0), 60, 80)
    ref = oracle.render_static(o, d, ck, opt)
    net = NeRFNetwork(encoding="hashgrid", bound=1.0, cuda_ray=True).to(DEV).load_checkpoint_dict(ck).eval()
    out = net.run_cuda(T(o)[None], T(d)[None], dt_gamma=opt["dt_gamma"], max_steps=opt["max_steps"], T_thresh=opt["T_thresh"])
    st = net.last_stats
    assert ref["samples"] > 3000 and st["samples"] == ref["samples"] and st["trips"] == ref["trips"] and st["alive_at_exit"] == 0
    assert np.abs(out["image"][0].cpu().numpy() - ref["image"]).max() < 1e-4
    assert np.abs(out["weights_sum"].cpu().numpy() - ref["weights_sum"]).max() < 1e-4
    dep, rd = out["depth"][0].cpu().numpy(), ref["depth"]
    assert np.array_equal(np.isfinite(dep), np.isfinite(rd)) and np.abs(dep[np.isfinite(dep)] - rd[np.isfinite(rd)]).max() < 1e-4
    # with the shaped density field the silhouette is the chair: opaque inside, background elsewhere
    ws = out["weights_sum"].cpu().numpy()
    assert (ws > 0.95).mean() > 0.03 and (ws < 1e-3).mean() > 0.5
