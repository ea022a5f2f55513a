"""Drop-in under the reference's own import names (shim/ first on sys.path): `import _raymarching as _backend`, `_gridencoder`, `_shencoder`,
`raymarching`, `gridencoder`, `shencoder`, `from simulator.solver import Simulator`.  The deformed render is driven exactly the way the reference's
Python does it — raymarching/raymarching.py's wrappers allocate and zero the outputs and call `_backend.<name>(...)` positionally (:21-51, 362-441),
nerf/renderer.py:755-907 runs the loop — and must reproduce the fused frame of pienerf_amd bit for bit."""
import os
import sys

import numpy as np
import pytest
import torch

import oracle
from conftest import ROOT
from pienerf_amd import scene

pytestmark = pytest.mark.gpu
DEV = "cuda"
SHIM = os.path.join(ROOT, "shim")


@pytest.fixture()
def shim_path():
    sys.path.insert(0, SHIM)
    yield
    sys.path.remove(SHIM)
    for k in [k for k in sys.modules if k.split(".")[0] in ("_raymarching", "_gridencoder", "_shencoder", "raymarching", "gridencoder", "shencoder", "simulator")]:
        del sys.modules[k]


def test_reference_call_sequence_through_the_backend_modules(shim_path, small_cloud, small_opt, ckpt):
    import _gridencoder
    import _raymarching as _backend
    import _shencoder
    import gridencoder
    import raymarching
    import shencoder
    from simulator.solver import Simulator
    assert raymarching.__file__.startswith(SHIM) and gridencoder.GridEncoder is not None and shencoder.SHEncoder is not None
    from pienerf_amd.nerf.network import NeRFNetwork
    from pienerf_amd.nerf.utils import get_pnts_in_grids, get_rays
    o = dict(small_opt, W=48, H=48)
    # main_gui.py:39-56
    sim = Simulator(dt=o["sim_dt"], iters=o["sim_iters"], bbox=torch.tensor([2.0 * o["bound"]] * 3), dx=o["sim_dx"], stiff=o["sim_stiff"],
                    base=torch.tensor([-o["bound"]] * 3), device=DEV)
    c = small_cloud
    sim.InitializeFromArrays(c["pos"], c["mass"], c["mu"], c["lam"], c["pin"])
    p_ori, _, _ = sim.get_IP_info()
    sim.update_force(sim.n_IP // 2, torch.tensor([300.0, 100.0, -200.0]))
    for _ in range(8):
        sim.stepforward()
    p_def, F_IP, dF_IP = sim.get_IP_info()
    net = NeRFNetwork(encoding="hashgrid", bound=1.0, cuda_ray=True).to(DEV).load_checkpoint_dict(ckpt)
    net.p_def, net.p_ori, net.IP_F, net.IP_dF, net.IP_dx = p_def, p_ori, F_IP, dF_IP, sim.dx * 1.05
    N = o["W"] * o["H"]
    rays = get_rays(torch.from_numpy(scene.orbit_pose(o["radius"], 20.0, -10.0))[None].to(DEV), scene.orbit_intrinsics(o["W"], o["H"], o["fovy"]), o["H"], o["W"])
    rays_o, rays_d = rays["rays_o"].view(-1, 3), rays["rays_d"].view(-1, 3)
    with torch.no_grad():
        want = net.render_deformed(rays["rays_o"], rays["rays_d"], **o)
        # ---- nerf/renderer.py:782-829 on the backend modules
        hgs = o["hash_grid_size"]
        bbmin = p_def.min(0).values - 1e-3 * torch.ones(3, device=DEV)
        bbmax = p_def.max(0).values + 1e-3 * torch.ones(3, device=DEV)
        resolution = torch.ceil((bbmax - bbmin) / hgs).to(torch.int32)
        aabb = torch.cat((bbmin, bbmax), 0)
        nears, fars = torch.empty(N, device=DEV), torch.empty(N, device=DEV)
        _backend.near_far_from_aabb(rays_o, rays_d, aabb, N, net.min_near, nears, fars)
        n_vtx, n_grid = p_ori.shape[0], int(resolution.prod())
        pig_cnt, pig_bgn, pig_idx = get_pnts_in_grids(n_vtx, n_grid, p_def, bbmin, bbmax, hgs, resolution)
        weights_sum, depth, image = torch.zeros(N, device=DEV), torch.zeros(N, device=DEV), torch.zeros(N, 3, device=DEV)
        rays_alive, rays_t = torch.arange(N, dtype=torch.int32, device=DEV), nears.clone()
        cut_bounds = torch.tensor(o["cut_bounds"], dtype=torch.float32, device=DEV)
        enc, sig_net, col_net = net.encoder, net.sigma_net, net.color_net
        S = np.log2(enc.per_level_scale)
        step = 0
        while step < o["max_steps"]:
            n_alive = rays_alive.shape[0]
            if n_alive <= 0:
                break
            n_step = max(min(N // n_alive, 8), 1)
            M = n_alive * n_step
            M += 128 - (M % 128)                                            # raymarching.py:410-413
            xyzs, dirs, deltas = torch.zeros(M, 3, device=DEV), torch.zeros(M, 3, device=DEV), torch.zeros(M, 2, device=DEV)
            noises = torch.zeros(n_alive, device=DEV)
            _backend.march_rays_quadratic_bending(pig_cnt, pig_bgn, pig_idx, n_vtx, n_grid, p_def, p_ori, F_IP, dF_IP, o["max_iter_num"], bbmin, bbmax, hgs, resolution,
                                                  o["num_seek_IP"], net.IP_dx, False, cut_bounds, n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, net.bound,
                                                  o["dt_gamma"], o["max_steps"], net.cascade, net.grid_size, net.density_bitfield, nears, fars, xyzs, dirs, deltas, noises)
            # ---- nerf/network.py:98-127 on the encoder backends (grid.py:46-57: [L,B,C] then permute; sphere_harmonics.py:26-30)
            u = ((xyzs + net.bound) / (2 * net.bound)).contiguous()
            feat = torch.empty(16, M, 2, device=DEV)
            _gridencoder.grid_encode_forward(u, enc.embeddings.detach(), enc.offsets, feat, M, 3, 2, 16, S, 16, None, 0, False, 0)
            h = feat.permute(1, 0, 2).reshape(M, 32)
            h = torch.relu(h @ sig_net[0].weight.t()) @ sig_net[1].weight.t()
            sigmas = torch.exp(h[:, 0])
            sh = torch.empty(M, 16, device=DEV)
            _shencoder.sh_encode_forward(dirs.contiguous(), sh, M, 3, 4, None)
            hc = torch.cat([sh, h[:, 1:]], -1)
            hc = torch.relu(torch.relu(hc @ col_net[0].weight.t()) @ col_net[1].weight.t()) @ col_net[2].weight.t()
            rgbs = torch.sigmoid(hc)
            _backend.composite_rays(n_alive, n_step, o["T_thresh"], rays_alive, rays_t, sigmas.contiguous(), rgbs.contiguous(), deltas, weights_sum, depth, image)
            rays_alive = rays_alive[rays_alive >= 0]                        # renderer.py:887
            step += n_step
        image = image + (1 - weights_sum).unsqueeze(-1)
    # the march / composite / compaction are the same kernels -> identical ray bookkeeping; the network here is torch fp32 GEMMs (as in the reference)
    # against the fused kernel: 1e-4 (north-star tolerance)
    assert (image - want["image"][0]).abs().max() < 1e-4 and (weights_sum - want["weights_sum"]).abs().max() < 1e-4
    assert (depth - want["depth_0"][0]).abs().max() < 1e-3 * max(1.0, float(depth.max()))
    assert float((weights_sum > 0.5).float().mean()) > 0.02               # the object is in view
    # utilities of the same module
    grid = torch.rand(128 ** 3, device=DEV)
    bits = torch.zeros(128 ** 3 // 8, dtype=torch.uint8, device=DEV)
    _backend.packbits(grid, 128 ** 3 // 8, 0.5, bits)
    assert np.array_equal(bits.cpu().numpy(), np.packbits((grid.cpu().numpy() > 0.5).reshape(-1, 8), axis=1, bitorder="little").reshape(-1))
    coords = torch.randint(0, 128, (1000, 3), dtype=torch.int32, device=DEV)
    idx, back = torch.empty(1000, dtype=torch.int32, device=DEV), torch.empty(1000, 3, dtype=torch.int32, device=DEV)
    _backend.morton3D(coords, 1000, idx)
    _backend.morton3D_invert(idx, 1000, back)
    assert torch.equal(back, coords)
    sph = torch.empty(N, 2, device=DEV)
    _backend.sph_from_ray(rays_o, rays_d, 8.0, N, sph)   # raymarching.h:8 (background model's texture coordinate)
    want_sph = oracle.sph_from_ray(rays_o.cpu().numpy(), rays_d.cpu().numpy(), 8.0)   # the camera sits inside the sphere
    assert np.abs(sph.cpu().numpy() - want_sph).max() < 2e-6 and float(sph.abs().max()) <= 1.0 + 1e-6   # atan2f of the device library vs libm: a few ulp


def test_backend_modules_reject_wrong_element_types(shim_path):
    """The reference's kernels take data_ptr<float>() / <int>() / <uint8_t>() and throw on anything else; the drop-in must not reinterpret a half or
    int64 tensor (round-2 advisor finding), and the march reports its device error flags (the reference printf's "ERROR: g0=..." for the same case)."""
    import _raymarching as _backend
    N = 64
    o, d = torch.zeros(N, 3, device=DEV), torch.ones(N, 3, device=DEV)
    aabb = torch.tensor([-1.0, -1, -1, 1, 1, 1], device=DEV)
    nears, fars = torch.empty(N, device=DEV), torch.empty(N, device=DEV)
    _backend.near_far_from_aabb(o, d, aabb, N, 0.2, nears, fars)
    with pytest.raises(RuntimeError, match="float32"):
        _backend.near_far_from_aabb(o.half(), d, aabb, N, 0.2, nears, fars)
    with pytest.raises(RuntimeError, match="float32"):
        _backend.near_far_from_aabb(o, d, aabb.double(), N, 0.2, nears, fars)
    alive = torch.arange(N, dtype=torch.int64, device=DEV)
    with pytest.raises(RuntimeError):
        _backend.composite_rays(N, 1, 1e-2, alive, nears, torch.zeros(N, device=DEV), torch.zeros(N, 3, device=DEV), torch.zeros(N, 2, device=DEV),
                                torch.zeros(N, device=DEV), torch.zeros(N, device=DEV), torch.zeros(N, 3, device=DEV))
