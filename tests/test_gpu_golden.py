"""The HIP path (through the C ABI) against the committed golden fixtures alone — no oracle involved at run time."""
import os

import numpy as np
import pytest
import torch

from conftest import rel_err
from pienerf_amd import scene

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DEV = "cuda"


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def test_ops_kat_gpu(ckpt):
    from pienerf_amd.gridencoder import grid_encode
    from pienerf_amd.nerf.network import NeRFNetwork
    from pienerf_amd.shencoder import sh_encode
    k = np.load(os.path.join(G, "ops_kat.npz"))
    g = grid_encode(T(k["u"]), T(ckpt["embeddings"]), T(ckpt["offsets"]), ckpt["per_level_scale"], ckpt["base_resolution"]).cpu().numpy()
    assert np.abs(g - k["grid"]).max() < 2e-6
    assert np.abs(sh_encode(T(k["d"]), 4).cpu().numpy() - k["sh"]).max() < 1e-6
    net = NeRFNetwork(encoding="hashgrid", bound=1.0, cuda_ray=True).to(DEV).load_checkpoint_dict(ckpt)
    with torch.no_grad():
        s, c = net(T(k["x"]), T(k["d"]))
    assert np.abs(s.cpu().numpy() / k["sigma"] - 1).max() < 1e-4 and np.abs(c.cpu().numpy() - k["rgb"]).max() < 1e-4


def test_render_kat_gpu(ckpt, small_opt):
    from pienerf_amd import raymarching
    from pienerf_amd.nerf.network import NeRFNetwork
    from pienerf_amd.nerf.utils import get_pnts_in_grids, get_rays
    k = np.load(os.path.join(G, "render_kat.npz"))
    W = int(k["W"])
    r = get_rays(T(k["pose"][None]), k["intrinsics"], W, W)
    assert np.array_equal(r["rays_d"][0].cpu().numpy(), k["rays_d"])
    n_grid = int(k["resolution"].prod())
    pig = get_pnts_in_grids(len(k["p_def"]), n_grid, T(k["p_def"]), T(k["bbmin"]), T(k["bbmax"]), float(k["hgs"]), T(k["resolution"]))
    for a, b in zip(pig, (k["pig_cnt"], k["pig_bgn"], k["pig_idx"])):
        assert np.array_equal(a.cpu().numpy(), b)
    nears, fars = raymarching.near_far_from_aabb(r["rays_o"][0], r["rays_d"][0], T(np.concatenate([k["bbmin"], k["bbmax"]])), 0.2)
    assert np.array_equal(nears.cpu().numpy(), k["nears"]) and np.array_equal(fars.cpu().numpy(), k["fars"])
    xyzs, dirs, deltas = raymarching.march_rays_quadratic_bending(
        *pig, len(k["p_def"]), n_grid, T(k["p_def"]), T(k["p_ori"]), T(k["F"]), T(k["dF"]), 1, T(k["bbmin"]), T(k["bbmax"]), float(k["hgs"]),
        T(k["resolution"]), 3, float(k["IP_dx"]), False, torch.zeros(6, device=DEV), W * W, 4, torch.arange(W * W, dtype=torch.int32, device=DEV), nears,
        r["rays_o"][0], r["rays_d"][0], 1.0, T(ckpt["density_bitfield"]), 1, 128, nears, fars, 128)
    assert np.array_equal(xyzs.cpu().numpy().view(np.uint32), k["xyzs"].view(np.uint32))
    assert np.array_equal(deltas.cpu().numpy().view(np.uint32), k["deltas"].view(np.uint32))
    net = NeRFNetwork(encoding="hashgrid", bound=1.0, cuda_ray=True).to(DEV).load_checkpoint_dict(ckpt)
    net.p_def, net.p_ori, net.IP_F, net.IP_dF, net.IP_dx = T(k["p_def"]), T(k["p_ori"]), T(k["F"]), T(k["dF"]), float(k["IP_dx"])
    with torch.no_grad():
        out = net.render_deformed(r["rays_o"], r["rays_d"], collect_stats=True, **small_opt)
    assert net.last_stats["trips"] == int(k["trips"]) and net.last_stats["samples"] == int(k["samples"])
    assert np.abs(out["image"][0].cpu().numpy() - k["image"]).max() < 1e-4
    assert np.abs(out["weights_sum"].cpu().numpy() - k["weights_sum"]).max() < 1e-4


def test_sim_kat_gpu(small_cloud, small_opt):
    from pienerf_amd.simulator.solver import Simulator
    k = np.load(os.path.join(G, "sim_kat.npz"))
    o = small_opt
    sim = Simulator(dt=o["sim_dt"], iters=o["sim_iters"], bbox=torch.tensor([2.0 * o["bound"]] * 3), dx=o["sim_dx"], stiff=o["sim_stiff"],
                    base=torch.tensor([-o["bound"]] * 3), device=DEV)
    c = small_cloud
    sim.InitializeFromArrays(c["pos"], c["mass"], c["mu"], c["lam"], c["pin"])
    assert (sim.n_IP, sim.n_k, len(sim.active_kernels)) == (int(k["n_IP"]), int(k["n_k"]), int(k["n_active"]))
    assert rel_err(sim.rhs_rest.cpu().numpy().reshape(-1, 3), k["rhs_rest"]) < 1e-9
    sim.update_force(int(k["force_vid"]), k["force"])
    rest = sim.dof_rest.cpu().numpy().reshape(-1, 3)
    for i in range(3):
        sim.stepforward()
        assert rel_err(sim.dof.cpu().numpy().reshape(-1, 3) - rest, k["dof_steps"][i] - rest) < 1e-4, i
    for _ in range(9):
        sim.stepforward()
    p, F, dF = (t.cpu().numpy() for t in sim.get_IP_info())
    assert np.abs(p - k["p_def_12"]).max() < 1e-5 and np.abs(F - k["F_12"]).max() < 1e-4 and rel_err(dF, k["dF_12"]) < 1e-3
    assert rel_err(sim.dof_vel.cpu().numpy().reshape(-1, 3), k["dof_vel_12"]) < 1e-3
