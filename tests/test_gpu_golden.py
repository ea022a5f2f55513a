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


def test_widened_ops_against_the_committed_fixture(ckpt):
    """tests/golden/train_kat.npz (make_golden_train.py): the HIP path of the static march, march_rays_train, composite_rays_train fwd/bwd,
    grid dy_dx / backward / TV, SH dy_dx / backward, packbits, morton3D against the committed vectors — index / sample work bit for bit,
    floating-point reductions within 1e-4 relative."""
    import os
    from pienerf_amd import raymarching
    from pienerf_amd._lib import check, lib, ptr, stream_ptr
    from pienerf_amd.gridencoder.grid import grid_encode
    from pienerf_amd.shencoder.sphere_harmonics import sh_encode
    k = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "train_kat.npz"))
    rel = lambda a, b: float(np.abs(np.asarray(a, np.float64) - b).max() / max(1e-30, np.abs(b).max()))
    o, d, nears, fars = (T(k[n]) for n in ("rays_o", "rays_d", "nears", "fars"))
    bits = T(ckpt["density_bitfield"])
    # march_rays_train with the fixture's noise vector (C ABI: the wrapper draws its own)
    N, M = k["rays_o"].shape[0], k["train_xyzs"].shape[0]
    xyzs, dirs, deltas = (torch.zeros(M, c, device=DEV) for c in (3, 3, 2))
    rays, counter, noise = torch.empty(N, 3, dtype=torch.int32, device=DEV), torch.zeros(2, dtype=torch.int32, device=DEV), T(k["noise"])
    check(lib().pn_march_rays_train(ptr(o), ptr(d), ptr(bits), 1.0, 0.0, 256, N, 1, 128, M, ptr(nears), ptr(fars), ptr(xyzs), ptr(dirs), ptr(deltas), ptr(rays),
                                    ptr(counter), ptr(noise), stream_ptr()))
    assert np.array_equal(counter.cpu().numpy(), k["train_counter"]) and np.array_equal(rays.cpu().numpy(), k["train_rays"])
    assert np.array_equal(xyzs.cpu().numpy(), k["train_xyzs"]) and np.array_equal(deltas.cpu().numpy(), k["train_deltas"])
    alive = T(k["static_alive"])
    sx, _, sl = raymarching.march_rays(len(k["static_alive"]), 8, alive, nears, o, d, 1.0, bits, 1, 128, nears, fars, 128, False, 0.0, 256)
    assert np.array_equal(sx.cpu().numpy(), k["static_xyzs"]) and np.array_equal(sl.cpu().numpy(), k["static_deltas"])
    Mv = int(k["train_counter"][0])
    sig, rgb = T(k["sig"]).requires_grad_(True), T(k["rgb"]).requires_grad_(True)
    ws, depth, image = raymarching.composite_rays_train(sig, rgb, deltas[:Mv], rays, 1e-2)
    assert rel(ws.detach().cpu().numpy(), k["comp_ws"]) < 1e-5 and rel(depth.detach().cpu().numpy(), k["comp_depth"]) < 1e-5
    assert rel(image.detach().cpu().numpy(), k["comp_image"]) < 1e-5
    ((ws * T(k["gws"])).sum() + (image * T(k["gim"])).sum()).backward()
    assert rel(sig.grad.cpu().numpy(), k["comp_gs"]) < 1e-4 and rel(rgb.grad.cpu().numpy(), k["comp_gc"]) < 1e-5
    x, emb = T(k["grid_x"]).requires_grad_(True), T(k["grid_emb"]).requires_grad_(True)
    grid_encode(x, emb, T(k["grid_offsets"]), 1.6, 8, True).backward(T(k["grid_grad"]))
    assert rel(emb.grad.cpu().numpy(), k["grid_ge"]) < 1e-4 and rel(x.grad.cpu().numpy(), k["grid_gi"]) < 1e-4
    g = torch.zeros_like(emb)
    off_host = torch.from_numpy(k["grid_offsets"])
    xd, ed = x.detach().contiguous(), emb.detach().contiguous()
    check(lib().pn_grad_total_variation(ptr(xd), ptr(ed), ptr(g), off_host.data_ptr(), 0.3, xd.shape[0], 3, 2, 6, float(np.float32(np.log2(1.6))), 8, 0, 0,
                                        stream_ptr()))
    assert rel(g.cpu().numpy(), k["grid_tv"]) < 1e-4
    sd = T(k["sh_dirs"]).requires_grad_(True)
    sh_encode(sd, 4, True).backward(T(k["sh_grad"]))
    assert rel(sd.grad.cpu().numpy(), k["sh_gi"]) < 1e-5
    assert np.array_equal(raymarching.packbits(T(k["pack_grid"]), 0.5).cpu().numpy(), k["pack_bits"])
    assert np.array_equal(raymarching.morton3D(T(k["mort_coords"])).cpu().numpy(), k["mort_idx"])


def test_get_rays_kernel_equals_reference_output():
    """k_get_rays against rays the REFERENCE's own get_rays (nerf/utils.py:54-138) returned for the same pose / intrinsics
    (tests/golden/make_golden_ref.py ran it in the build container): origins bit for bit, directions within float32 rounding of a unit
    vector (the reference normalises with torch.norm and rotates with a batched matmul)."""
    from pienerf_amd.nerf.utils import get_rays
    k = np.load(os.path.join(G, "ref_kat.npz"))
    for tag in "abc":
        W, H = (int(v) for v in k[f"rays_{tag}_WH"])
        r = get_rays(T(k[f"rays_{tag}_pose"][None]), k[f"rays_{tag}_intr"], H, W)
        o, d = r["rays_o"][0].cpu().numpy(), r["rays_d"][0].cpu().numpy()
        if tag == "c":
            o, d = o[k["rays_c_idx"]], d[k["rays_c_idx"]]
        assert np.array_equal(o, k[f"rays_{tag}_o"])
        assert np.abs(d - k[f"rays_{tag}_d"]).max() < 2.5e-7


def test_trunc_exp_on_device_equals_reference_output():
    from pienerf_amd.nerf.activation import trunc_exp
    k = np.load(os.path.join(G, "ref_kat.npz"))
    x = T(k["trunc_exp_x"]).requires_grad_(True)
    y = trunc_exp(x)
    y.backward(T(k["trunc_exp_g"]))
    fin = np.isfinite(k["trunc_exp_y"])
    assert rel_err(y.detach().cpu().numpy()[fin], k["trunc_exp_y"][fin]) < 1e-6
    assert rel_err(x.grad.cpu().numpy(), k["trunc_exp_dx"]) < 1e-6
