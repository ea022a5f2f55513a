"""SURVEY §8(f) rank 2 — point sampling from the density field (pienerf_amd/sampling.py, mirror of main_sample.py) against the CPU
restatement oracle/sampling.py, README.md:91 parameters (--sub_coeff 0.55) at test resolution."""
import numpy as np
import pytest
import torch

from oracle import sampling as osamp
from pienerf_amd import scene
from test_gpu_parity import DEV, T

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def shaped():
    from pienerf_amd.nerf.network import NeRFNetwork
    ck = scene.make_checkpoint(bound=1.0, seed=0, shaped=True)
    net = NeRFNetwork(encoding="hashgrid", bound=1.0, cuda_ray=True).to(DEV).load_checkpoint_dict(ck)
    opt = scene.default_opt(sub_res=24, sub_coeff=0.55, density_threshold=0.05, sim_dx=0.1)
    rand = np.random.default_rng(3).random((512, 3)).astype(np.float32)
    return dict(ck=ck, net=net, opt=opt, rand=rand)


def test_density_matches_oracle_and_forward(shaped):
    """pn_nerf_density (the fused kernel stopped after the sigma net) vs the CPU restatement, and vs the full forward's sigma."""
    net, ck = shaped["net"], shaped["ck"]
    rng = np.random.default_rng(8)
    x = (rng.random((5003, 3)).astype(np.float32) * 2 - 1) * 0.98
    s_ref, g_ref = osamp.nerf_density(x, ck, 1.0)
    with torch.no_grad():
        out = net.density(T(x))
        s_full, _ = net(T(x), T(np.tile(np.array([[0, 0, 1]], np.float32), (len(x), 1))))
    s, g = out["sigma"].cpu().numpy(), out["geo_feat"].cpu().numpy()
    assert g.shape == (5003, 15)
    assert np.abs(s / s_ref - 1).max() < 1e-4 and np.abs(g - g_ref).max() < 1e-4
    assert np.array_equal(s, s_full.cpu().numpy())                      # same kernel, same arithmetic up to that point
    inside = scene.chair_solid(x, margin=-0.04)
    assert np.median(s[inside]) > 30 and np.median(s[~scene.chair_solid(x, margin=0.06)]) < 0.05   # the field has the chair's shape


def test_sample_equals_the_restatement_given_the_same_density(shaped):
    """Algorithm parity, bit for bit: the oracle's kernels-by-kernel loops are fed the GPU network's density values."""
    from pienerf_amd.sampling import AdaptiveUniformSampling
    net, opt, rand = shaped["net"], shaped["opt"], shaped["rand"]
    s = AdaptiveUniformSampling(opt, net, device=DEV)
    pts, vols = s.sample(rand=torch.from_numpy(rand))

    def gpu_density(p):
        with torch.no_grad():
            return net.density(T(np.ascontiguousarray(p, np.float32)))["sigma"].cpu().numpy()
    rp, rv, info = osamp.sample(opt, gpu_density, rand)
    assert info["boundary_points"] > 100 and info["boundary_points"] == s.last["boundary_points"] and info["kept"] == s.last["kept"]
    assert np.array_equal(pts.cpu().numpy(), rp)
    assert np.array_equal(vols.cpu().numpy(), rv)


def test_sample_end_to_end_and_feeds_the_simulator(shaped, tmp_path):
    """Mirror with its own density vs oracle with the CPU density (threshold decisions may differ on a handful of borderline points);
    the cloud has the chair's shape and volume, round-trips through the PLY the reference writes and initialises the simulator."""
    from pienerf_amd.sampling import AdaptiveUniformSampling, simulator_cloud, write_ply
    from pienerf_amd.simulator.solver import Simulator
    net, ck, opt, rand = shaped["net"], shaped["ck"], shaped["opt"], shaped["rand"]
    s = AdaptiveUniformSampling(opt, net, device=DEV)
    pts, vols = s.sample(rand=torch.from_numpy(rand))
    rp, rv, info = osamp.sample(opt, lambda p: osamp.nerf_density(p, ck, 1.0)[0], rand)
    assert abs(len(rp) - pts.shape[0]) <= 3 and abs(float(vols.sum()) - float(rv.sum())) < 2e-3
    p = pts.cpu().numpy()
    assert scene.chair_solid(p, margin=0.07).all()
    solid_volume = 0.3913                                                # union of scene.CHAIR_BOXES
    assert 0.7 * solid_volume < float(vols.sum()) < 1.3 * solid_volume
    hgs = opt["hash_grid_size"]
    cells = len(np.unique(np.floor((p - (p.min(0) - np.float32(1e-3))) / np.float32(hgs)).astype(np.int64), axis=0))
    assert abs(float(vols.sum()) - cells * hgs ** 3) < 1e-4 * cells * hgs ** 3      # volumes partition the occupied hash cells
    path = tmp_path / "model" / "chair_0.ply"
    write_ply(str(path), pts, vols)
    back = scene.read_ply(str(path))
    assert np.array_equal(back["x"], p[:, 0].astype(np.float64)) and np.array_equal(back["vp"], vols.cpu().numpy().astype(np.float64))
    c = simulator_cloud(pts, vols)
    sim = Simulator(dt=opt["sim_dt"], iters=2, bbox=torch.tensor([2.0] * 3), dx=opt["sim_dx"], stiff=opt["sim_stiff"], base=torch.tensor([-1.0] * 3), device=DEV)
    sim.InitializeFromArrays(c["pos"], c["mass"], c["mu"], c["lam"], c["pin"])
    assert sim.n_IP > 50 and sim.n_k > 10
    for _ in range(3):
        sim.stepforward()
    assert bool(torch.isfinite(sim.dof).all())
