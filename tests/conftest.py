import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu); everything else runs on CPU")


def _torch():
    import torch
    return torch


# ---------------------------------------------------------------- shared small scene (CPU-cheap)
SMALL = dict(sub_res=30, sim_dx=0.1, sim_iters=4)


@pytest.fixture(scope="session")
def small_opt():
    from pienerf_amd import scene
    return scene.default_opt(sim_dx=SMALL["sim_dx"], sim_iters=SMALL["sim_iters"], W=48, H=48)


@pytest.fixture(scope="session")
def small_cloud(small_opt):
    from pienerf_amd import scene
    return scene.make_chair_points(sub_res=SMALL["sub_res"], hgs=small_opt["hash_grid_size"])


@pytest.fixture(scope="session")
def ckpt():
    from pienerf_amd import scene
    return scene.make_checkpoint(bound=1.0, seed=0)


@pytest.fixture(scope="session")
def oracle_sim(small_cloud, small_opt):
    """Oracle simulator on the small scene, at rest."""
    torch = _torch()
    from oracle.sim_init import OracleSimulator
    o = small_opt
    s = OracleSimulator(dt=o["sim_dt"], iters=o["sim_iters"], bbox=torch.tensor([2.0 * o["bound"]] * 3), dx=o["sim_dx"], stiff=o["sim_stiff"],
                        base=torch.tensor([-o["bound"]] * 3))
    c = small_cloud
    s.InitializeFromArrays(c["pos"], c["mass"], c["mu"], c["lam"], c["pin"])
    return s


def make_oracle_sim(cloud, opt):
    torch = _torch()
    from oracle.sim_init import OracleSimulator
    s = OracleSimulator(dt=opt["sim_dt"], iters=opt["sim_iters"], bbox=torch.tensor([2.0 * opt["bound"]] * 3), dx=opt["sim_dx"],
                        stiff=opt["sim_stiff"], base=torch.tensor([-opt["bound"]] * 3))
    s.InitializeFromArrays(cloud["pos"], cloud["mass"], cloud["mu"], cloud["lam"], cloud["pin"])
    return s


@pytest.fixture(scope="session")
def deformed_ip_state(small_cloud, small_opt):
    """IP state (p_def, p_ori, F, dF) after a few oracle steps: a genuinely deformed configuration for the render tests."""
    s = make_oracle_sim(small_cloud, small_opt)
    p_ori, _, _ = s.get_IP_info()
    s.update_force(s.n_IP // 2, np.array([300.0, 100.0, -200.0]))
    for _ in range(12):
        s.stepforward()
    p_def, F, dF = s.get_IP_info()
    return dict(p_def=p_def, p_ori=p_ori, F=F, dF=dF, IP_dx=s.dx * 1.05)


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))
