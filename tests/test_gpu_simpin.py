"""What can be said about the simulator half WITHOUT the reference's Warp kernels (warp-lang is absent: SURVEY.md §8c, "parity unpinned").

The one convention of `wp.svd3` (cuda_utils.py:107) that the restatement has to ASSUME is what happens at an inverted element: with det F < 0
a decomposition may return proper rotations U, V and a negative smallest singular value (the contract this repository and its oracle state), or
reflections and positive singular values — and `R = U V^T`, `volume_invariant_project(sigma)` differ between the two.  For det F > 0 there is
nothing to assume: every SVD with sigma >= 0 gives the same polar rotation R and the same `U diag(sigma') V^T` (the projection is symmetric
in the singular values).  So the tests here run the trajectories of BASELINE.json's three single-GPU configurations — with the forces bench.py
applies — and check that det F stays positive at every integration point of every substep: on these workloads the assumed branch is dormant
and the simulator's results do not depend on it.  (A scene that does invert elements is covered by
test_gpu_edges.py::test_calc_elastic_on_adversarial_deformation_gradients against the oracle's contract only.)

Also here: the arrival counters of the chunked right-hand-side gather are cyclic (round-3 advisor finding)."""
import numpy as np
import pytest
import torch

from pienerf_amd import scene
from test_gpu_parity import DEV

pytestmark = pytest.mark.gpu


def _sim(cloud, opt):
    from pienerf_amd.simulator.solver import Simulator
    s = Simulator(dt=opt["sim_dt"], iters=opt["sim_iters"], bbox=torch.tensor([2.0 * opt["bound"]] * 3), dx=opt["sim_dx"], stiff=opt["sim_stiff"],
                  base=torch.tensor([-opt["bound"]] * 3), device=DEV)
    s.InitializeFromArrays(cloud["pos"], cloud["mass"], cloud["mu"], cloud["lam"], cloud["pin"])
    return s


def _det3(F9):
    """det of [n, 9] matrices (either index order: det F = det F^T)."""
    F = F9.double().view(-1, 3, 3)
    return torch.linalg.det(F)


CONFIGS = {
    # name: (opt, cloud kwargs, force) — the same scenes and forces as bench.py's make_config
    "chair": (lambda: scene.default_opt(), dict(), None),
    "chair_forced": (lambda: scene.default_opt(), dict(), (300.0, 100.0, -200.0)),
    "trex": (lambda: scene.trex_opt(radius=4.5), dict(bound=2.0), (250.0, 120.0, -180.0)),
    "stress": (lambda: scene.stress_opt(), dict(sub_res=180), (400.0, -150.0, 250.0)),
}


@pytest.mark.parametrize("name", list(CONFIGS))
def test_det_F_stays_positive_on_the_baseline_trajectories(name):
    """min over integration points and substeps of det F on the configs[1] / [2] / [4] trajectories: > 0, i.e. the svd3 sign convention is never
    exercised there.  300 substeps = the length of the bench's timed region (prime + warm-up + 200 steps)."""
    mk, ckw, force = CONFIGS[name]
    opt = mk()
    cloud = scene.make_chair_points(hgs=opt["hash_grid_size"], **dict({"bound": opt["bound"]}, **ckw))
    s = _sim(cloud, opt)
    if force is not None:
        s.update_force(s.n_IP // 2, np.array(force))
    lo, hi = float("inf"), 0.0
    for step in range(300):
        s.stepforward()
        if step % 4 == 0 or step < 20:
            _, F, _ = s.get_IP_info()
            d = _det3(F)
            lo, hi = min(lo, float(d.min())), max(hi, float(d.max()))
    torch.cuda.synchronize()
    print(f"{name}: n_IP {s.n_IP}, det F in [{lo:.4f}, {hi:.4f}] over 300 substeps")
    assert lo > 0.2, (name, lo)          # far from inversion
    assert np.isfinite(hi) and hi < 5.0
    assert abs(hi - 1.0) > 1e-4 or abs(lo - 1.0) > 1e-4  # the body did deform


def test_gather_arrival_counters_are_cyclic():
    """k_rhs_gather_chunk's per-kernel arrival counters return to 0 at the end of every launch (the last arriver stores 0), so they cannot wrap
    however long a simulator lives; rounds 1-3 let them grow by `chunks` per local/global iteration and tested (n % chunks) == 0."""
    from pienerf_amd._lib import lib
    opt = scene.default_opt()   # 139 kernels, CSR lists of up to 770 entries: most kernels have several chunks of 64
    s = _sim(scene.make_chair_points(hgs=opt["hash_grid_size"]), opt)
    for _ in range(5):
        s.stepforward()
    torch.cuda.synchronize()
    n_k, n_IP = s.n_k, s.n_IP
    chunks_max = n_IP * 8 // 64 + n_k                       # pn_gather_chunks_max (PN_GCH = 64)
    part0 = 4 * n_k * 30 + n_IP * 9 + n_IP * 8 * 9           # doubles in front of the chunk sums
    kc_bg0 = (part0 + chunks_max * 30) * 2                  # int index of kc_bg in the work buffer
    slot = (n_k + 2) & ~1
    assert int(lib().pn_sim_work_doubles(n_k, n_IP)) == s._work.numel()
    ints = s._work.view(torch.int32)
    kc_bg = ints[kc_bg0:kc_bg0 + n_k + 1].cpu().numpy()
    kcount = ints[kc_bg0 + slot:kc_bg0 + slot + n_k].cpu().numpy()
    assert kc_bg[0] == 0 and (np.diff(kc_bg) >= 0).all() and kc_bg[-1] > 0   # we are looking at the plan
    assert (np.diff(kc_bg) > 1).any()                                        # kernels with several chunks exist: the counters are used
    assert not kcount.any()
