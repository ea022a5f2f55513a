"""The substep's local/global iterations as ONE persistent kernel (pn_sim_stepforward_coop, csrc/pn_sim.hip: k_substep_coop) against the launch form
(pn_sim_stepforward) and the fp64 CPU oracle: same trajectory, bit-reproducible, inside captured graphs, and the fallback when a scene does not fit."""
import numpy as np
import pytest
import torch

from conftest import make_oracle_sim, rel_err
from pienerf_amd import scene
from test_gpu_parity import DEV

pytestmark = pytest.mark.gpu


def _sim(cloud, opt, persistent, iters=None):
    from pienerf_amd.simulator.solver import Simulator
    s = Simulator(dt=opt["sim_dt"], iters=opt["sim_iters"] if iters is None else iters, bbox=torch.tensor([2.0 * opt["bound"]] * 3), dx=opt["sim_dx"],
                  stiff=opt["sim_stiff"], base=torch.tensor([-opt["bound"]] * 3), device=DEV, persistent=persistent)
    s.InitializeFromArrays(cloud["pos"], cloud["mass"], cloud["mu"], cloud["lam"], cloud["pin"])
    return s


def _run(s, steps, force_at=None, f=(300.0, 100.0, -200.0)):
    traj = []
    for k in range(steps):
        if force_at is not None and k == force_at:
            s.update_force(s.n_IP // 2, np.array(f))
        s.stepforward()
        traj.append(s.dof.clone())
    torch.cuda.synchronize()
    return traj


@pytest.mark.parametrize("iters", [1, 4, 10])
def test_persistent_substep_equals_the_launch_form_small_scene(small_cloud, small_opt, iters):
    a, b = _sim(small_cloud, small_opt, False, iters), _sim(small_cloud, small_opt, True, iters)
    ta, tb = _run(a, 8, force_at=2), _run(b, 8, force_at=2)
    assert b.persistent and b._coop is not None and not b.persistent_timed_out()
    rest = a.dof_rest
    worst = 0.0
    for k, (x, y) in enumerate(zip(ta, tb)):
        # different summation orders (pieces of <= 272 entries instead of per-cell chunks, a different lane -> column map in the matrix rows) and, since
        # round 5, different stopping rules of the warm-started Jacobi SVD (off-diagonals below 1e-11 of the diagonal in the cell form, 1e-12 in the
        # persistent one): rotations agree to ~1e-11 absolute, the first frames' displacements are ~1e-3
        e = rel_err((y - rest).cpu().numpy(), (x - rest).cpu().numpy())
        worst = max(worst, e)
        assert e < 1e-8, (k, e)
    print(f"iters {iters}: persistent vs cell form, worst relative difference of the displacements {worst:.2e}")
    assert rel_err(b.dof_vel.cpu().numpy(), a.dof_vel.cpu().numpy()) < 1e-6
    assert float((ta[-1] - rest).abs().max()) > 1e-3  # the force moved it


def test_persistent_substep_full_size_against_the_oracle_and_reproducible():
    """139 kernels / 3 576 IPs on every CU: three substeps against the fp64 oracle (the launch form's own bar, 1e-6), the same bits from a second
    simulator, the same trajectory as the launch (cell) form to 1e-8."""
    opt = scene.default_opt()
    cloud = scene.make_chair_points(hgs=opt["hash_grid_size"])
    ref = make_oracle_sim(cloud, opt)
    a, b, c = _sim(cloud, opt, True), _sim(cloud, opt, True), _sim(cloud, opt, False)
    assert (a.n_k, a.n_IP) == (ref.n_k, ref.n_IP) == (139, 3576)
    f = np.array([300.0, 100.0, -200.0])
    for s in (a, b, c, ref):
        s.update_force(s.n_IP // 2, f)
    for step in range(3):
        for s in (a, b, c, ref):
            s.stepforward()
            torch.cuda.synchronize()
        da = a.dof.cpu().numpy().reshape(-1, 3) - ref.dof_rest
        assert rel_err(da, ref.dof - ref.dof_rest) < 1e-6, step
        assert torch.equal(a.dof, b.dof) and torch.equal(a.dof_vel, b.dof_vel), step
        e = rel_err(da, c.dof.cpu().numpy().reshape(-1, 3) - ref.dof_rest)   # the cell form: other summation orders, SVD stopped at 1e-11 instead of 1e-12
        print(f"step {step}: persistent vs cell form {e:.2e}")
        assert e < 1e-8, (step, e)
    assert a._coop is not None and not a.persistent_timed_out() and not b.persistent_timed_out()


def test_persistent_substep_inside_a_captured_graph(small_cloud, small_opt):
    """Replays of a captured persistent substep == eager persistent substeps, bit for bit.  (One persistent kernel at a time per device: two of
    them on different streams can each hold part of the CUs and wait for the rest — the kernel's barrier guard then ends both with the timed-out
    flag — so the two simulators take turns here.)"""
    a, b = _sim(small_cloud, small_opt, True), _sim(small_cloud, small_opt, True)
    for s in (a, b):
        s.update_force(s.n_IP // 3, np.array([0.0, 250.0, 50.0]))
        s.stepforward()
        torch.cuda.synchronize()
    st = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.graph(g, stream=st):
        b.stepforward()
    with torch.cuda.stream(st):
        for _ in range(4):  # a capture only records: 1 eager step + 4 replays
            g.replay()
    torch.cuda.synchronize()
    for _ in range(4):
        a.stepforward()
    torch.cuda.synchronize()
    assert not a.persistent_timed_out() and not b.persistent_timed_out()
    assert torch.equal(b.dof, a.dof) and torch.equal(b.dof_vel, a.dof_vel)


def test_scene_that_does_not_fit_keeps_the_launch_form():
    """343 kernels: 3430 unknowns per component > 2048 register-resident columns: pn_sim_coop_bytes says 0 and the simulator steps as before."""
    from pienerf_amd._lib import lib
    assert int(lib().pn_sim_coop_bytes(343, 20000, 256)) == 0
    assert int(lib().pn_sim_coop_bytes(139, 3576, 256)) > 0
    assert int(lib().pn_sim_coop_bytes(139, 3576, 4)) == 0       # 894 integration points per workgroup


@pytest.mark.parametrize("iters", [1, 3, 10])
def test_cell_form_of_the_substep_equals_the_csr_form_and_is_reproducible(small_cloud, small_opt, iters):
    """pn_sim_stepforward_cells (round 5: calc_elastic + collect_rhs_IP as one launch per kernel-grid cell chunk, 1 + 2 iters launches) against
    pn_sim_stepforward (rounds 1-4: per-kernel CSR lists, 1 + 3 iters launches): the same trajectory under a changing force to the summation order and
    the SVD's stopping rule, the same bits from a second simulator, and inside a captured graph."""
    a, a2, b = _sim(small_cloud, small_opt, False, iters), _sim(small_cloud, small_opt, False, iters), _sim(small_cloud, small_opt, False, iters)
    assert a.cell_form and a._cells_work is not None
    b.cell_form = False
    ta, ta2, tb = _run(a, 8, force_at=2), _run(a2, 8, force_at=2), _run(b, 8, force_at=2)
    rest = a.dof_rest
    for k, (x, x2, y) in enumerate(zip(ta, ta2, tb)):
        assert torch.equal(x, x2), k
        assert rel_err((x - rest).cpu().numpy(), (y - rest).cpu().numpy()) < 1e-8, k
    assert float((ta[-1] - rest).abs().max()) > 1e-3
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        a.stepforward()
    with torch.cuda.stream(s):
        for _ in range(3):
            g.replay()
    s.synchronize()
    for _ in range(3):
        a2.stepforward()
    torch.cuda.synchronize()
    assert torch.equal(a.dof, a2.dof) and torch.equal(a.dof_vel, a2.dof_vel)
