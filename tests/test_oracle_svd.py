"""The two restatements of `wp.svd3` (simulator/cuda_utils.py:107) in oracle/sim_oracle.cpp, one against the other (CPU only).

`svd3_converged` restates the CONTRACT (proper rotations U, V; the sign of det F on the last singular value) with a Jacobi run to fp64
convergence; `svd3_mcadams` restates the published ALGORITHM (McAdams et al., UW-Madison TR1690: a fixed number of Jacobi sweeps with the
approximate Givens quaternion, negating-swap sort, Givens-quaternion QR).  The tests here show
  * the algorithm is a valid svd3 by the contract,
  * on the adversarial deformation-gradient set and on the BASELINE trajectories how far R = U V^T, U diag(sigma') V^T and the DOF
    displacements after 10 substeps are from the converged decomposition — by sweep count (4, 8), by reciprocal-sqrt flavour and with the
    paper's 10-digit constants against their fp64 values (which is where the 8-sweep gap comes from),
  * BASELINE configs[0] as it is stated (chair_0.ply stand-in, 1 local/global iteration, dx 0.05, simulator only, CPU).
The numbers are printed (run with -s) and recorded in EXPERIMENTS.md (round 6)."""
import numpy as np
import pytest

import oracle
from conftest import make_oracle_sim
from svd_cases import adversarial_F, elastic_inputs_for, trajectory_scene, well_conditioned

MODES = {
    "mcadams 8 sweeps": dict(mode="mcadams", sweeps=8),
    "mcadams 8 sweeps, seeded rsqrt": dict(mode="mcadams", sweeps=8, rsqrt="seeded"),
    "mcadams 8 sweeps, fp64 constants": dict(mode="mcadams", sweeps=8, constants="exact"),
    "mcadams 4 sweeps": dict(mode="mcadams", sweeps=4),
    "mcadams 4 sweeps, fp64 constants": dict(mode="mcadams", sweeps=4, constants="exact"),
}


def _svd_batch(Fs, **mode):
    out = []
    with oracle.svd_mode(**mode):
        for F in Fs:
            out.append(oracle.svd3(F))
    return out


def test_mcadams_satisfies_the_svd3_contract():
    """U, V proper rotations, U diag(sigma) V^T = F, sigma_0 >= sigma_1 >= |sigma_2|, sign(sigma_2) = sign(det F) — to the accuracy the 10-digit
    constants leave (1e-7), on random, inverted and scaled deformation gradients; the polar rotation equals numpy's."""
    rng = np.random.default_rng(1)
    Fs = [np.eye(3) + s * rng.standard_normal((3, 3)) for s in (1e-6, 1e-3, 0.1, 0.5) for _ in range(100)]
    Fs += [rng.standard_normal((3, 3)) for _ in range(200)]
    Fs += [F @ np.diag([1, 1, -1.0]) for F in Fs[300:400]]
    Fs = np.stack(Fs)
    ok = well_conditioned(Fs)
    assert ok.sum() > 500
    for name in ("mcadams 8 sweeps", "mcadams 8 sweeps, seeded rsqrt"):
        for F, (U, s, V) in zip(Fs[ok], _svd_batch(Fs[ok], **MODES[name])):
            sc = np.abs(F).max()
            assert np.abs(U @ U.T - np.eye(3)).max() < 1e-7 and abs(np.linalg.det(U) - 1) < 1e-7, name
            assert np.abs(V @ V.T - np.eye(3)).max() < 1e-7 and abs(np.linalg.det(V) - 1) < 1e-7, name
            assert np.abs(U @ np.diag(s) @ V.T - F).max() < 2e-7 * sc, name
            assert s[0] >= s[1] - 1e-7 * sc and s[1] >= abs(s[2]) - 1e-7 * sc and s[1] > 0
            assert np.sign(s[2]) == np.sign(np.linalg.det(F))
            Un, sn, Vtn = np.linalg.svd(F)
            if np.linalg.det(F) > 0:
                assert np.abs(U @ V.T - Un @ Vtn).max() < 2e-7, name


def test_mcadams_gap_on_the_adversarial_set():
    """max |R - R_converged| and |U diag(sigma') V^T - (...)_converged| through calc_elastic (cuda_utils.py:83-121) on the adversarial set, per
    variant.  Asserted where the answer is determined (well-conditioned F): 8 sweeps <= 1e-6, with fp64 constants <= 1e-9; 4 sweeps are NOT
    converged in fp64 (the paper's 4 are its single-precision setting) — reported, bounded loosely."""
    Fs = adversarial_F()
    topo, dNx, dof = elastic_inputs_for(Fs)
    tiny = np.abs(Fs).max(axis=(1, 2)) < 1e-9       # the 1e-12-scaled F: the algorithm's QR epsilon is ABSOLUTE (1e-12), see below
    ok = well_conditioned(Fs) & ~tiny
    assert ok.sum() > 300 and tiny.sum() == 2       # (the zero matrix and the 1e-12-scaled one)
    R0, V0, F0 = oracle.calc_elastic(topo, dNx, dof)
    scale = np.maximum(1.0, np.abs(Fs).max(axis=(1, 2)))[:, None, None]
    rows = {}
    for name, mode in MODES.items():
        with oracle.svd_mode(**mode):
            R, V, FF = oracle.calc_elastic(topo, dNx, dof)
        fin = np.isfinite(V).all(axis=(1, 2)) & np.isfinite(V0).all(axis=(1, 2))
        rows[name] = (np.abs(R - R0)[ok].max(), np.abs((V - V0) / scale)[ok & fin].max(), np.abs((FF - F0) / scale)[ok].max())
        print(f"adversarial set, {name:36s}: max|R - R_conv| {rows[name][0]:.2e}  max|V - V_conv| {rows[name][1]:.2e}  max|U S V^T - F| {rows[name][2]:.2e}")
        # the det-+1 contract holds for EVERY input, determined or not
        assert np.abs(np.linalg.det(R[np.isfinite(R).all(axis=(1, 2))]) - 1).max() < 1e-6, name
    # a property of the algorithm, not of the contract: entries at the scale of its absolute QR epsilon are not rotated out, so a 1e-12-scaled
    # F (nothing a deformation gradient ever is) gets a different "rotation" than from the scale-free converged decomposition
    i = int(np.flatnonzero(tiny & (np.abs(Fs).max(axis=(1, 2)) > 0))[0])
    with oracle.svd_mode(**MODES["mcadams 8 sweeps"]):
        Rt = oracle.calc_elastic(topo[i:i + 1], dNx[i:i + 1], dof)[0]
    print(f"adversarial set, the 1e-12-scaled F (absolute epsilon 1e-12 in the QR): |R - R_conv| {np.abs(Rt[0] - R0[i]).max():.2e}")
    assert np.isfinite(Rt).all() and abs(np.linalg.det(Rt[0]) - 1) < 1e-6
    assert rows["mcadams 8 sweeps"][0] < 1e-6 and rows["mcadams 8 sweeps"][1] < 1e-6
    assert rows["mcadams 8 sweeps, seeded rsqrt"][0] < 1e-6
    assert rows["mcadams 8 sweeps, fp64 constants"][0] < 1e-9 and rows["mcadams 8 sweeps, fp64 constants"][1] < 1e-9
    assert rows["mcadams 4 sweeps"][0] > 10 * rows["mcadams 8 sweeps"][0]      # 4 sweeps are visibly short of convergence on this set
    assert rows["mcadams 4 sweeps"][0] < 0.1


@pytest.fixture(scope="module")
def chair_state():
    opt, cloud, _ = trajectory_scene("chair")
    s = make_oracle_sim(cloud, opt)
    assert (s.n_k, s.n_IP, s.iters) == (139, 3576, 10)
    return s


def _copy_state(s):
    return {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in s.state().items()}


def _rest_rhs(st):
    """rhs_rest = build_rhs() + (M / dt^2) dof at rest (solver.py:314) under the svd in effect: the reference computes it with the same wp.svd3 as the
    steps, so a trajectory "on the algorithm" needs it from the algorithm too (at F = I every pair is degenerate: 24 fallback rotations)."""
    RF, VF, _ = oracle.calc_elastic(st["IP_kernel"], st["IP_dNx"], st["dof_rest"])
    n10 = st["dof_rest"].shape[0]
    return oracle.collect_rhs_IP(st["dx"], st["IP_kernel"], st["IP_mu"], st["IP_lam"], st["IP_dNx"], RF, VF, n10) + oracle.matvec3(st["Mmat"], st["dof_rest"])


def _run(st0, steps, mode):
    st = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in st0.items()}
    traj = []
    with oracle.svd_mode(**(mode or dict(mode="converged"))):
        st["rhs_rest"] = _rest_rhs(st)
        for _ in range(steps):
            oracle.stepforward(st)
            traj.append(st["dof"].copy())
    return np.array(traj)


@pytest.mark.parametrize("name", ["chair", "chair_forced", "trex"])
def test_mcadams_gap_on_the_baseline_trajectories(name, chair_state):
    """DOF-displacement gap after 10 substeps (each 10 local/global iterations) of the configs[1] / [2] simulations between the two
    restatements (rhs_rest recomputed under each: solver.py:314 is a build_rhs() too): <= 1e-5 of the largest displacement with 8 sweeps (measured
    3e-6 chair, 4e-8 trex), <= 1e-9 with fp64 constants.  (configs[4]'s 268 k-point cloud takes two
    minutes to initialise on the CPU: its trajectory is in tests/test_gpu_simpin.py, from the GPU-initialised state.)"""
    if name == "chair":
        s = chair_state
        st0 = _copy_state(s)
    else:
        opt, cloud, force = trajectory_scene(name)
        s = chair_state if name == "chair_forced" else make_oracle_sim(cloud, opt)
        st0 = _copy_state(s)
        if force is not None:
            keep = s.dof_f.copy()
            s.update_force(s.n_IP // 2, np.array(force))
            st0["dof_f"] = s.dof_f.copy()
            s.dof_f = keep
    base = _run(st0, 10, None)
    disp = np.abs(base[-1] - st0["dof_rest"]).max()
    assert disp > 1e-2
    gaps = {}
    for mname, mode in MODES.items():
        got = _run(st0, 10, mode)
        gaps[mname] = np.abs(got[-1] - base[-1]).max() / disp
        print(f"{name}: 10 substeps, max |displacement| {disp:.3e}; {mname:36s}: gap {gaps[mname]:.2e} of it")
    # 8 sweeps: converged as an iteration (fp64 constants: 1e-11), but the paper's 10-digit cos / sin(pi / 8) leave the quaternion 5e-10 short of unit
    # length per fallback rotation, and at F = I (rest: rhs_rest, solver.py:314) all 24 conjugations are fallbacks — R_rest is off I by ~1e-8, which
    # the stiffness (mu = 1e6) turns into ~3e-6 of the displacements.  Far inside the 1e-4 bar, but not "identical": hence PN_SIM_SVD=mcadams.
    assert gaps["mcadams 8 sweeps"] < 1e-5
    assert gaps["mcadams 8 sweeps, seeded rsqrt"] < 1e-5
    assert gaps["mcadams 8 sweeps, fp64 constants"] < 1e-9
    assert gaps["mcadams 4 sweeps"] < 1e-2      # reported: ~2.6e-4 on the chair — above the 1e-4 bar, which is why the sweep count matters


def test_configs0_as_baseline_states_it():
    """BASELINE configs[0]: the chair cloud, ONE local/global iteration (`1 Newton iter`), sim_dx = 0.05, simulator step only, CPU — SURVEY §8d
    "Config 1 (CPU plumbing)": initialize() + 1 stepforward() + get_IP_info(); shapes, partition of unity, finite DOFs, and the step moves
    the body the way gravity points."""
    from pienerf_amd import scene
    opt = scene.default_opt(sim_iters=1)
    assert opt["sim_dx"] == 0.05
    cloud = scene.make_chair_points(hgs=opt["hash_grid_size"])
    s = make_oracle_sim(cloud, opt)
    assert s.iters == 1 and (s.n_k, s.n_IP) == (139, 3576)
    assert np.abs(s.IP_Nx[:, :, 0].sum(1) - 1).max() < 1e-9                 # partition of unity on the translation slot
    p0, F0, dF0 = s.get_IP_info()
    assert np.abs(F0.reshape(-1, 3, 3) - np.eye(3, dtype=np.float32)).max() < 1e-6 and np.abs(dF0).max() < 1e-5   # rest: F = I, dF = 0
    s.stepforward()
    p1, F1, dF1 = s.get_IP_info()
    assert p1.shape == (s.n_IP, 3) and F1.shape == (s.n_IP, 9) and dF1.shape == (s.n_IP, 27) and p1.dtype == np.float32
    assert np.isfinite(s.dof).all() and np.isfinite(s.dof_vel).all()
    g = np.asarray(s.gravity, float)
    drift = (p1 - p0).mean(0)
    assert drift @ g > 0 and np.abs(p1 - p0).max() < 0.05                     # falls along gravity, by a substep's worth
    with oracle.svd_mode("mcadams", sweeps=8):                              # and the algorithmic restatement (init included) takes the same step
        s2 = make_oracle_sim(cloud, opt)
        s2.stepforward()
    assert np.abs(s2.dof - s.dof).max() < 1e-4 * np.abs(s.dof - s.dof_rest).max()
