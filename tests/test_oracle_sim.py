"""Pins the simulator half of the CPU oracle with independent maths (SURVEY.md §8c "What pins the build instead")."""
import numpy as np
import torch

import oracle
from conftest import make_oracle_sim, rel_err


def _polar_rotation(F):
    U, s, Vt = np.linalg.svd(F)
    if np.linalg.det(U @ Vt) < 0:
        U[:, -1] *= -1
        s[-1] *= -1
    return U @ Vt, U, s, Vt


def test_svd3_contract_vs_numpy():
    rng = np.random.default_rng(0)
    mats = [np.eye(3) + 0.3 * rng.standard_normal((3, 3)) for _ in range(200)]
    mats += [m @ np.diag([1, 1, -1]) for m in mats[:50]]                    # inverted (det < 0)
    mats += [np.diag([2.0, 0.5, 1e-9]), np.outer([1, 2, 3], [0.5, -1, 2.0]), np.zeros((3, 3)), np.eye(3)]  # near-singular, rank 1, zero, identity
    for F in mats:
        U, s, V = oracle.svd3(F)
        assert abs(np.linalg.det(U) - 1) < 1e-9 and abs(np.linalg.det(V) - 1) < 1e-9  # proper rotations (wp.svd3 contract)
        assert np.abs(U @ U.T - np.eye(3)).max() < 1e-12 and np.abs(V @ V.T - np.eye(3)).max() < 1e-12
        assert np.abs(U @ np.diag(s) @ V.T - F).max() < 1e-12 * max(1.0, np.abs(F).max())
        assert s[0] >= s[1] >= abs(s[2]) - 1e-15
        if abs(np.linalg.det(F)) > 1e-6:
            R_np, _, s_np, _ = _polar_rotation(F)
            assert np.abs(U @ V.T - R_np).max() < 1e-9           # R = U V^T is the polar rotation, inversion-safe
            assert np.allclose(s, s_np, atol=1e-10)
            assert np.sign(s[2]) == np.sign(np.linalg.det(F))


def test_volume_invariant_project():
    rng = np.random.default_rng(1)
    for _ in range(100):
        s = np.exp(0.3 * rng.standard_normal(3))
        p = oracle.volume_invariant_project(s)
        # three fixed-point iterations towards the surface prod(sigma) = 1 (func_utils.py:21-40): the constraint violation shrinks
        assert abs(np.prod(p) - 1) < 0.35 * abs(np.prod(s) - 1) + 1e-12
    assert np.allclose(oracle.volume_invariant_project(np.ones(3)), 1.0)
    s = np.array([1.05, 0.98, 1.01])
    # first iteration by hand: D = -(C / |dC|^2) dC
    C = np.prod(s) - 1
    dC = np.array([s[1] * s[2], s[0] * s[2], s[0] * s[1]])
    one = s - C / (dC @ dC) * dC
    assert abs(np.prod(oracle.volume_invariant_project(s)) - 1) < abs(np.prod(one) - 1) + 1e-15


def test_shape_functions_reproduce_affine_fields(oracle_sim):
    s = oracle_sim
    # partition of unity of the value shape functions (slot 0)
    assert np.abs(s.IP_Nx[:, :, 0].sum(1) - 1).max() < 1e-12 and np.abs(s.pts_Nx[:, :, 0].sum(1) - 1).max() < 1e-12
    # rest DOFs (translation = kernel position, affine = I) reproduce the identity map: pos = p, F = I, dF = 0
    pos, F, dF = s.get_IP_info()
    assert np.abs(pos - s.IP_pos.numpy()).max() < 1e-6
    assert np.abs(F.reshape(-1, 3, 3) - np.eye(3)).max() < 1e-6 and np.abs(dF).max() < 1e-5
    assert np.abs(s.update_pos() - s.pos.numpy()).max() < 1e-12
    # an arbitrary affine field u(p) = A p + b is reproduced exactly by dof_k = [A q_k + b ; A columns ; 0]
    rng = np.random.default_rng(2)
    A, b = np.eye(3) + 0.2 * rng.standard_normal((3, 3)), rng.standard_normal(3)
    kp = s.kernel_pos.numpy()
    dof = np.zeros((s.n_k, 10, 3))
    dof[:, 0, :] = kp @ A.T + b
    for x in range(3):
        dof[:, 1 + x, :] = A[:, x]
    keep = s.dof
    s.dof = dof.reshape(-1, 3).copy()
    pos, F, dF = s.get_IP_info()
    s.dof = keep
    assert np.abs(pos - (s.IP_pos.numpy() @ A.T + b)).max() < 1e-5
    Fm = F.reshape(-1, 3, 3).transpose(0, 2, 1)  # flat layout is column-major: F_flat[c*3+r] (solver.py:423)
    assert np.abs(Fm - A).max() < 1e-5 and np.abs(dF).max() < 1e-4


def test_system_matrices(oracle_sim):
    s = oracle_sim
    lst = (s.active[:, None] * 10 + np.arange(10)[None, :]).reshape(-1)
    A = s.A[np.ix_(lst, lst)]
    assert np.abs(A - A.T).max() < 1e-9 * np.abs(A).max()
    assert np.linalg.eigvalsh(A).min() > -1e-6 * np.abs(A).max()  # PSD up to roundoff; +1e-3 I makes it SPD (solver.py:507)
    Ai = s.Ainv[np.ix_(lst, lst)]
    # cond(A + 1e-3 I) ~ 1e9 (largest eigenvalue ~1e6 over the 1e-3 regulariser), so the residual sits near 1e9 * eps
    assert np.abs(Ai @ (A + 1e-3 * np.eye(len(lst))) - np.eye(len(lst))).max() < 5e-3
    inactive = np.setdiff1d(np.arange(s.n_k), s.active)
    if len(inactive):
        rows = (inactive[:, None] * 10 + np.arange(10)[None, :]).reshape(-1)
        assert np.all(s.Ainv[rows] == 0) and np.all(s.Ainv[:, rows] == 0)
    # total mass: sum over translation DOFs of M equals sum(m) / dt^2 (partition of unity)
    M00 = s.Mmat[0::10, 0::10]
    assert abs(M00.sum() * s.dt ** 2 - s.mass.numpy().sum()) < 1e-9 * s.mass.numpy().sum()
    assert abs((s.IP_rho * s.dx ** 3).sum() - s.mass.numpy().sum()) < 1e-9 * s.mass.numpy().sum()
    # gravity load: translation rows sum to total weight
    assert np.allclose(s.rhs_gravity.reshape(s.n_k, 10, 3)[:, 0, :].sum(0), s.mass.numpy().sum() * s.gravity.numpy(), rtol=1e-10)


def test_kron_structure_equals_reference_dense_form(oracle_sim):
    """`global_matrix @ rhs` with the literal (30 n_k)^2 matrix (solver.py:493-496) == A applied per xyz component."""
    s = oracle_sim
    if s.n_k > 90:
        return
    G, M = s.full_matrices()
    rng = np.random.default_rng(3)
    x = rng.standard_normal((s.n_k * 10, 3))
    assert np.abs(G @ x.reshape(-1) - oracle.matvec3(s.Ainv, x).reshape(-1)).max() < 1e-9 * np.abs(G).max() * 10
    assert np.abs(M @ x.reshape(-1) - oracle.matvec3(s.Mmat, x).reshape(-1)).max() < 1e-9 * np.abs(M).max() * 10


def _literal_step(s, dof, vel, dof_f):
    """solver.py:574-602 transcribed with numpy on flat [30 n_k] vectors (the oracle's own C++ step is checked against this)."""
    RF_VF = lambda d: oracle.calc_elastic(s.IP_kernel.numpy(), s.IP_dNx, d.reshape(-1, 3))[:2]
    build = lambda d: oracle.collect_rhs_IP(s.dx, s.IP_kernel.numpy(), s.IP_mu, s.IP_lam, s.IP_dNx, *RF_VF(d), s.n_k * 10).reshape(-1)
    A3 = lambda A, v: (A @ v.reshape(-1, 3)).reshape(-1)
    tilde = dof + s.dt * vel
    momentum = A3(s.Mmat, tilde) + dof_f + s.rhs_gravity.reshape(-1)
    last = dof.copy()
    for _ in range(s.iters):
        rhs = momentum + build(dof) - s.rhs_rest.reshape(-1)
        dof = s.dof_rest.reshape(-1) + A3(s.Ainv, rhs)
    return dof, (dof - last) / s.dt * 0.998


def test_stepforward_matches_literal_transcription(small_cloud, small_opt):
    s = make_oracle_sim(small_cloud, small_opt)
    s.update_force(s.n_IP // 3, np.array([50.0, -20.0, 80.0]))
    dof, vel = s.dof.reshape(-1).copy(), s.dof_vel.reshape(-1).copy()
    for step in range(3):
        dof, vel = _literal_step(s, dof, vel, s.dof_f.reshape(-1))
        s.stepforward()
        scale = np.abs(s.dof - s.dof_rest).max()
        assert np.abs(s.dof.reshape(-1) - dof).max() < 1e-9 * max(scale, 1e-6)
        assert rel_err(s.dof_vel.reshape(-1), vel) < 1e-8
    assert np.abs(s.dof - s.dof_rest).max() > 1e-4


def test_rest_is_equilibrium_without_loads(small_cloud, small_opt):
    s = make_oracle_sim(small_cloud, small_opt)
    s.rhs_gravity[:] = 0
    d0 = s.dof.copy()
    for _ in range(2):
        s.stepforward()
    assert np.abs(s.dof - d0).max() < 1e-8 and np.abs(s.dof_vel).max() < 1e-6


def test_update_force_distributes_a_point_load(oracle_sim):
    s = oracle_sim
    f = np.array([1.0, 2.0, -3.0])
    vid = 5
    s.update_force(vid, f)
    tr = s.dof_f.reshape(s.n_k, 10, 3)[:, 0, :].sum(0)
    m = s.IP_rho[vid] * s.dx ** 3
    assert np.allclose(tr, m * f, rtol=1e-10)  # value shape functions sum to one
    assert np.count_nonzero(np.abs(s.dof_f).sum(1)) <= 80
    s.clear_force()
    assert np.all(s.dof_f == 0)


def test_elastic_rhs_vanishes_for_rigid_motion(oracle_sim):
    """R = rotation, sigma = 1 -> F = R, V = R, so the elastic rhs equals the rest rhs rotated: b(Q x) = Q b(x)."""
    s = oracle_sim
    th = 0.3
    Q = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1.0]])
    RF0, VF0, _ = oracle.calc_elastic(s.IP_kernel.numpy(), s.IP_dNx, s.dof_rest)
    assert np.abs(RF0 - np.eye(3)).max() < 1e-9 and np.abs(VF0 - np.eye(3)).max() < 1e-9
    dofQ = s.dof_rest @ Q.T
    RF, VF, FF = oracle.calc_elastic(s.IP_kernel.numpy(), s.IP_dNx, dofQ)
    assert np.abs(RF - Q).max() < 1e-9 and np.abs(VF - Q).max() < 1e-9 and np.abs(FF - Q).max() < 1e-9
    b0 = oracle.collect_rhs_IP(s.dx, s.IP_kernel.numpy(), s.IP_mu, s.IP_lam, s.IP_dNx, RF0, VF0, s.n_k * 10)
    bQ = oracle.collect_rhs_IP(s.dx, s.IP_kernel.numpy(), s.IP_mu, s.IP_lam, s.IP_dNx, RF, VF, s.n_k * 10)
    assert np.abs(bQ - b0 @ Q.T).max() < 1e-9 * np.abs(b0).max()
