"""Shared inputs of the svd3 tests (CPU: test_oracle_svd.py; GPU: test_gpu_simpin.py, test_gpu_edges.py): the adversarial deformation-gradient
set and the BASELINE trajectories' scenes.  Data only — no oracle, no product code."""
import numpy as np


def rot(axis, ang):
    axis = np.asarray(axis, float) / np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K


def adversarial_F(seed=0):
    """[n, 3, 3] deformation gradients that decide R = U V^T (cuda_utils.py:107-116): random, inverted (det < 0), repeated singular values,
    pure rotations, +-identity, rank 2 / 1 / 0, 1e-12- and 1e+8-scaled, nearly repeated.  The set of
    test_gpu_edges.py::test_calc_elastic_on_adversarial_deformation_gradients, in its order."""
    rng = np.random.default_rng(seed)
    Q1, Q2 = rot([1, 2, 3], 0.7), rot([-2, 1, 0.5], 2.1)
    mats = [np.eye(3) + 0.3 * rng.standard_normal((3, 3)) for _ in range(300)]
    mats += [m @ np.diag([1, 1, -1]) for m in mats[:80]]
    mats += [Q1 @ np.diag(s) @ Q2.T for s in ([2.0, 2.0, 0.5], [1.5, 1.5, 1.5], [3.0, 1.0, 1.0], [1.0, 1.0, -1.0], [2.0, 2.0, -2.0])]
    mats += [Q1, Q2, Q1 @ Q2, np.eye(3), -np.eye(3)]
    mats += [Q1 @ np.diag([2.0, 0.7, 0.0]) @ Q2.T, Q1 @ np.diag([1.3, 0.0, 0.0]) @ Q2.T, np.outer([1, 2, 3], [0.5, -1, 2.0]), np.zeros((3, 3))]
    mats += [1e-12 * (np.eye(3) + 0.2 * rng.standard_normal((3, 3))), 1e8 * (np.eye(3) + 0.2 * rng.standard_normal((3, 3))),
             Q1 @ np.diag([2.0, 0.5, 1e-9]) @ Q2.T, Q1 @ np.diag([1.0, 1.0 + 1e-13, 1.0 - 1e-13]) @ Q2.T]
    return np.stack(mats)


def elastic_inputs_for(Fs, seed=0):
    """(topo, dNx, dof) that make calc_elastic see exactly the deformation gradients Fs: one kernel per neighbour slot whose affine DOF rows are
    the identity, F[r][c] = sum_i dNx[v, i, c, 1 + r], F split over the 8 neighbours by random positive parts."""
    rng = np.random.default_rng(seed + 1)
    n = len(Fs)
    topo = np.tile(np.arange(8, dtype=np.int32), (n, 1))
    dof = np.zeros((8, 10, 3))
    for j in range(3):
        dof[:, 1 + j, j] = 1.0
    dNx = np.zeros((n, 8, 3, 10))
    part = rng.uniform(0.05, 1.0, (n, 8))
    part /= part.sum(1, keepdims=True)
    for r in range(3):
        for c in range(3):
            dNx[:, :, c, 1 + r] = Fs[:, r, c][:, None] * part
    return topo, dNx, dof.reshape(-1, 3)


def well_conditioned(Fs, gap=1e-3):
    """mask of the F whose polar rotation R and U diag(sigma') V^T are determined: non-singular AND sigma_2 + sigma_3 well away from zero (for
    det F < 0 the rotation with the smallest singular value negated is unique only while sigma_2 > sigma_3)."""
    s = np.linalg.svd(Fs, compute_uv=False)
    scale = np.maximum(s[:, 0], 1e-300)
    det = np.linalg.det(Fs)
    ok = s[:, 2] / scale > 1e-6
    inv = det < 0
    ok &= ~inv | ((s[:, 1] - s[:, 2]) / scale > gap)
    return ok


# name: (opt factory name in pienerf_amd.scene, opt overrides, cloud kwargs, force) — the scenes and forces of bench.py's make_config
TRAJECTORIES = {
    "chair": ("default_opt", {}, {}, None),
    "chair_forced": ("default_opt", {}, {}, (300.0, 100.0, -200.0)),
    "trex": ("trex_opt", dict(radius=4.5), dict(bound=2.0), (250.0, 120.0, -180.0)),
    "stress": ("stress_opt", {}, dict(sub_res=180), (400.0, -150.0, 250.0)),
}


def trajectory_scene(name):
    from pienerf_amd import scene
    mk, over, ckw, force = TRAJECTORIES[name]
    opt = getattr(scene, mk)(**over)
    cloud = scene.make_chair_points(hgs=opt["hash_grid_size"], **dict({"bound": opt["bound"]}, **ckw))
    return opt, cloud, force
