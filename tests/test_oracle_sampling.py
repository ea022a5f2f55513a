"""CPU checks of oracle/sampling.py (restatement of main_sample.py) by independent properties: the reference has no test for it."""
import numpy as np

from oracle import sampling as osamp
from pienerf_amd import scene


def _run(res=20, coeff=0.55):
    ck = scene.make_checkpoint(bound=1.0, seed=0, shaped=True)
    opt = scene.default_opt(sub_res=res, sub_coeff=coeff, density_threshold=0.05, sim_dx=0.1)
    rand = np.random.default_rng(5).random((512, 3)).astype(np.float32)
    return ck, opt, rand, osamp.sample(opt, lambda p: osamp.nerf_density(p, ck, 1.0)[0], rand)


def test_lattice_order_and_extent():
    opt = scene.default_opt(sub_res=7)
    p = osamp.lattice(opt).reshape(7, 7, 7, 3)
    assert p[0, 0, 0].tolist() == [-1, -1, -1] and p[6, 6, 6].tolist() == [1, 1, 1]
    assert np.all(np.diff(p[0, 0, :, 0]) > 0) and np.all(p[:, :, 3, 0] == p[0, 0, 3, 0])   # x runs fastest (main_sample.py:225-226)
    assert np.all(np.diff(p[:, 0, 0, 2]) > 0)
    cut = osamp.lattice(scene.default_opt(sub_res=5, cut=True, cut_bounds=[-0.5, 3.0, -2.0, 0.25, -0.1, 0.1])).reshape(5, 5, 5, 3)
    assert cut[..., 0].min() == -0.5 and cut[..., 0].max() == 1.0 and cut[..., 1].min() == -1.0 and cut[..., 1].max() == 0.25   # clamped to +-bound
    assert np.allclose([cut[..., 2].min(), cut[..., 2].max()], [-0.1, 0.1])


def test_density_field_and_sampled_cloud_have_the_solids_shape():
    ck, opt, rand, (pts, vols, info) = _run()
    assert info["boundary_points"] > 50 and info["kept"] == len(pts) > 200
    assert scene.chair_solid(pts, margin=0.08).all()
    # every lattice-cell centre well inside the solid was kept
    res = opt["sub_res"]
    centres = osamp.lattice(opt) + np.float32(1.0 / res)
    deep = centres[scene.chair_solid(centres, margin=-0.05)]
    kept = {tuple(np.round(p, 5)) for p in pts}
    assert all(tuple(np.round(c, 5)) in kept for c in deep) and len(deep) > 50
    # volumes: hgs^3 shared equally by the points of a spatial-hash cell
    hgs = opt["hash_grid_size"]
    cell = np.floor((pts - (pts.min(0) - np.float32(1e-3))) / np.float32(hgs)).astype(np.int64)
    _, inv, cnt = np.unique(cell, axis=0, return_inverse=True, return_counts=True)
    assert np.allclose(vols, hgs ** 3 / cnt[inv.reshape(-1)], rtol=1e-6)
    assert 0.6 * 0.3913 < vols.sum() < 1.4 * 0.3913


def test_more_coefficient_more_boundary_points_and_reproducible():
    _, _, _, (p1, v1, i1) = _run(coeff=0.55)
    _, _, _, (p2, v2, i2) = _run(coeff=0.55)
    assert np.array_equal(p1, p2) and np.array_equal(v1, v2)
    _, _, _, (_, _, i3) = _run(coeff=1.2)
    assert i3["boundary_points"] > i1["boundary_points"]
