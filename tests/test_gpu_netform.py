"""The two forms of the fp32 network's dense layers (include/pienerf_hip.h: pn_net_form): fp16 hi/lo pieces (three products per K chunk on the fp16 matrix
pipe, the default where an interval bound over the weights and tables allows it) against three bf16 pieces (six products, any weights).  Both stand for
NeRFNetwork.forward in fp32 (nerf/network.py:98-127): they agree with each other and with the sequential-fp32 oracle far inside the 1e-4 bar; weights
whose intermediates would leave fp16's range, and tiny tables (features of 1e-6), keep the form and its accuracy through the per-layer power-of-two
scales folded into the weight image; weights without a finite positive interval bound select the bf16 form by themselves."""
import os

import numpy as np
import pytest
import torch

import oracle
from conftest import rel_err
from pienerf_amd._lib import lib
from test_gpu_parity import DEV, T

pytestmark = pytest.mark.gpu


def _model(ck, form=None):
    from pienerf_amd.nerf.network import NeRFNetwork
    old = os.environ.get("PN_NET_FORM")
    if form:
        os.environ["PN_NET_FORM"] = form   # read by pn_net_create / pn_net_update, per network
    try:
        m = NeRFNetwork(encoding="hashgrid", bound=1.0, cuda_ray=True).to(DEV).load_checkpoint_dict(ck)
        m._net_handle()
    finally:
        if old is None:
            os.environ.pop("PN_NET_FORM", None)
        else:
            os.environ["PN_NET_FORM"] = old
    return m


def _scaled(ckpt, tables=1.0, w0=1.0):
    """w0: the first layer's weights times w0 and the second layer's divided by it (ReLU is positively homogeneous: the same network, other intermediates)."""
    ck = dict(ckpt)
    ck["embeddings"] = (ckpt["embeddings"] * np.float32(tables)).astype(np.float32)
    ck["W0"] = (ckpt["W0"] * np.float32(w0)).astype(np.float32)
    ck["W1"] = (ckpt["W1"] / np.float32(w0)).astype(np.float32)
    return ck


def _samples(n, seed):
    rng = np.random.default_rng(seed)
    x = ((rng.random((n, 3)) * 2 - 1) * 0.95).astype(np.float32)
    d = rng.standard_normal((n, 3)).astype(np.float32)
    return x, d / np.linalg.norm(d, axis=-1, keepdims=True)


def test_both_forms_agree_with_each_other_and_the_oracle(ckpt):
    x, d = _samples(60_001, 3)
    mx, mb = _model(ckpt), _model(ckpt, "bf16")
    assert lib().pn_net_form(mx._net) == 2 and lib().pn_net_form(mb._net) == 0
    with torch.no_grad():
        sx, cx = [t.cpu().numpy() for t in mx(T(x), T(d))]
        sb, cb = [t.cpu().numpy() for t in mb(T(x), T(d))]
    want_s, want_c = oracle.nerf_forward(x, d, ckpt, 1.0)
    for name, s, c in (("fp16 hi/lo", sx, cx), ("bf16 x 3", sb, cb)):
        es, ec = float(np.abs(s / want_s - 1).max()), float(np.abs(c - want_c).max())
        print(f"{name}: sigma rel {es:.2e}, rgb abs {ec:.2e}")
        assert es < 2e-5 and ec < 2e-6, (name, es, ec)
    assert float(np.abs(sx / sb - 1).max()) < 2e-5 and float(np.abs(cx - cb).max()) < 2e-6


def test_weights_that_would_overflow_fp16_unscaled_keep_the_form_and_the_accuracy(ckpt):
    """The first layer's outputs reach 1e5 and more: no fp16 piece could hold them as they are; carried at the per-layer scale of net_choose_form they sit
    where every other network's do, and the form's accuracy is that of the unscaled network (power-of-two factors are exact)."""
    x, d = _samples(8_000, 5)
    ck = _scaled(ckpt, w0=32768.0)
    big = _model(ck)
    assert lib().pn_net_form(big._net) == 2
    with torch.no_grad():
        s, c = [t.cpu().numpy() for t in big(T(x), T(d))]
    want_s, want_c = oracle.nerf_forward(x, d, ck, 1.0)
    assert np.isfinite(s).all() and np.isfinite(c).all() and np.isfinite(want_s).all()
    assert float(np.abs(c - want_c).max()) < 2e-6
    fin = want_s > 0
    assert float(np.abs(s[fin] / want_s[fin] - 1).max()) < 2e-5


def test_weights_without_a_finite_bound_take_the_bf16_form(ckpt):
    """An all-zero layer (bound 0) or an infinite weight: no scale to choose, the three-way bf16 split runs and gives what fp32 arithmetic gives."""
    x, d = _samples(4_000, 6)
    ck = dict(ckpt)
    ck["W1"] = np.zeros_like(ckpt["W1"])
    zero = _model(ck)
    assert lib().pn_net_form(zero._net) == 0
    with torch.no_grad():
        s, c = [t.cpu().numpy() for t in zero(T(x), T(d))]
    want_s, want_c = oracle.nerf_forward(x, d, ck, 1.0)
    assert float(np.abs(s - want_s).max()) < 1e-5 and float(np.abs(c - want_c).max()) < 2e-6
    ck = dict(ckpt)
    ck["W3"] = ckpt["W3"].copy()
    ck["W3"][5, 7] = np.inf
    assert lib().pn_net_form(_model(ck)._net) == 0


def test_tiny_tables_keep_their_accuracy(ckpt):
    """Features of 1e-6 and hidden activations of 1e-5: as fp16 pieces they would be subnormal (6e-8 absolute, percents of the value); the layers' scales keep 22 bits."""
    x, _ = _samples(30_000, 7)
    ck = _scaled(ckpt, tables=1e-5)
    tiny_x, tiny_b = _model(ck), _model(ck, "bf16")
    assert lib().pn_net_form(tiny_x._net) == 2
    with torch.no_grad():
        a, b = tiny_x.density(T(x))["geo_feat"].cpu().numpy(), tiny_b.density(T(x))["geo_feat"].cpu().numpy()
    assert np.abs(b).max() > 0 and rel_err(a, b) < 2e-5, rel_err(a, b)


def test_captured_graphs_follow_an_in_place_weight_refresh_that_moves_the_scales(small_cloud, small_opt, ckpt):
    """Round-4 advisor (high): the per-layer power-of-two scales of the fp16 hi/lo form were by-value kernel arguments, so graphs captured before an in-place
    weight refresh (pn_net_update) replayed the new weight image with the old scales — a sigma logit off by 2^k.  They now live in device memory beside the
    image.  Capture, drain, multiply the tables by 8 and divide W0 by 8 (the features' scale xs[0] moves by 2^-3), double W1 (the density net's output scale
    xs[2] moves, the sigma logit doubles), refresh in place, replay: the replayed frames equal an eager harness built on the new weights, bit for bit."""
    from pienerf_amd.harness import SimRenderHarness
    opt = dict(small_opt, W=48, H=48)
    h = SimRenderHarness(opt, cloud=small_cloud, ckpt=ckpt, device=DEV).capture_pipelined(lanes=2, n_trips=8)
    for _ in range(3):
        h.step_pipelined()
    h.drain_pipeline()
    assert lib().pn_net_form(h.model._net) == 2
    with torch.no_grad():
        h.model.encoder.embeddings.mul_(8.0)
        h.model.sigma_net[0].weight.mul_(0.125)
        h.model.sigma_net[1].weight.mul_(2.0)
        x, d = _samples(64, 1)
        h.model(T(x), T(d))                  # the refresh (in place, same handle)
    assert lib().pn_net_form(h.model._net) == 2 and lib().pn_net_form_epoch(h.model._net) == 0
    ck2 = dict(ckpt)
    ck2["W0"] = (ckpt["W0"] * np.float32(0.125)).astype(np.float32)
    ck2["W1"] = (ckpt["W1"] * np.float32(2.0)).astype(np.float32)
    ck2["embeddings"] = (ckpt["embeddings"] * np.float32(8.0)).astype(np.float32)
    first = h._pipe.frame
    got = []
    for _ in range(3):
        got += [(i, r["image"].copy()) for i, r in h.step_pipelined()]
    got += [(i, r["image"].copy()) for i, r in h.drain_pipeline()]
    got = [g for g in got if g[0] >= first]
    assert len(got) == 3
    # every frame f is rendered from the state before substep f (the substeps do not depend on the network): an eager harness on the new weights
    want_h = SimRenderHarness(opt, cloud=small_cloud, ckpt=ck2, device=DEV)
    for f in range(first + 3):
        out = want_h.step()
        if f >= first:
            want = out["image"][0].cpu().numpy()
            assert got[f - first][1].shape == want.shape and np.array_equal(got[f - first][1], want), (f, float(np.abs(got[f - first][1] - want).max()))


def test_graphs_captured_under_another_network_form_are_refused(small_cloud, small_opt, ckpt):
    """The form (fp16 hi/lo or bf16 pieces) selects the kernel template and the weight image and is fixed in a captured launch: a refresh that flips it
    (here: an all-zero layer has no positive interval bound) bumps pn_net_form_epoch and the harness refuses to replay the stale graphs."""
    from pienerf_amd.harness import SimRenderHarness
    opt = dict(small_opt, W=32, H=32)
    h = SimRenderHarness(opt, cloud=small_cloud, ckpt=ckpt, device=DEV).capture_pipelined(lanes=2, n_trips=8)
    h.step_pipelined()
    h.drain_pipeline()
    with torch.no_grad():
        h.model.color_net[1].weight.zero_()
        x, d = _samples(64, 1)
        h.model(T(x), T(d))
    assert lib().pn_net_form(h.model._net) == 0 and lib().pn_net_form_epoch(h.model._net) == 1
    with pytest.raises(RuntimeError, match="capture again"):
        h.step_pipelined()
    h.capture_pipelined(lanes=2, n_trips=8)      # recapture: runs
    h.step_pipelined()
    h.drain_pipeline()


def test_heavy_tailed_weights_and_an_outlier_table_entry(ckpt):
    """Round-4 advisor (low): the form is chosen from interval bounds, which heavy-tailed weights or one outlier table entry push far above the typical
    magnitudes (and the slack compounds layer by layer).  Student-t (3 degrees of freedom) weights — single entries 10-30 standard deviations out — and a
    table whose typical entry is 1e-2 with one entry at 40: whichever form net_choose_form takes, the outputs stay inside the 1e-4 bar of north_star
    against the sequential-fp32 oracle, and the two forms agree."""
    rng = np.random.default_rng(11)
    ck = dict(ckpt)
    ck["embeddings"] = (ckpt["embeddings"] * np.float32(0.02)).astype(np.float32)
    ck["embeddings"][12345, 1] = 40.0
    for k, fan in (("W0", 32), ("W1", 64), ("W2", 31), ("W3", 64), ("W4", 64)):
        t = rng.standard_t(3, size=ckpt[k].shape) * np.sqrt(2.0 / fan) / np.sqrt(3.0)
        ck[k] = t.astype(np.float32)
    ck["W1"][0, :] *= 0.1   # keep exp(logit) finite
    x, d = _samples(40_000, 9)
    mx, mb = _model(ck), _model(ck, "bf16")
    with torch.no_grad():
        sx, cx = [t.cpu().numpy() for t in mx(T(x), T(d))]
        sb, cb = [t.cpu().numpy() for t in mb(T(x), T(d))]
    want_s, want_c = oracle.nerf_forward(x, d, ck, 1.0)
    assert np.isfinite(want_s).all() and want_s.min() > 0
    for name, s, c in ((f"chosen form {lib().pn_net_form(mx._net)}", sx, cx), ("bf16 x 3", sb, cb)):
        es, ec = float(np.abs(s / want_s - 1).max()), float(np.abs(c - want_c).max())
        print(f"{name}: sigma rel {es:.2e}, rgb abs {ec:.2e}")
        assert es < 1e-4 and ec < 1e-4, (name, es, ec)
