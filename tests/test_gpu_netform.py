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
