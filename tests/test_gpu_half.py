"""BASELINE configs[4]'s numeric form — the network as the reference runs it under torch.cuda.amp.autocast (trainer.py:561 with
Trainer(fp16=True)): fp16 hash tables (gridencoder/grid.py:43-44) and fp16 matrix-core layers — HIP path through the C ABI against the
CPU oracle's restatement of the same half arithmetic (oracle.half_precision).

Tolerances, and why they are not the fp32 path's 1e-4: every half nn.Linear accumulates in float and rounds ONCE to half; the
accumulation order (cuBLAS in the reference, the matrix core here, sequential in the oracle) moves the float sum by ~1e-7 relative, which
flips the half rounding of a few outputs in a thousand by one ulp (2^-11 relative).  One ulp of the sigma logit (values in [4, 8): 2^-8)
is 0.4 % of sigma.  So: the half hash-grid features are required BIT-EXACT (their arithmetic has a fixed order), the network outputs
must be bitwise equal to the oracle's for the bulk of the samples, and the stragglers within a few half ulps."""
import numpy as np
import pytest
import torch

import oracle
from conftest import make_oracle_sim, rel_err
from pienerf_amd import scene

pytestmark = pytest.mark.gpu
DEV = "cuda"


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def amp():
    return torch.autocast("cuda", dtype=torch.float16)


def _samples(M, seed):
    rng = np.random.default_rng(seed)
    x = (rng.random((M, 3)).astype(np.float32) * 2 - 1) * 0.9
    x[:3] = [[0, 0, 0], [0.99, -0.99, 0.5], [1.5, 0, 0]]  # incl. one out-of-bound sample -> zero features
    d = rng.standard_normal((M, 3)).astype(np.float32)
    return x, d / np.linalg.norm(d, axis=-1, keepdims=True)


def test_grid_encode_half_bit_exact_vs_oracle(ckpt):
    """kernel_grid<at::Half>: the op under autocast returns half features equal, bit for bit, to the oracle's per-corner half accumulation."""
    from pienerf_amd.gridencoder import grid_encode
    rng = np.random.default_rng(2)
    u = rng.random((6000, 3)).astype(np.float32)
    u[:6] = [[0, 0, 0], [1, 1, 1], [0.5, 0.5, 0.5], [-0.1, 0.5, 0.5], [0.5, 1.2, 0.5], [0.999999, 0.999999, 0.999999]]
    emb = T(ckpt["embeddings"])
    with amp():
        got = grid_encode(T(u), emb, T(ckpt["offsets"]), ckpt["per_level_scale"], ckpt["base_resolution"])
    assert got.dtype == torch.float16 and got.shape == (6000, 32)
    ref = oracle.grid_encode_forward_half(u, ckpt["embeddings"], ckpt["offsets"], ckpt["per_level_scale"], ckpt["base_resolution"])
    assert np.array_equal(got.float().cpu().numpy(), ref)
    assert np.all(ref[3] == 0) and np.all(ref[4] == 0)
    # a half table handed in directly (what the reference kernel sees) gives the same
    got2 = grid_encode(T(u), emb.half(), T(ckpt["offsets"]), ckpt["per_level_scale"], ckpt["base_resolution"])
    assert torch.equal(got, got2)
    for gridtype, align, interp in ((1, False, 0), (0, True, 1)):
        ref = oracle.grid_encode_forward_half(u, ckpt["embeddings"], ckpt["offsets"], ckpt["per_level_scale"], ckpt["base_resolution"], gridtype, align, interp)
        with amp():
            got = grid_encode(T(u), emb, T(ckpt["offsets"]), ckpt["per_level_scale"], ckpt["base_resolution"], False, gridtype, align, interp)
        assert np.array_equal(got.float().cpu().numpy(), ref), (gridtype, align, interp)


def test_nerf_forward_half_vs_oracle_and_ops(ckpt):
    """The fused fp16 kernel (fp16 tables, v_mfma_f32_32x32x16_f16, half-rounded activations) vs the oracle's half restatement and vs the
    reference's op sequence on the GPU under autocast (HIP half grid encoder + torch half Linear)."""
    from pienerf_amd.nerf.network import NeRFNetwork
    M = 20003
    x, d = _samples(M, 5)
    with oracle.half_precision():
        s_ref, c_ref = oracle.nerf_forward(x, d, ckpt, 1.0)
    net = NeRFNetwork(encoding="hashgrid", bound=1.0, cuda_ray=True).to(DEV).load_checkpoint_dict(ckpt)
    with torch.no_grad(), amp():
        s, c = net(T(x), T(d))
        s2, c2 = net.forward_ops(T(x), T(d))
        dn = net.density(T(x))
    assert c.dtype == torch.float16 and s.dtype == torch.float32 and c2.dtype == torch.float16
    s, c, s2, c2 = s.cpu().numpy(), c.float().cpu().numpy(), s2.float().cpu().numpy(), c2.float().cpu().numpy()
    assert np.all(np.isfinite(s)) and np.all(np.isfinite(c))
    # sigma = expf(half logit) in float: the device's expf and glibc's differ in the last float bit, so "same logit" is |rel| < 1e-6
    eq_s, eq_c = np.mean(np.abs(s / s_ref - 1) < 1e-6), np.mean(c == c_ref)
    print(f"fused fp16 vs oracle: sigma same-logit {eq_s:.4f}, max rel {np.abs(s / s_ref - 1).max():.2e}; rgb bitwise equal {eq_c:.4f}, "
          f"max abs {np.abs(c - c_ref).max():.2e}; torch half ops vs oracle: sigma equal {np.mean(np.abs(s2 / s_ref - 1) < 1e-6):.4f}, rgb equal {np.mean(c2 == c_ref):.4f}")
    assert eq_s > 0.97 and np.abs(s / s_ref - 1).max() < 1.2e-2        # <= 3 ulps of a half logit in [4, 8)
    assert eq_c > 0.97 and np.abs(c - c_ref).max() < 3e-3              # <= 3 half ulps at 0.5 .. 1
    assert np.mean(np.abs(s2 / s_ref - 1) < 1e-6) > 0.95 and np.abs(s2 / s_ref - 1).max() < 1.2e-2 and np.abs(c2 - c_ref).max() < 3e-3
    # density(): same sigma, and the 15 geometry features are half values
    assert np.array_equal(dn["sigma"].cpu().numpy(), s) and dn["geo_feat"].dtype == torch.float16
    # the flag is not a no-op: the fp32 path gives different (more accurate) numbers
    with torch.no_grad():
        s32, c32 = net(T(x), T(d))
    assert 1e-5 < np.abs(s / s32.cpu().numpy() - 1).max() < 5e-2 and c32.dtype == torch.float32


def test_nerf_forward_half_is_bit_reproducible(ckpt):
    from pienerf_amd.nerf.network import NeRFNetwork
    M = 600_001
    x, d = _samples(M, 11)
    x, d = T(x), T(d)
    net = NeRFNetwork(encoding="hashgrid", bound=1.0, cuda_ray=True).to(DEV).load_checkpoint_dict(ckpt)
    with torch.no_grad(), amp():
        s0, c0 = net(x, d)
        for _ in range(30):
            s, c = net(x, d)
            assert torch.equal(s, s0) and torch.equal(c, c0)


def test_weight_refresh_in_place_and_during_capture(ckpt):
    """pn_net_update: a parameter change refreshes the packed weight images (fp32-split and fp16) and the fp16 tables in place — same
    context handle, new outputs equal to a freshly built network's — and is refused during stream capture."""
    from pienerf_amd.nerf.network import NeRFNetwork
    x, d = _samples(5000, 7)
    x, d = T(x), T(d)
    net = NeRFNetwork(encoding="hashgrid", bound=1.0, cuda_ray=True).to(DEV).load_checkpoint_dict(ckpt)
    with torch.no_grad():
        a32 = net(x, d)
        with amp():
            a16 = net(x, d)
        h0 = net._net.value
        net.sigma_net[1].weight.mul_(1.01)
        net.color_net[0].weight.add_(0.003)
        net.encoder.embeddings.mul_(0.97)
        b32 = net(x, d)
        with amp():
            b16 = net(x, d)
        assert net._net.value == h0                                              # refreshed, not rebuilt
        assert not torch.equal(a32[0], b32[0]) and not torch.equal(a16[1], b16[1])
        ck2 = dict(ckpt, embeddings=net.encoder.embeddings.cpu().numpy(), W1=net.sigma_net[1].weight.cpu().numpy(), W2=net.color_net[0].weight.cpu().numpy())
        fresh = NeRFNetwork(encoding="hashgrid", bound=1.0, cuda_ray=True).to(DEV).load_checkpoint_dict(ck2)
        c32 = fresh(x, d)
        with amp():
            c16 = fresh(x, d)
        assert torch.equal(b32[0], c32[0]) and torch.equal(b32[1], c32[1]) and torch.equal(b16[0], c16[0]) and torch.equal(b16[1], c16[1])
        # a refresh cannot be part of a captured graph: clear error instead of a silently stale replay
        net.sigma_net[0].weight.mul_(1.001)
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with pytest.raises(RuntimeError, match="captured"):
            with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
                net(x, d)
        torch.cuda.synchronize()
        net(x, d)  # outside capture the refresh goes through


@pytest.mark.parametrize("num_seek_IP,W", [(3, 96), (1, 64)])
def test_render_deformed_half_frame(deformed_ip_state, small_opt, ckpt, num_seek_IP, W):
    """A whole deformed frame under autocast (opt fp16: fused driver with the fp16 network kernel) against the oracle's frame with the
    half network.  The march is unchanged (bit-exact samples); sigma can differ by a half ulp of its logit on isolated samples, which can
    move a ray's T_thresh exit by one sample: pixels agree to 1e-2 worst case, 5e-4 on average, and the sample count to 0.5 %."""
    from pienerf_amd.nerf.network import NeRFNetwork
    opt = dict(small_opt, W=W, H=W, num_seek_IP=num_seek_IP)
    o, d = oracle.get_rays(scene.orbit_pose(opt["radius"]), scene.orbit_intrinsics(W, W, opt["fovy"]), W, W)
    with oracle.half_precision():
        ref = oracle.render_deformed(o, d, deformed_ip_state, ckpt, opt)
    ref32 = oracle.render_deformed(o, d, deformed_ip_state, ckpt, opt)
    net = NeRFNetwork(encoding="hashgrid", bound=1.0, cuda_ray=True).to(DEV).load_checkpoint_dict(ckpt)
    st = deformed_ip_state
    net.p_def, net.p_ori, net.IP_F, net.IP_dF, net.IP_dx = T(st["p_def"]), T(st["p_ori"]), T(st["F"]), T(st["dF"]), float(st["IP_dx"])
    with torch.no_grad(), amp():
        out = net.render_deformed(T(o)[None], T(d)[None], collect_stats=True, **opt)
        ops = net.rund_cuda_ops(T(o)[None], T(d)[None], **opt)
    s = net.last_stats
    img = out["image"][0].cpu().numpy()
    assert abs(s["samples"] - ref["samples"]) <= 0.005 * ref["samples"] + 8 and s["alive_at_exit"] == 0
    assert np.abs(img - ref["image"]).max() < 1e-2 and np.abs(img - ref["image"]).mean() < 5e-4
    assert np.abs(out["weights_sum"].cpu().numpy() - ref["weights_sum"]).max() < 1e-2
    assert rel_err(out["depth_0"][0].cpu().numpy(), ref["depth_0"]) < 1e-2
    assert np.abs(img - ref32["image"]).max() > 1e-4                  # fp16 really is in effect ...
    assert np.abs(img - ref32["image"]).mean() < 2e-3                 # ... and is a faithful low-precision version of the fp32 frame
    # fused driver and the op-by-op loop launch the same fp16 kernel on the same samples: identical bits
    assert torch.equal(out["image"], ops["image"]) and torch.equal(out["depth_0"], ops["depth_0"])


def test_harness_fp16_option_pipelined_equals_eager(small_cloud, small_opt, ckpt):
    """opt['fp16'] (Trainer(fp16=True)) through the harness: eager steps, and the captured pipelined form (the autocast state is read when
    the graphs are captured), give the same frames bit for bit; they differ from the fp32 harness."""
    from pienerf_amd.harness import SimRenderHarness
    opt = dict(small_opt, W=64, H=64, fp16=True)
    eager = SimRenderHarness(opt, cloud=small_cloud, ckpt=ckpt, device=DEV)
    pipe = SimRenderHarness(opt, cloud=small_cloud, ckpt=ckpt, device=DEV).capture_pipelined(lanes=2, n_trips=8)
    f32 = SimRenderHarness(dict(opt, fp16=False), cloud=small_cloud, ckpt=ckpt, device=DEV)
    want = [eager.step()["image"].clone() for _ in range(5)]
    eager.synchronize()
    got = []
    for f in range(5):
        got += [(i, r["image"].copy()) for i, r in pipe.step_pipelined()]
    got += [(i, r["image"].copy()) for i, r in pipe.drain_pipeline()]
    assert [i for i, _ in got] == list(range(5))
    for f in range(5):
        assert np.array_equal(got[f][1], want[f][0].cpu().numpy()), f
    a = f32.step()["image"]
    assert 1e-4 < (a - want[0]).abs().max() < 2e-2


def test_weight_refresh_is_refused_while_frames_are_in_flight(small_cloud, small_opt, ckpt):
    """The packed weight image is refreshed in place on the CURRENT stream only; renders in flight on the pipeline's lanes would read it half
    written (round-2 advisor finding).  The harness tells the network how many frames are in flight and the refresh is refused until the
    pipeline is drained; an autocast dtype other than float16 is refused as well (the grid encoder and the layers would disagree)."""
    from pienerf_amd.harness import SimRenderHarness
    opt = dict(small_opt, W=48, H=48)
    h = SimRenderHarness(opt, cloud=small_cloud, ckpt=ckpt, device=DEV).capture_pipelined(lanes=2, n_trips=8)
    for _ in range(3):
        h.step_pipelined()
    with torch.no_grad():
        h.model.sigma_net[0].weight.mul_(1.0)            # bumps the parameter's version: the packed image is stale now
    x = torch.rand(100, 3, device=DEV) * 2 - 1
    d = torch.nn.functional.normalize(torch.randn(100, 3, device=DEV), dim=-1)
    with pytest.raises(RuntimeError, match="drain_pipeline"):
        with torch.no_grad():
            h.model(x, d)
    h.drain_pipeline()
    with torch.no_grad():
        h.model(x, d)                                    # drained: the refresh goes through
        with pytest.raises(RuntimeError, match="float16"):
            with torch.autocast("cuda", dtype=torch.bfloat16):
                h.model(x, d)
