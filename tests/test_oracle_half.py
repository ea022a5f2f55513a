"""fp16 form of the network (BASELINE configs[4]; the reference under torch.cuda.amp.autocast): the oracle's software half rounding and
its restatement of the half arithmetic, pinned on the CPU by torch's own half kernels; the C ABI's host-side rounding (the weight image of
the fp16 kernel) against numpy.  The HIP-vs-oracle parity tests are in tests/test_gpu_half.py."""
import ctypes as C

import numpy as np
import torch

import oracle


def _vals():
    rng = np.random.default_rng(0)
    v = np.concatenate([rng.standard_normal(20000).astype(np.float32) * np.float32(10.0) ** rng.integers(-9, 6, 20000).astype(np.float32),
                        np.float32([0.0, -0.0, 65504.0, 65519.99, 65520.0, 65536.0, 1e30, -1e30, 2.0 ** -24, 2.0 ** -25, 2.0 ** -25 * 1.0000001, 2.0 ** -26,
                                    6.1035e-5, 6.0975e-5, 5.96e-8, 1.0 + 2.0 ** -11, 1.0 + 2.0 ** -11 + 2.0 ** -20, 1.0 + 3 * 2.0 ** -11, 2047.5, 2048.5,
                                    np.inf, -np.inf])])
    # every half value and every midpoint between neighbouring halves (the ties)
    h = np.arange(0, 0x7c00, dtype=np.uint16).view(np.float16).astype(np.float32)
    mid = ((h[:-1].astype(np.float64) + h[1:].astype(np.float64)) / 2).astype(np.float32)
    return np.concatenate([v, h, -h, mid, -mid, np.nextafter(mid, np.float32(np.inf)), np.nextafter(mid, np.float32(-np.inf))])


def test_oracle_hround_is_round_to_nearest_even():
    v = _vals()
    with np.errstate(over="ignore"):
        want = v.astype(np.float16).astype(np.float32)
    got = oracle.hround(v)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_c_abi_host_float_to_half_matches_numpy():
    from pienerf_amd._lib import lib
    v = _vals()
    out = np.empty(v.size, np.uint16)
    assert lib().pn_host_float_to_half(v.ctypes.data, out.ctypes.data, v.size) == 0
    with np.errstate(over="ignore"):
        want = v.astype(np.float16).view(np.uint16)
    assert np.array_equal(out, want)


def test_oracle_half_network_matches_torch_half_layers(ckpt):
    """The oracle's half MLP arithmetic (nerf_one with nt.half) against torch's CPU half kernels applied layer by layer to the oracle's
    own half features: every Linear = half inputs x half weights -> half output.  torch accumulates in its own order, so single outputs
    may differ by one half ulp; the bulk must agree exactly."""
    rng = np.random.default_rng(3)
    x = rng.uniform(-0.95, 0.95, (4000, 3)).astype(np.float32)
    d = rng.standard_normal((4000, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    with oracle.half_precision():
        sig, rgb = oracle.nerf_forward(x, d, ckpt, 1.0)
    u = (x + np.float32(1.0)) / np.float32(2.0)
    enc = oracle.grid_encode_forward_half(u, ckpt["embeddings"], ckpt["offsets"], ckpt["per_level_scale"], ckpt["base_resolution"])
    assert np.array_equal(enc, oracle.hround(enc))                       # half values
    W = [torch.from_numpy(ckpt[f"W{i}"]).to(torch.float16) for i in range(5)]
    h = torch.from_numpy(enc).to(torch.float16)
    h = torch.relu(torch.nn.functional.linear(h, W[0]))
    h = torch.nn.functional.linear(h, W[1])
    sig_t = torch.exp(h[:, 0].float()).numpy()
    sh = torch.from_numpy(oracle.sh_encode_forward(d, 4))
    c = torch.cat([sh, h[:, 1:].float()], -1).to(torch.float16)
    c = torch.relu(torch.nn.functional.linear(c, W[2]))
    c = torch.relu(torch.nn.functional.linear(c, W[3]))
    rgb_t = torch.sigmoid(torch.nn.functional.linear(c, W[4])).float().numpy()
    assert np.array_equal(rgb, oracle.hround(rgb))                       # the colour is a half tensor
    assert np.mean(sig == sig_t) > 0.97 and np.abs(sig / sig_t - 1).max() < 1e-2      # <= a couple of half ulps of the logit (ulp(4..8) = 2^-8)
    assert np.mean(rgb == rgb_t) > 0.97 and np.abs(rgb - rgb_t).max() < 2e-3
    # and the half form really differs from the fp32 form (the flag is not a no-op) while staying close to it
    sig32, rgb32 = oracle.nerf_forward(x, d, ckpt, 1.0)
    assert 1e-5 < np.abs(rgb - rgb32).max() < 2e-2 and 1e-5 < np.abs(sig / sig32 - 1).max() < 5e-2


def test_oracle_half_grid_accumulation_rounds_per_corner(ckpt):
    """kernel_grid<at::Half>: res = Half(res + Half(w * v)) per corner — restated in float64 numpy for one dense and one hashed level."""
    rng = np.random.default_rng(5)
    u = rng.uniform(0, 1, (300, 3)).astype(np.float32)
    got = oracle.grid_encode_forward_half(u, ckpt["embeddings"], ckpt["offsets"], ckpt["per_level_scale"], ckpt["base_resolution"])
    scales, res = oracle.grid_level_params(16, ckpt["per_level_scale"], ckpt["base_resolution"])
    off = ckpt["offsets"]
    emb_h = ckpt["embeddings"].astype(np.float16)
    f16 = np.float16
    for lvl in (0, 3, 9):
        hs = int(off[lvl + 1] - off[lvl])
        table = emb_h[off[lvl]:off[lvl + 1]]
        stride = int(res[lvl]) + 1
        for b in range(300):
            pos = np.float32(np.float64(u[b].astype(np.float64) * np.float64(scales[lvl]) + 0.5).astype(np.float32))  # fmaf: one rounding
            pg = np.floor(pos).astype(np.uint32)
            fr = (pos - pg.astype(np.float32)).astype(np.float32)
            acc = np.zeros(2, f16)
            for idx in range(8):
                w = np.float32(1.0)
                pl = pg.copy()
                for dd in range(3):
                    if idx & (1 << dd):
                        w = np.float32(w * fr[dd]); pl[dd] += 1
                    else:
                        w = np.float32(w * np.float32(np.float32(1.0) - fr[dd]))
                if stride ** 3 <= hs:
                    index = int(pl[0]) + int(pl[1]) * stride + int(pl[2]) * stride * stride
                else:
                    index = (int(pl[0]) ^ (int(pl[1]) * 2654435761 & 0xffffffff) ^ (int(pl[2]) * 805459861 & 0xffffffff)) % hs
                for c in range(2):
                    prod = f16(np.float32(w * np.float32(table[index, c])))
                    acc[c] = f16(np.float32(acc[c]) + np.float32(prod))
            assert np.array_equal(got[b, 2 * lvl:2 * lvl + 2], acc.astype(np.float32)), (lvl, b)
