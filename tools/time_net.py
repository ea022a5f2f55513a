#!/usr/bin/env python
"""Times the fused network kernel and the stand-alone hash-grid kernel on the samples of one real 800x800 chair frame."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from pienerf_amd import scene  # noqa: E402
from pienerf_amd.gridencoder import grid_encode  # noqa: E402
from pienerf_amd.harness import SimRenderHarness  # noqa: E402
from pienerf_amd._lib import check, lib, ptr, stream_ptr  # noqa: E402

h = SimRenderHarness(scene.default_opt(), device="cuda:0")
for _ in range(20):
    h.sim.stepforward()
with torch.no_grad():
    out = h.step(simulate=True)
    xyz, dirs = bench.collect_samples(h.model, out["rays_o"], out["rays_d"], h.render_kwargs())
    m, enc = h.model, h.model.encoder
    B = xyz.shape[0]
    t_net = bench.cuda_time_ms(lambda: m(xyz, dirs), iters=30)
    with torch.autocast("cuda", dtype=torch.float16):  # fp16 form: fp16 tables + fp16 MFMA (BASELINE configs[4])
        t_half = bench.cuda_time_ms(lambda: m(xyz, dirs), iters=30)
    u = ((xyz + m.bound) / (2 * m.bound)).contiguous()
    res = {}
    for blm in (1, 0):
        o = torch.empty(B * 32, device="cuda")
        S = float(torch.tensor(enc.per_level_scale).log2().float())
        f = lambda: check(lib().pn_grid_encode_forward(ptr(u), ptr(enc.embeddings), enc._offsets_host.data_ptr(), ptr(o), B, 3, 2, 16, S, 16, None, 0, 0, 0, blm,
                                                       stream_ptr()), "grid")
        res[blm] = bench.cuda_time_ms(f, iters=30)
print(f"fp16 net {t_half*1e3:.1f} us ({(1068 - 512)*B/t_half/1e6:.0f} GB/s of 556 B/sample, {18688*B/t_half/1e9:.1f} TF)")
print(f"variant={os.environ.get('PN_NERF_VARIANT','0')} B={B} net {t_net*1e3:.1f} us ({1076*B/t_net/1e6:.0f} GB/s, {18688*B/t_net/1e9:.1f} TF)  "
      f"grid[B,LC] {res[1]*1e3:.1f} us ({1164*B/res[1]/1e6:.0f} GB/s)  grid[L,B,C] {res[0]*1e3:.1f} us ({1164*B/res[0]/1e6:.0f} GB/s)")
