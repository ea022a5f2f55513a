#!/usr/bin/env python
"""Times the fused network kernel on a FIXED sample set (the samples of one chair frame rendered with the default library, saved once): for A/B runs of
builds whose numbers may differ (tools/build_variant.py).   python tools/time_net_fixed.py make | time"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from pienerf_amd import scene  # noqa: E402
from pienerf_amd.harness import SimRenderHarness  # noqa: E402

F = "/tmp/pn_samples.pt"
h = SimRenderHarness(scene.default_opt(), device="cuda:0")
with torch.no_grad():
    if sys.argv[1] == "make":
        for _ in range(20):
            h.sim.stepforward()
        out = h.step(simulate=True)
        xyz, dirs = bench.collect_samples(h.model, out["rays_o"], out["rays_d"], h.render_kwargs())
        torch.save((xyz.cpu(), dirs.cpu()), F)
        print("saved", xyz.shape)
    else:
        xyz, dirs = [t.cuda() for t in torch.load(F)]
        m = h.model
        t = bench.cuda_time_ms(lambda: m(xyz, dirs), iters=30)
        print(f"{os.environ.get('PN_LIB_PATH', 'default')}: {xyz.shape[0]} samples, network kernel {t * 1e3:.1f} us")
