#!/usr/bin/env python
"""Soak of the pipelined simulate-and-render form: N frames twice from scratch; reports sustained steps/s and whether the final image and
DOF vector of the two runs are bit-identical (every kernel on the path is order-deterministic).

    python tools/soak.py [--frames 3000] [--lanes 3]
"""
import argparse
import hashlib
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pienerf_amd import scene  # noqa: E402
from pienerf_amd.harness import SimRenderHarness  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=3000)
ap.add_argument("--lanes", type=int, default=3)
args = ap.parse_args()
runs = []
for _ in range(2):
    h = SimRenderHarness(scene.default_opt(), device="cuda:0")
    with torch.no_grad():
        h.capture_pipelined(lanes=args.lanes, n_trips=8)
        for _ in range(20):
            h.step_pipelined()
        torch.cuda.synchronize()
        t0 = time.time()
        last = None
        for _ in range(args.frames):
            for last in h.step_pipelined():
                pass
        for last in h.drain_pipeline():
            pass
        dt = time.time() - t0
        out = last[1]
    sha = lambda t: hashlib.sha1(t.detach().cpu().numpy().tobytes()).hexdigest()
    runs.append(dict(steps_per_s=round(args.frames / dt, 1), image_sha=hashlib.sha1(out["image"].tobytes()).hexdigest(), continued=h._pipe_backend.continued, dof_sha=sha(h.sim.dof),
                     finite=bool(torch.isfinite(h.sim.dof).all()), status=h.model.render_status(slot=0)))
    del h
    torch.cuda.empty_cache()
print(json.dumps({"frames": args.frames, "lanes": args.lanes, "runs": runs,
                  "bit_identical": runs[0]["image_sha"] == runs[1]["image_sha"] and runs[0]["dof_sha"] == runs[1]["dof_sha"]}))
