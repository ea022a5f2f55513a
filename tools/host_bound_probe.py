#!/usr/bin/env python
"""Is the pipelined step bound by the host?  The same pipeline (same graphs: same number of kernel nodes, same simulator) at 64 x 64 pixels, where the GPU
has next to nothing to do: its rate is what the host side (Python loop, graph launches, events, copier submissions) can enqueue per second.

    python tools/host_bound_probe.py [--lanes 3]
"""
import argparse
import cProfile
import os
import pstats
import sys
import time

import torch

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pienerf_amd import scene  # noqa: E402
from pienerf_amd.harness import SimRenderHarness  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--lanes", type=int, default=3)
ap.add_argument("--profile", action="store_true")
args = ap.parse_args()
for W in (64, 800):
    h = SimRenderHarness(scene.default_opt(W=W, H=W), device="cuda:0")
    with torch.no_grad():
        h.capture_pipelined(lanes=args.lanes, n_trips=None)
        for _ in range(60):
            h.step_pipelined()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 400
        if args.profile and W == 64:
            pr = cProfile.Profile()
            pr.enable()
        for _ in range(n):
            h.step_pipelined()
        if args.profile and W == 64:
            pr.disable()
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        h.drain_pipeline()
    print(f"{W}x{W}: host loop {1e3 * t_host / n:.3f} ms per step ({n / t_host:.0f} steps/s enqueued), incl. final sync {n / t_all:.0f} steps/s")
    if args.profile and W == 64:
        pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
    del h
    torch.cuda.empty_cache()
