#!/usr/bin/env python
"""Where does the pipelined step spend its time?  Replays the captured sim / render graphs of harness.capture_pipelined
separately and together (MI355X probe, not part of the product).

    python tools/pipe_probe.py [--lanes 2 3 4 6 8] [--steps 300]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pienerf_amd import scene  # noqa: E402
from pienerf_amd.harness import SimRenderHarness  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--lanes", type=int, nargs="+", default=[1, 2, 3, 4, 6, 8])
ap.add_argument("--steps", type=int, default=300)
ap.add_argument("--trips", type=int, default=8)
ap.add_argument("--sim-priority", type=int, default=0)
ap.add_argument("--no-substep", action="store_true")
ap.add_argument("--sim-cus", type=int, default=0)
args = ap.parse_args()
opt = scene.default_opt()
dev = torch.device("cuda:0")


def timed(fn, n):
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(n):
        fn(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


for L in args.lanes:
    h = SimRenderHarness(opt, device=dev)
    for _ in range(20):
        h.sim.stepforward()
    h.capture_pipelined(lanes=L, n_trips=args.trips, sim_priority=args.sim_priority, sim_cus=args.sim_cus, _probe_no_substep=args.no_substep)
    be = h._pipe_backend
    for _ in range(3 * L):
        h.step_pipelined()
    h.drain_pipeline()
    full = timed(lambda i: h.step_pipelined(), args.steps)
    h.drain_pipeline()

    def sim_only(i):
        with torch.cuda.stream(be._streams["sim"]):
            be.sim_graph.replay()

    def ren_only(i):
        with torch.cuda.stream(be._streams[f"lane{i % L}"]):
            be.graph[(i % L) * be.depth].replay()
    s = timed(sim_only, args.steps)
    r = timed(ren_only, args.steps)
    print(f"lanes={L}: full {full:.3f} ms/step ({1e3 / full:.0f}/s)   sim graphs alone {s:.3f} ms   render graphs alone ({L} streams) {r:.3f} ms", flush=True)
    del h
    torch.cuda.empty_cache()
