#!/usr/bin/env python
"""Phase clocks of the fused later-trips launch (csrc/pn_trips_fused.h) on the 800x800 chair: where a wave's time goes.

    python tools/fused_clocks.py [--frames 5]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pienerf_amd import scene  # noqa: E402
from pienerf_amd.harness import SimRenderHarness  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=5)
ap.add_argument("--W", type=int, default=800)
args = ap.parse_args()
opt = scene.default_opt(W=args.W, H=args.W)
h = SimRenderHarness(opt, device="cuda:0")
for _ in range(20):
    h.sim.stepforward()
for _ in range(3):
    h.step(simulate=False, collect_stats=True)
torch.cuda.synchronize()
m = h.model
print(m.last_stats, m.trip_records())
m.march_counters(4)
m.fused_clocks(reset=True)
for _ in range(args.frames):
    h.step(simulate=False)
c = m.fused_clocks()
m.march_counters(0)
n = args.frames
waves = c["waves"] / n
print(f"per frame: {waves:.0f} waves, {c['wave_rounds'] / n:.0f} wave-rounds ({c['wave_rounds'] / max(c['waves'], 1):.2f} per wave)")
keys = ("refill", "march", "windows", "network", "composite", "a_march", "a_windows", "a_network", "a_composite", "a_barrier")
tot = sum(c[k] for k in keys)
for k in keys:
    print(f"  {k:10s} {c[k] / max(c['waves'], 1) / 1e3:8.1f} kcycles per wave  {100 * c[k] / tot:5.1f} %   {c[k] / max(c['wave_rounds'], 1):8.0f} cycles per wave-round")
print(f"  total      {tot / max(c['waves'], 1) / 1e3:8.1f} kcycles per wave (~{tot / max(c['waves'], 1) / 2.4e3:.0f} us at 2.4 GHz)")
life = c["lifetime_ticks"] / max(c["waves"], 1) * 10.0  # ns
print(f"  wave lifetime: mean {life / 1e3:.1f} us, longest {c['max_lifetime_ticks'] / 100:.1f} us; most rounds of one wave {c['max_rounds']}; shader clock ~ {tot / max(c['lifetime_ticks'], 1) / 10:.2f} GHz")
