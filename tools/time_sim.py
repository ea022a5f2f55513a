#!/usr/bin/env python
"""stepforward(sim_iters) of the chair simulator alone on the GPU: ms per substep from a captured graph (what the frame pipeline replays).
    python tools/time_sim.py [--iters 10] [--reps 300]        (environment knobs of csrc/pn_sim.hip select experimental variants)"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pienerf_amd import scene  # noqa: E402
from pienerf_amd.simulator.solver import Simulator  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--reps", type=int, default=300)
ap.add_argument("--persistent", action="store_true", help="the local/global iterations as one cooperative kernel (pn_sim_stepforward_coop)")
args = ap.parse_args()
o = scene.default_opt()
c = scene.make_chair_points(hgs=o["hash_grid_size"])
sim = Simulator(dt=o["sim_dt"], iters=args.iters, bbox=torch.tensor([2.0 * o["bound"]] * 3), dx=o["sim_dx"], stiff=o["sim_stiff"], base=torch.tensor([-o["bound"]] * 3),
                device="cuda:0", persistent=args.persistent)
sim.InitializeFromArrays(c["pos"], c["mass"], c["mu"], c["lam"], c["pin"])
for _ in range(20):
    sim.stepforward()
torch.cuda.synchronize()
s = torch.cuda.Stream()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s):
    sim.stepforward()
with torch.cuda.stream(s):
    for _ in range(20):
        g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(args.reps):
        g.replay()
    e1.record(s)
s.synchronize()
ms = e0.elapsed_time(e1) / args.reps
disp = float((sim.dof - sim.dof_rest).abs().max())
if args.persistent and sim._coop is not None and int(os.environ.get("PN_SIM_COOP_DBG", "0")) & 4:
    import ctypes as C
    from pienerf_amd._lib import lib, ptr
    ticks = (C.c_uint64 * 9)()
    lib().pn_sim_coop_clocks(ptr(sim._coop[0]), ticks)
    launches = 20 + 1 + 20 + args.reps  # eager steps, capture warm-up... every launch since prepare
    names = ["integration points", "exchange 1", "pieces", "exchange 2", "assembly", "rows", "exchange 3", "assembly: loads, until the rank passes", "kernel start (fill)"]
    tot = sum(ticks)
    print("phase clocks of workgroup 0, us per iteration (100 MHz ticks / launches / iters):")
    for nme, tk in zip(names, ticks):
        print(f"   {nme:40s} {tk * 0.01 / launches / args.iters:7.2f} us per iteration")
if args.persistent:
    print("persistent:", sim._coop is not None and (sim._coop[1], list(sim._coop[2])), "timed out:", sim.persistent_timed_out())
print(f"stepforward({args.iters}): {ms:.4f} ms per substep, {ms / max(args.iters, 1) * 1e3:.1f} us per local/global iteration; n_k {sim.n_k}, n_IP {sim.n_IP}; max |dof - rest| {disp:.4e}",
      {k: v for k, v in os.environ.items() if k.startswith("PN_SIM")})
