#!/usr/bin/env python
"""sim+render steps/s at 800x800 on the synthetic chair (BASELINE.json configs[1]); one JSON line on rank 0.

    python bench.py [--gpus N --steps K --warmup W]          (N > 1: launched by torch.distributed.run, one rank per GPU)

A step = one GUI-frame equivalent (nerf/gui.py:588-603 / nerf/trainer.py:300-318 of the reference):
get_rays -> get_IP_info -> stepforward(iters=10) -> render_deformed at 800x800, inputs resident in HBM, outputs left in HBM.
N = 1 replays captured HIP graphs: `--lanes` renders in flight on their own streams, the simulator running ahead on dof snapshots
(harness.capture_pipelined; --single-graph: one graph per step, one frame at a time; --eager: kernel by kernel).
N > 1 is frame-parallel (BASELINE.json configs[3]): rank 0 owns the simulator and broadcasts every dof snapshot
(<= 82 KB per frame) over RCCL; every rank holds the checkpoint and renders frames f = rank (mod N) on its own lanes
(harness.capture_frame_parallel).  K steps per rank = K*N frames in total (weak scaling); value = frames all ranks completed /
max-over-ranks time.
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")  # one hardware queue per stream (see pienerf_amd/__init__.py); before HIP initialises

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic bytes / flops per unit
HASH_BYTES_PER_SAMPLE = 1164   # 16 levels x 8 corners x 8 B gathered + 12 B position in + 128 B features out
FUSED_BYTES_PER_SAMPLE = 1068  # fused network kernel: 1024 B gathered + 4 B slot id + 24 B xyz/dir in + 16 B sigma/rgb out
MLP_FLOP_PER_SAMPLE = 18688
MARCH_BYTES = dict(iteration=8, candidate=16, warp=64, sample=32 + 4, ray_trip=40)  # cell range / list entry / record head / outputs / ray state
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
F32_MFMA_PEAK_TF = 157.3       # dense fp32-input MFMA peak


def cuda_time_ms(fn, iters=20, warmup=3):
    """Average duration of fn() on torch's current stream, HIP events (kernels are launched on that stream)."""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def cpu_baseline(opt, cloud, ckpt, budget_s=20.0):
    """The CPU oracle ("port") on this box's host cores: sim steps + full 800x800 renders for ~budget_s seconds."""
    import oracle
    from oracle.sim_init import OracleSimulator
    from pienerf_amd import scene
    t0 = time.time()
    ref = OracleSimulator(dt=opt["sim_dt"], iters=opt["sim_iters"], bbox=torch.tensor([2.0 * opt["bound"]] * 3), dx=opt["sim_dx"],
                          stiff=opt["sim_stiff"], base=torch.tensor([-opt["bound"]] * 3))
    ref.InitializeFromArrays(cloud["pos"], cloud["mass"], cloud["mu"], cloud["lam"], cloud["pin"])
    init_s = time.time() - t0
    p_ori, _, _ = ref.get_IP_info()
    pose, intr = scene.orbit_pose(opt["radius"]), scene.orbit_intrinsics(opt["W"], opt["H"], opt["fovy"])
    times = []
    t_all = time.time()
    while True:
        t = time.time()
        o, d = oracle.get_rays(pose, intr, opt["H"], opt["W"])
        p_def, F, dF = ref.get_IP_info()
        ref.stepforward()
        oracle.render_deformed(o, d, dict(p_def=p_def, p_ori=p_ori, F=F, dF=dF, IP_dx=ref.dx * 1.05), ckpt, opt)
        times.append(time.time() - t)
        if len(times) >= 3 and (time.time() - t_all > budget_s or len(times) >= 12):
            break
    steady = times[1:]
    return {"value": round(1.0 / float(np.median(steady)), 4), "unit": "steps/s", "cores": oracle.num_threads(), "kind": "port",
            "sample": f"{len(steady)} full 800x800 sim+render steps of the C++/OpenMP oracle (median; first step discarded; init {init_s:.1f}s untimed)"}


def collect_samples(m, rays_o, rays_d, kw):
    """All (xyz, dir) samples of one frame, gathered with the op-by-op loop (same kernels as the fused path)."""
    xs, ds = [], []
    orig = m.forward

    def tap(x, d):
        s, c = orig(x, d)
        xs.append(x)
        ds.append(d)
        return s, c
    m.forward = tap
    try:
        m.rund_cuda_ops(rays_o, rays_d, **kw)
    finally:
        m.forward = orig
    x, d = torch.cat(xs), torch.cat(ds)
    keep = d.abs().sum(-1) > 0  # the op-level path also evaluates padded slots; real samples carry a non-zero direction
    return x[keep].contiguous(), d[keep].contiguous()


def kernel_report(h, opt, dev):
    """Per-kernel figures on one real frame (HIP events on the launch stream) + the march kernel's work counters."""
    from pienerf_amd._lib import check, lib, ptr, stream_ptr
    m = h.model
    out = h.step(simulate=False, collect_stats=True)   # also makes sure the frame workspace exists
    st = dict(m.last_stats)
    # (1) work counters of the march kernel (separate pass: the counters add atomics)
    m.march_counters(1)
    h.step(simulate=False)
    cnt = m.march_counters(0, read=True)
    # (2) per-trip launch durations, events around every march / network launch of a blocking render
    m.march_counters(2)
    for _ in range(3):
        h.step(simulate=False)
    reps = []
    for _ in range(5):
        h.step(simulate=False)
        reps.append(m.trip_times())
    m.march_counters(0)
    march_ms = np.median(np.array([r[0] for r in reps]), axis=0)
    net_ms = np.median(np.array([r[1] for r in reps]), axis=0)
    real = st["trips"]
    march_total, march_launch = float(march_ms[:real].sum()), float(march_ms[:real].mean())
    march_bytes = (cnt["iterations"] * MARCH_BYTES["iteration"] + cnt["candidates"] * MARCH_BYTES["candidate"] + cnt["warps"] * MARCH_BYTES["warp"]
                   + cnt["samples"] * MARCH_BYTES["sample"] + opt["W"] * opt["H"] * MARCH_BYTES["ray_trip"])  # trip 0 touches every ray once
    march_gbs = march_bytes / (march_total * 1e-3) / 1e9
    # (3) stand-alone network / hash-grid kernels on the frame's real sample set
    xyz, dirs = collect_samples(m, out["rays_o"], out["rays_d"], h.render_kwargs())
    B = xyz.shape[0]
    u = ((xyz + m.bound) / (2 * m.bound)).contiguous()
    enc = m.encoder
    feats = torch.empty(B * 32, device=dev)
    S = float(np.float32(np.log2(enc.per_level_scale)))

    def grid_launch(bl_major):
        check(lib().pn_grid_encode_forward(ptr(u), ptr(enc.embeddings), enc._offsets_host.data_ptr(), ptr(feats), B, 3, 2, 16, S, 16, None, 0, 0, 0,
                                           bl_major, stream_ptr()), "grid")
    t_grid = cuda_time_ms(lambda: grid_launch(0))      # [L,B,C]: the reference kernel's own output layout (gridencoder.cu:105)
    t_grid_bl = cuda_time_ms(lambda: grid_launch(1))   # [B,L*C] written directly (what grid.py:57 obtains with an extra permute pass)
    t_net = cuda_time_ms(lambda: m(xyz, dirs))
    t_sim = cuda_time_ms(lambda: h.sim.stepforward(), iters=10)
    t_frame = cuda_time_ms(lambda: h.step(simulate=False), iters=10)
    grid_gbs = HASH_BYTES_PER_SAMPLE * B / (t_grid * 1e-3) / 1e9
    net_gbs = FUSED_BYTES_PER_SAMPLE * B / (t_net * 1e-3) / 1e9
    net_tf = MLP_FLOP_PER_SAMPLE * B / (t_net * 1e-3) / 1e12
    # HBM traffic per launch from the committed PMC passes (profiles/pmc_traffic.json: FETCH_SIZE + WRITE_SIZE, KB, separate rocprofv3 runs);
    # the profile averaged over all enqueued trips, the empty ones move ~nothing, so scale to the real launches like `achieved`
    traffic = {}
    pmc_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_traffic.json")
    if os.path.exists(pmc_path):
        with open(pmc_path) as f:
            pmc = json.load(f)
        for kname in ("k_nerf_forward", "k_march", "k_march_tail", "k_march_skip"):
            pm = pmc.get(kname)
            if pm:  # bytes per FRAME of this kernel / real trips per frame
                frames = pm["dispatches"] / float(pm.get("enqueued_trips_per_frame", 8))
                traffic[kname] = int((pm["fetch_kb_per_launch"] + pm["write_kb_per_launch"]) * 1024 * pm["dispatches"] / (frames * pm["real_trips_per_frame"]))
        if all(k in traffic for k in ("k_march", "k_march_tail", "k_march_skip")):  # one launch group = one trip; the skip pre-pass runs once per frame
            traffic["march_group"] = traffic["k_march"] + traffic["k_march_tail"] + traffic["k_march_skip"] // real
    # dominant kernel = the fused network kernel (largest single-kernel share of the step's GPU time, profiles/README.md): its launches in
    # the render loop, HIP events on the launch stream around each of them (pn_frame_trip_times)
    net_loop_ms = float(net_ms[:real].sum())
    net_loop_gbs = FUSED_BYTES_PER_SAMPLE * st["samples"] / (net_loop_ms * 1e-3) / 1e9
    net_loop_tf = MLP_FLOP_PER_SAMPLE * st["samples"] / (net_loop_ms * 1e-3) / 1e12
    network = {
        "kernel": "k_nerf_forward<2,4> (hash-grid gather + SH + 5-layer MLP fused; dense layers as three-way bf16-split MFMA at fp32 accuracy), "
                  "launches inside the render loop",
        # cutting the matrix time by 2.7x (f32-input MFMA -> bf16 split) left the stand-alone kernel time unchanged, the
        # gather-only variant of the kernel takes 69 % of its time, the MLP-only variant 55 %: the bound is the gather path
        "bound": "hbm", "achieved": round(net_loop_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(net_loop_gbs / HBM_PEAK_GBS, 4),
        "bytes_per_sample": FUSED_BYTES_PER_SAMPLE, "algorithmic_bytes_per_launch": int(FUSED_BYTES_PER_SAMPLE * st["samples"] / real),
        "traffic": traffic.get("k_nerf_forward"),
        "traffic_note": "HBM bytes per launch, FETCH_SIZE+WRITE_SIZE PMC passes (profiles/pmc_traffic.json); below the algorithmic bytes because the "
                        "dense levels and part of the hashed tables are served by L2 / Infinity Cache",
        "launch_ms": round(net_loop_ms / real, 4), "launches_per_frame": real, "ms_per_frame": round(net_loop_ms, 4),
        "launch_ms_incl_empty_trips": round(float(net_ms.mean()), 4), "launches_enqueued_per_frame": int(len(net_ms)),
        "samples_per_frame": st["samples"],
        "mfma_view": {"flop_per_sample_fp32_equivalent": MLP_FLOP_PER_SAMPLE, "bf16_mfma_flop_per_sample_issued": 6 * 20 * 32768 // 32,
                      "fp32_equivalent_TFLOPs": round(net_loop_tf, 2), "frac_of_f32_mfma_peak": round(net_loop_tf / F32_MFMA_PEAK_TF, 4)},
        "all_samples_one_launch": {"launch_ms": round(t_net, 4), "achieved_GBps": round(net_gbs, 1), "frac_of_hbm_peak": round(net_gbs / HBM_PEAK_GBS, 4),
                                   "fp32_equivalent_TFLOPs": round(net_tf, 2)},
        "note": "achieved = 1 068 algorithmic bytes per sample (16 levels x 8 corners x 8 B gathered + 4 B slot id + 24 B xyz/dir in + 16 B sigma/rgb "
                "out) x samples of the launch / HIP-event time of the launch; peak = 8 TB/s HBM3E (MI355X_MICROARCH.md)",
    }
    # dominant kernel = the ray march (k_march 19.4 % + k_march_tail 18.5 % + k_march_skip 4.4 % of the step's GPU time in
    # profiles/r01_final_kernel_stats.csv; the network kernel is 15.8 %): one algorithm in two passes per trip plus the trip-0 pre-pass,
    # timed as one launch group by HIP events on the launch stream (pn_frame_trip_times)
    roofline = {
        "kernel": "ray march + inverse-GMLS warp: k_march<3,false> + k_march_tail<3,false> per loop trip (+ k_march_skip on trip 0), one launch group",
        "bound": "hbm", "achieved": round(march_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(march_gbs / HBM_PEAK_GBS, 4),
        "traffic": traffic.get("march_group"),
        "traffic_note": "HBM bytes per launch group, FETCH_SIZE+WRITE_SIZE PMC passes (profiles/pmc_traffic.json): far below the algorithmic bytes "
                        "because the candidate lists and IP records (~1 MB per frame) are re-read from L2",
        "launch_ms": round(march_launch, 4), "launches_per_frame": real, "ms_per_frame": round(march_total, 4),
        "launch_ms_incl_empty_trips": round(float(march_ms.mean()), 4), "launches_enqueued_per_frame": int(len(march_ms)),
        "units_per_frame": cnt, "algorithmic_bytes_per_frame": int(march_bytes), "algorithmic_bytes_per_launch": int(march_bytes / real),
        "bytes_per_unit": MARCH_BYTES,
        "note": "achieved = algorithmic bytes (8 B per marched ray point, 16 B per candidate-list entry, 64 B per warped IP record head, 36 B per "
                "emitted sample, 40 B of ray state per ray and trip) / HIP-event time of the launch group.  The kernel is a divergent pointer chase "
                "over cache-resident tables: latency-bound, not a streaming kernel, so the HBM roofline is an upper bound it cannot approach "
                "(DESIGN.md 4.1)",
    }
    extra = {
        "network": network,
        "hash_lookup": {"kernel": "k_grid_encode<2> (stand-alone hash-grid lookup, output [L,B,C] like the reference kernel)",
                        "achieved_GBps": round(grid_gbs, 1), "frac_of_hbm_peak": round(grid_gbs / HBM_PEAK_GBS, 4), "launch_ms": round(t_grid, 4),
                        "bytes_per_sample": HASH_BYTES_PER_SAMPLE, "launch_ms_direct_BLC_output": round(t_grid_bl, 4)},
        "breakdown_ms": {"stepforward_alone": round(t_sim, 4), "render_frame_eager": round(t_frame, 4),
                         "march_per_trip": [round(float(v), 4) for v in march_ms[:real]], "network_per_trip": [round(float(v), 4) for v in net_ms[:real]],
                         "local_global_iters_per_s": round(opt["sim_iters"] / (t_sim * 1e-3), 1)},
    }
    return st, roofline, extra


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--eager", action="store_true", help="launch every kernel from the host instead of replaying the captured HIP graph")
    ap.add_argument("--trips", type=int, default=8, help="render-loop trips baked into the captured graph")
    ap.add_argument("--lanes", type=int, default=3,
                    help="renders in flight on the GPU (1 = strictly one frame after the other; 3 render streams + the simulator stream = the 4 "
                         "compute pipes of an XCD, more streams only time-slice)")
    ap.add_argument("--single-graph", action="store_true", help="whole step as ONE captured graph (sim on a forked stream), one frame at a time")
    ap.add_argument("--cpu-budget", type=float, default=20.0)
    ap.add_argument("--dedicated-sim", choices=("auto", "on", "off"), default="auto",
                    help="N > 1: the sim owner only simulates and broadcasts, the other ranks render (auto: from 3 ranks on, frames.dedicated_sim_default)")
    ap.add_argument("--force", type=float, nargs=3, default=None, metavar=("FX", "FY", "FZ"),
                    help="constant update_force on the middle integration point (SURVEY 8d, config 2 second pass); default: gravity only")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # "nccl" IS RCCL on ROCm.  PN_DIST_BACKEND=gloo lets the N > 1 code path be exercised on a one-GPU box (ranks share cuda:0).
        backend = os.environ.get("PN_DIST_BACKEND", "nccl")
        dev_index = local_rank % torch.cuda.device_count()
        torch.cuda.set_device(dev_index)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    else:
        dev_index = 0
        torch.cuda.set_device(0)
    dev = torch.device("cuda", dev_index)

    from pienerf_amd import scene
    from pienerf_amd.harness import SimRenderHarness

    opt = scene.default_opt()  # chair demo options (README.md:123): 800x800, bound 1, dt_gamma 0, num_seek_IP 3, max_iter_num 1, sim_dx 0.05, iters 10
    cloud = scene.make_chair_points(hgs=opt["hash_grid_size"])
    ckpt = scene.make_checkpoint(bound=opt["bound"], seed=0)
    h = SimRenderHarness(opt, cloud=cloud, ckpt=ckpt, device=dev)
    if args.force is not None and (world == 1 or rank == 0):  # Simulator.update_force (solver.py:578-588): the dragged-point load of the GUI
        h.sim.update_force(h.sim.n_IP // 2, np.asarray(args.force, dtype=np.float64))

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    if world == 1:
        if args.eager:
            run_steps = lambda n: [h.step() for _ in range(n)]
        elif not args.single_graph:
            # `lanes` renders in flight on their own streams, the simulator running ahead on dof snapshots (harness.capture_pipelined);
            # lanes = 1 is one render at a time with the next substep overlapping it
            h.capture_pipelined(lanes=args.lanes, n_trips=args.trips)
            run_steps = lambda n: [h.step_pipelined() for _ in range(n)]
        else:
            # the whole step (get_rays, get_IP_info, stepforward on a forked stream, render prologue + loop trips + epilogue) is one
            # captured HIP graph; each replay re-checks that the previous frame left no ray alive
            h.capture(n_trips=args.trips)
            run_steps = lambda n: [h.step_graph() for _ in range(n)]
    else:
        # frame-parallel (harness.capture_frame_parallel): rank 0 owns the simulator, runs it ahead on dof snapshots and
        # broadcasts each snapshot (<= 82 KB) over RCCL on a communication stream; frame f is rendered by rank f % world on one of its
        # `lanes` render streams.  `n` steps per rank = n * world frames in total.
        from pienerf_amd.frames import broadcast_tensors
        m = h.model
        broadcast_tensors([m.encoder.embeddings.data, m.density_bitfield] + [l.weight.data for l in list(m.sigma_net) + list(m.color_net)], src=0)
        # ROCm time-slices badly once more than 4 hardware queues are busy (launch structure): with the simulator stream
        # and the RCCL communication stream that leaves 2 render lanes per rank; a rank renders only every world-th frame anyway
        args.lanes = min(args.lanes, 2)
        dedicated = {"auto": None, "on": True, "off": False}[args.dedicated_sim]
        h.capture_frame_parallel(lanes=args.lanes, n_trips=args.trips, dedicated_sim=dedicated)

        def run_steps(n):
            for _ in range(n * world):
                h.step_frame_parallel()

    with torch.no_grad():
        run_steps(args.warmup)
        barrier()
        t0 = time.perf_counter()
        run_steps(args.steps)
        barrier()
        elapsed = time.perf_counter() - t0
        if world > 1 or not args.eager:  # the last replayed frame(s) must be complete too
            if world > 1 or not args.single_graph:
                h.drain_pipeline()
            else:
                h._check_previous_graph_frame()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        with torch.no_grad():
            st, roofline, extra = kernel_report(h, opt, dev)
        res = {
            "metric": "sim+render steps/s @800x800 chair", "value": round(args.steps * world / elapsed, 3), "unit": "steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32 render / f64 sim", "data": "synthetic",
            "config": {"workload": "configs[1]: synthetic chair 800x800, sim_dx=0.05, sim_iters=10, num_seek_IP=3, max_iter_num=1, fp32, "
                                   "1 sim+render step per frame" + (f", constant force {args.force} on IP {h.sim.n_IP // 2}" if args.force is not None else ", gravity only"),
                       "rays": opt["W"] * opt["H"], "n_IP": h.sim.n_IP, "n_kernels": h.sim.n_k,
                       "samples_per_frame": st["samples"], "trips_per_frame": st["trips"],
                       "launch": "eager" if (args.eager and world == 1) else (f"one hip graph per step, {args.trips} trips" if args.single_graph else
                                                                                   f"hip graphs, {args.trips} trips, {args.lanes} render(s) in flight, simulator running ahead"),
                       "parallelism": (f"frame-parallel x{world}, DOF broadcast over RCCL, " + ("rank 0 simulates only, frames round-robin over the other ranks"
                                                                                                if h._pipe["dedicated"] else "frames round-robin over all ranks"))
                       if world > 1 else "single GPU"},
            "roofline": roofline,
        }
        res.update(extra)
        if not args.no_cpu_baseline and world == 1:  # a reported baseline, timed on rank 0 at N = 1 only
            res["cpu_baseline"] = cpu_baseline(opt, cloud, ckpt, args.cpu_budget)
        print(json.dumps(res))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
