#!/usr/bin/env python
"""sim+render steps/s of the PIE-NeRF simulate-and-render step on MI355X; one JSON line on rank 0.

    python bench.py [--gpus N --steps K --warmup W] [--config chair|stress|trex]     (N > 1: launched by torch.distributed.run, one rank per GPU)

A step = one GUI-frame equivalent of the reference (nerf/gui.py:588-603 / nerf/trainer.py:300-318, 531-602):
get_rays(pose) -> get_IP_info -> stepforward(sim_iters) -> render_deformed -> image / depth / depth_0 copied to host memory.
Inputs are resident in HBM when the timed region starts; `value` INCLUDES the device-to-host copy of the three outputs (12.8 MB per 800x800
frame into pinned memory on a copy stream: it is part of the reference's step); the device-resident rate is reported beside it.

  --config chair  (default)  BASELINE.json configs[1]: synthetic chair 800x800, sim_dx 0.05, 10 local/global iterations, num_seek_IP 3, max_iter_num 1, fp32.
                             N = 1: frames.FramePipeline — `--lanes` render streams x `--depth` workspaces, simulator running ahead on dof snapshots,
                             everything replayed from HIP graphs.  N > 1 (configs[3]): the same pipeline frame-parallel over the ranks, rank 0 simulates
                             and broadcasts each dof snapshot (<= 82 KB) over RCCL, frames round-robin; K steps per rank = K*N frames (weak scaling).
  --config stress            configs[4]: sub_res 180 point cloud, max_iter_num 5, num_seek_IP 3, the 800x800 frame in ray batches of 4096, network under
                             autocast (fp16 hash tables + fp16 MFMA layers).  harness.capture_staged.
  --config trex              configs[2]: 1008x756, bound 2 (two cascades), --cut, dt_gamma 1/128, max_steps 300, T_thresh 5e-2, num_seek_IP 1, a static
                             background in 2 % of the density-grid blocks.
After the timed region rank 0 measures (N = 1 only, each a fraction of a second): the device-resident rate, the one-frame-at-a-time latency,
the dominant kernel's launch durations (HIP events on the launch stream: blocking render AND inside the pipelined graphs), the stand-alone hash-grid
and network kernels, and the CPU oracle on the host cores.
"""
import argparse
import gc
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
HASH_BYTES_PER_SAMPLE = 1164        # 16 levels x 8 corners x 8 B gathered + 12 B position in + 128 B features out
FUSED_BYTES_PER_SAMPLE = 1068       # fused network kernel, fp32 tables: 1024 B gathered + 4 B slot id + 24 B xyz/dir in + 16 B sigma/rgb out
FUSED_BYTES_PER_SAMPLE_FP16 = 556   # fp16 tables: 512 B gathered + 44 B
MLP_FLOP_PER_SAMPLE = 18688
MARCH_BYTES = dict(iteration=8, candidate=16, warp=64, sample=32 + 4, ray_trip=40)  # cell range / list entry / record head / outputs / ray state
HBM_PEAK_GBS = 8000.0               # MI355X_MICROARCH.md: HBM3E 8 TB/s
F32_MFMA_PEAK_TF = 157.3            # dense fp32-input MFMA peak
F16_MFMA_PEAK_TF = 2500.0           # dense fp16 / bf16 MFMA peak


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


def make_config(name, sigma_gain=1.0):
    """(opt, cloud, ckpt, pose, force, description) of a BASELINE.json configuration on the synthetic assets."""
    from pienerf_amd import scene
    if name == "chair":
        opt = scene.default_opt()  # README.md:123
        if os.environ.get("PN_PROBE_SIM_ITERS"):   # diagnosis only (is the pipeline bound by the simulator's chain of launches?): NOT the benchmark's configuration
            opt["sim_iters"] = int(os.environ["PN_PROBE_SIM_ITERS"])
        cloud = scene.make_chair_points(hgs=opt["hash_grid_size"])
        ckpt = scene.make_checkpoint(bound=opt["bound"], seed=0, sigma_target=60.0 * sigma_gain)
        return opt, cloud, ckpt, scene.orbit_pose(opt["radius"]), None, ("configs[1]: synthetic chair 800x800, sim_dx=0.05, sim_iters=10, num_seek_IP=3, max_iter_num=1, "
                                                                         "fp32, 1 sim+render step per frame incl. D2H of image/depth/depth_0"
                                                                         + (f" [PN_PROBE_SIM_ITERS={opt['sim_iters']}: a diagnosis run, not the benchmark]" if os.environ.get("PN_PROBE_SIM_ITERS") else ""))
    if name == "stress":
        opt = scene.stress_opt()
        cloud = scene.make_chair_points(sub_res=opt["sub_res"], hgs=opt["hash_grid_size"])
        ckpt = scene.make_checkpoint(bound=opt["bound"], seed=0, sigma_target=60.0 * sigma_gain)
        return opt, cloud, ckpt, scene.orbit_pose(opt["radius"]), np.array([400.0, -150.0, 250.0]), (
            "configs[4] stress: sub_res=180 point cloud (268 k points), 800x800 frame in ray batches of 4096, max_iter_num=5, num_seek_IP=3, fp16 hash "
            "tables + fp16 MFMA MLP (autocast), sim_dx=0.05, sim_iters=10, incl. D2H")
    if name == "trex":
        opt = scene.trex_opt(radius=4.5)  # README.md:134
        cloud = scene.make_chair_points(hgs=opt["hash_grid_size"], bound=opt["bound"])
        ckpt = scene.make_checkpoint(bound=2.0, seed=3, sigma_target=60.0 * sigma_gain)
        blobs = np.repeat(np.random.default_rng(5).random(len(ckpt["density_bitfield"]) // 64) < 0.02, 64)  # static background: 2 % of the 8^3-voxel blocks
        ckpt["density_bitfield"] = ckpt["density_bitfield"] | np.where(blobs, 0xFF, 0).astype(np.uint8)
        return opt, cloud, ckpt, scene.orbit_pose(4.5, 25.0, -10.0), np.array([250.0, 120.0, -180.0]), (
            "configs[2] trex option set: 1008x756, bound 2 (2 cascades), --cut, dt_gamma 1/128, max_steps 300, T_thresh 5e-2, num_seek_IP 1, static "
            "background in 2 % of the density blocks, synthetic assets, incl. D2H")
    raise ValueError(name)


def cpu_baseline(opt, cloud, ckpt, pose, force, budget_s=20.0):
    """The CPU oracle ("port") on this box's host cores: sim steps + full-size renders for ~budget_s seconds."""
    import oracle
    from oracle.sim_init import OracleSimulator
    from pienerf_amd import scene
    t0 = time.time()
    ref = OracleSimulator(dt=opt["sim_dt"], iters=opt["sim_iters"], bbox=torch.tensor([2.0 * opt["bound"]] * 3), dx=opt["sim_dx"],
                          stiff=opt["sim_stiff"], base=torch.tensor([-opt["bound"]] * 3))
    ref.InitializeFromArrays(cloud["pos"], cloud["mass"], cloud["mu"], cloud["lam"], cloud["pin"])
    if force is not None:
        ref.update_force(ref.n_IP // 2, force)
    init_s = time.time() - t0
    p_ori, _, _ = ref.get_IP_info()
    intr = scene.orbit_intrinsics(opt["W"], opt["H"], opt["fovy"])
    times = []
    t_all = time.time()
    import contextlib
    ctx = oracle.half_precision() if opt.get("fp16") else contextlib.nullcontext()
    with ctx:
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
            "sample": f"{len(steady)} full {opt['W']}x{opt['H']} sim+render steps of the C++/OpenMP oracle (median; first step discarded; init {init_s:.1f}s untimed)"}


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


TRAFFIC_INSTANCES = {}


def load_traffic(real_trips, config="chair"):
    """HBM-side bytes per launch from the committed PMC passes of THIS workload — only when they were taken on THIS code (profiles/pmc_traffic*.json
    are stamped with pienerf_amd.build.source_hash by tools/pmc_traffic.py); a stale file gives null, never a number that belongs to other kernels."""
    from pienerf_amd.build import source_hash
    name = "pmc_traffic.json" if config == "chair" else f"pmc_traffic_{config}.json"
    path = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(path):
        return {}, f"profiles/{name} not present"
    with open(path) as f:
        pmc = json.load(f)
    if pmc.get("lib_hash") != source_hash():
        return {}, f"profiles/{name} was measured on other kernel sources (stamp {str(pmc.get('lib_hash'))[:10]}, tree {source_hash()[:10]}): not reported"
    out = {}
    for k, v in pmc["kernels"].items():
        out[k.split("<")[0]] = out.get(k.split("<")[0], 0) + v["fetch_bytes_per_frame"] + v["write_bytes_per_frame"]
        if "<" in k and v["fetch_bytes_per_frame"] + v["write_bytes_per_frame"] > 0:
            TRAFFIC_INSTANCES.setdefault(k.split("<")[0], k)   # the template instance the passes saw (checked against the timed launch's)
    per_launch = {k: int(v / max(real_trips, 1)) for k, v in out.items()}
    if all(k in out for k in ("k_march", "k_march_tail", "k_march_skip")):
        per_launch["march_group"] = int((out["k_march"] + out["k_march_tail"] + out["k_march_skip"]) / max(real_trips, 1))
        per_launch["first_trip_march"] = int(out["k_march"] + out["k_march_tail"] + out["k_march_skip"])   # with the later trips fused these only run on trip 0
    if "k_trips_fused" in out:
        per_launch["k_trips_fused"] = int(out["k_trips_fused"])   # one launch per frame
    return per_launch, "FETCH_SIZE x 2 + WRITE_SIZE, separate rocprofv3 PMC passes on the same kernels (tools/pmc_traffic.py); per launch"


def march_bytes(cnt, ray_trips):
    """Algorithmic bytes of marching work (SURVEY 8d / DESIGN 4): cell range per visited point, list entry per candidate scanned, record head per inverse
    warp, outputs per emitted sample, ray state per ray and trip."""
    return (cnt["iterations"] * MARCH_BYTES["iteration"] + cnt["candidates"] * MARCH_BYTES["candidate"] + cnt["warps"] * MARCH_BYTES["warp"]
            + cnt["samples"] * MARCH_BYTES["sample"] + ray_trips * MARCH_BYTES["ray_trip"])


def kernel_report(h, opt, dev, form_kw=None, graph_ms=None):
    """Per-kernel figures of one real frame in the FORM that produced `value` (form_kw: the render options the pipeline picked — throughput form of the
    first trip, fused launch from which trip on, its grid): HIP events on the launch stream around every launch group of a blocking render ("alone"),
    graph_ms = the same brackets as time stamps inside the pipelined graphs ("in the pipeline": the durations behind `value`), the march's work counters
    split into first trip / fused launch, the fused launch's phase clocks, the stand-alone network and hash-grid kernels."""
    from pienerf_amd._lib import check, lib, ptr, stream_ptr
    m = h.model
    fp16 = bool(opt.get("fp16"))
    x_form = (not fp16) and lib().pn_net_form(m._net_handle()) == 2   # how the fp32 network's dense layers run for these weights (include/pienerf_hip.h)
    N = opt["W"] * opt["H"]
    for k, v in (form_kw or {}).items():   # render_kwargs() hands the option set to the renderer by name
        if v is not None:
            h.opt[k] = v
    for _ in range(20):   # a deformed state like the ones the timed region rendered, not the rest pose
        h.step()
    h.synchronize()
    out = h.step(simulate=False, collect_stats=True)   # also makes sure the frame workspace exists
    st = dict(m.last_stats)
    st["hit_rays"] = int((~torch.isnan(out["depth"])).sum())  # rays that meet the bounding box of the deformed IPs (miss: near = far = FLT_MAX -> NaN depth)
    recs = m.trip_records(max_trips=140)
    fc0 = m.fused_clocks()
    ff = fc0["first_trip"]                              # trip at which the fused launch took over (-1: trip-by-trip launches only)
    folded = ff >= 1 and fc0.get("mode") == 2           # ... with the first trip's network / composite / compaction inside it (pn_render_opts.fused_fold)
    real = st["trips"]
    # (1) work counters of the march (separate passes: the counters add atomics): the whole frame, and its first trip alone
    m.march_counters(1)
    h.step(simulate=False)
    cnt = m.march_counters(0, read=True)
    cnt0 = None
    if ff >= 1:
        m.march_counters(1)
        with h._amp():
            m.render_deformed(out["rays_o"], out["rays_d"], staged=True, bg_color=None, perturb=False,
                              **dict(h.render_kwargs(), async_trips=ff, fused_from=-1))   # the trips in front of the fused launch, as per-trip launches
        torch.cuda.synchronize()
        cnt0 = m.march_counters(0, read=True)
        h.step(simulate=False)   # leave a finished frame on the workspace
    # (2) launch durations, events around every launch group of a blocking render
    m.march_counters(2)
    for _ in range(3):
        h.step(simulate=False)
    reps = []
    for _ in range(5):
        h.step(simulate=False)
        reps.append(m.trip_times())
    m.march_counters(0)
    n_timed = min(len(r[0]) for r in reps)
    march_ms = np.median(np.array([r[0][:n_timed] for r in reps]), axis=0)
    net_ms = np.median(np.array([r[1][:n_timed] for r in reps]), axis=0)
    # (3) phase clocks of the fused launch
    phases = None
    if ff >= 0:
        m.march_counters(4)
        m.fused_clocks(reset=True)
        for _ in range(5):
            h.step(simulate=False)
        c = m.fused_clocks()
        m.march_counters(0)
        keys = ("refill", "march", "windows", "network", "composite", "a_march", "a_windows", "a_network", "a_composite", "a_barrier")
        tot = float(sum(c[k] for k in keys)) or 1.0
        phases = {"share_of_wave_time": {k: round(c[k] / tot, 4) for k in keys if c[k]}, "waves": int(c["waves"] / 5), "wave_rounds_per_frame": int(c["wave_rounds"] / 5),
                  "mean_wave_lifetime_us": round(c["lifetime_ticks"] / max(c["waves"], 1) / 100.0, 1), "longest_wave_lifetime_us": round(c["max_lifetime_ticks"] / 100.0, 1),
                  "note": "shader-clock cycles per phase summed over the launch's waves (march_counters(4): a drain of the memory counters at every phase "
                          "boundary — a measurement build of the same launch, never the timed one)"}
    # (4) stand-alone network / hash-grid kernels on the frame's real sample set
    with h._amp():
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
    with torch.autocast("cuda", dtype=torch.float16):
        t_net_h = cuda_time_ms(lambda: m(xyz, dirs))
    t_sim = cuda_time_ms(lambda: h.sim.stepforward(), iters=10)
    # the same substep as ONE persistent kernel (csrc/pn_sim.hip: k_substep_coop) — the form a GPU that only simulates uses (the dedicated owner of
    # a frame-parallel job); never beside renders, so it is measured here, alone, and switched off again
    t_sim_coop = None
    alone = not (torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1)
    if alone and h.sim.enable_persistent():  # (in a multi-rank job other ranks may still hold this GPU: one-GPU dry runs)
        t_sim_coop = cuda_time_ms(lambda: h.sim.stepforward(), iters=10)
        if h.sim._coop is None or h.sim.persistent_timed_out():
            t_sim_coop = None
    h.sim.persistent = False
    t_frame = cuda_time_ms(lambda: h.step(simulate=False), iters=10)
    grid_gbs = HASH_BYTES_PER_SAMPLE * B / (t_grid * 1e-3) / 1e9
    bps = FUSED_BYTES_PER_SAMPLE_FP16 if fp16 else FUSED_BYTES_PER_SAMPLE
    t_used = t_net_h if fp16 else t_net
    if opt.get("_config_name", "chair") in ("chair", "stress", "trex"):  # PMC passes are taken per workload (tools/run_frames.py --config)
        traffic, traffic_note = load_traffic(real, opt.get("_config_name", "chair"))
    else:
        traffic, traffic_note = {}, "no PMC passes for this variant of the workload (sigma gain != 1): not reported"

    def gbs(nbytes, ms):
        return nbytes / (ms * 1e-3) / 1e9 if ms and ms > 0 else 0.0
    g_march, g_net = (None, None) if graph_ms is None else graph_ms
    per_trip = max(ff, 0) if ff >= 0 else real    # trips that ran as per-trip launches
    kname = "k_nerf_forward_h" if fp16 else "k_nerf_forward"
    if ff >= 0:
        # ---- the frame = `ff` per-trip trips + ONE fused launch: the fused launch is the dominant kernel
        head = cnt0 if cnt0 is not None else {k: 0 for k in cnt}
        cnt_f = {k: cnt[k] - head[k] for k in cnt}
        ray_trips_f = int(sum(r[0] for r in recs[max(ff, 1):])) if ff >= 1 else int(sum(r[0] for r in recs[1:])) + N
        bytes_f_march = march_bytes(cnt_f, ray_trips_f)
        bytes_f_net = (cnt_f["samples"] + (head["samples"] if folded else 0)) * bps   # (folded: the first trip's samples go through the launch's network tiles too)
        t_f_alone = float(march_ms[ff]) if ff < n_timed else None
        t_f_graph = float(g_march[ff]) if g_march is not None and ff < len(g_march) else None
        t_f = t_f_graph or t_f_alone
        fused_gbs = gbs(bytes_f_march + bytes_f_net, t_f)
        head_march_alone = float(march_ms[:ff].sum()) if ff >= 1 else None
        head_march_graph = float(np.sum(g_march[:ff])) if (g_march is not None and ff >= 1) else None
        head_bytes = march_bytes(head, N + int(sum(r[0] for r in recs[1:ff]))) if ff >= 1 else 0
        net_share = phases["share_of_wave_time"].get("network", 0.0) + phases["share_of_wave_time"].get("a_network", 0.0) if phases else 0.0
        # the template instance rocprofv3 names: k_trips_fused<K, MULTI, NF, MODE> (NF: 0 bf16 pieces, 1 fp16 network, 2 fp16 hi / lo; MODE: 0 later trips, 1 whole
        # frame, 2 first trip's network / composite / compaction folded in) — the PMC stamp must name the same one
        from pienerf_amd import _lib
        nf = 1 if fp16 else int(_lib.lib().pn_net_form(h.model._net_handle()))
        fused_instance = f"k_trips_fused<{opt['num_seek_IP']}, {'true' if opt['max_iter_num'] > 1 else 'false'}, {nf}, {int(fc0.get('mode', 0))}>"
        if traffic.get("k_trips_fused") is not None and TRAFFIC_INSTANCES.get("k_trips_fused") not in (None, fused_instance):
            traffic_note = (f"profiles/pmc_traffic*.json holds {TRAFFIC_INSTANCES.get('k_trips_fused')}, the timed launch is {fused_instance}: not reported "
                            "(tools/gpu_run.sh traffic runs the passes on the launch set the bench times)")
            traffic = dict(traffic, k_trips_fused=None)
        roofline = {
            "kernel": f"{fused_instance} — "
                      f"every loop trip from trip {ff} on as ONE persistent launch: per ray { '{' } march 8 samples + inverse-GMLS warp; hash grid + SH + MLP on MFMA; composite { '}' } "
                      "until the ray dies (csrc/pn_trips_fused.h); the largest kernel of the mode that produced `value`",
            "bound": "hbm", "achieved": round(fused_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(fused_gbs / HBM_PEAK_GBS, 4),
            "traffic": traffic.get("k_trips_fused"), "traffic_note": traffic_note,
            "launch_ms": round(t_f, 4) if t_f else None, "launches_per_frame": 1,
            "measured_in": ("time stamps inside the pipelined render graphs around the launch (the mode that produced `value`: "
                            "other frames' kernels, the simulator and the frame copies share the GPU)" if t_f_graph else
                            "blocking single-frame render, HIP events on the launch stream around the launch"),
            "launch_ms_alone": round(t_f_alone, 4) if t_f_alone else None, "frac_alone": round(gbs(bytes_f_march + bytes_f_net, t_f_alone) / HBM_PEAK_GBS, 4) if t_f_alone else None,
            "workgroups": int(h.opt.get("fused_grid") or 0) or "one per CU",
            "first_trip_folded_in": bool(folded),
            "algorithmic_bytes_per_launch": int(bytes_f_march + bytes_f_net),
            "algorithmic_bytes": {"march": int(bytes_f_march), "network": int(bytes_f_net), "bytes_per_unit": dict(MARCH_BYTES, network_sample=bps)},
            "units_per_launch": dict(cnt_f, ray_trips=ray_trips_f),
            "phases": phases,
            "first_trips_march": None if ff < 1 else {
                "kernel": "k_march_skip + k_march (one lane per ray) + k_march_tail + k_list_pack: the march of the trip(s) in front of the fused launch — round 3's dominant kernel group",
                "ms_per_frame_alone": round(head_march_alone, 4), "ms_per_frame_in_pipeline": round(head_march_graph, 4) if head_march_graph else None,
                "algorithmic_bytes_per_frame": int(head_bytes), "achieved_GBps_alone": round(gbs(head_bytes, head_march_alone), 1),
                "frac_alone": round(gbs(head_bytes, head_march_alone) / HBM_PEAK_GBS, 4),
                "frac_in_pipeline": round(gbs(head_bytes, head_march_graph) / HBM_PEAK_GBS, 4) if head_march_graph else None,
                "traffic": traffic.get("first_trip_march"), "units": head},
            "note": "achieved = algorithmic bytes of the launch (march: 8 B per visited ray point, 16 B per candidate-list entry, 64 B per warped IP record head, 36 B per "
                    "emitted sample, 40 B of ray state per ray and trip; network: 1024 B of hash-table corners + 44 B per sample) / duration of the launch.  The tables "
                    "are cache-resident (`traffic`: HBM-side bytes from the PMC passes) and what bounds the launch is the per-CU gather path of the network phases "
                    "(distinct cache lines per wave instruction) and the dependent-issue chains of the march phases that share the SIMDs with them (`phases`), "
                    "not HBM bytes: the HBM roofline is the nominal ruler BASELINE asks for, an upper bound the launch cannot approach",
        }
        net_loop_ms = float(net_ms[:ff].sum()) + (t_f_alone or 0.0) * net_share
        net_samples_head = int(head["samples"])
    else:
        # ---- trip-by-trip launches only (ray batches, static): the march launch group per trip, as in rounds 1-3
        march_total = float(march_ms[:real].sum())
        mb = march_bytes(cnt, N)
        roofline = {
            "kernel": "ray march + inverse-GMLS warp: k_march + k_march_tail per loop trip (+ k_march_skip on trip 0), one launch group",
            "bound": "hbm", "achieved": round(gbs(mb, march_total), 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs(mb, march_total) / HBM_PEAK_GBS, 4),
            "traffic": traffic.get("march_group"), "traffic_note": traffic_note,
            "launch_ms": round(march_total / real, 4), "launches_per_frame": real, "ms_per_frame": round(march_total, 4),
            "ms_per_frame_in_pipeline": round(float(np.sum(g_march[:real])), 4) if g_march is not None else None,
            "measured_in": "blocking single-frame render, HIP events on the launch stream around each trip's march launches (pn_frame_trip_times)",
            "units_per_frame": cnt, "algorithmic_bytes_per_frame": int(mb), "algorithmic_bytes_per_launch": int(mb / real), "bytes_per_unit": MARCH_BYTES,
            "note": "achieved = algorithmic bytes / HIP-event time of the launch group; the tables are cache-resident and the kernels are bound by VALU issue, not bytes (DESIGN.md 4.1)",
        }
        net_loop_ms = float(net_ms[:real].sum())
        net_samples_head = st["samples"]
    net_loop_gbs = gbs(bps * st["samples"], net_loop_ms)
    net_loop_tf = MLP_FLOP_PER_SAMPLE * st["samples"] / (net_loop_ms * 1e-3) / 1e12 if net_loop_ms > 0 else 0.0
    network = {
        "kernel": ("k_nerf_forward_h<4,4> / the same tile inside k_trips_fused (fp16 hash tables + SH + 5-layer MLP fused; dense layers on v_mfma_f32_32x32x16_f16, half activations)" if fp16 else
                   "k_nerf_forward<2,4,X> / the same tile inside k_trips_fused (hash-grid gather + SH + 5-layer MLP fused; dense layers at fp32 accuracy: " +
                   ("every value as fp16 hi + lo pieces, 3 products per K chunk on v_mfma_f32_32x32x16_f16)" if x_form else "three bf16 pieces, 6 products per K chunk on v_mfma_f32_32x32x16_bf16)")),
        "dense_form": "fp16" if fp16 else ("fp16 hi/lo x3" if x_form else "bf16 x6"),
        "bound": "hbm", "achieved": round(net_loop_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(net_loop_gbs / HBM_PEAK_GBS, 4),
        "bytes_per_sample": bps, "traffic": traffic.get(kname), "traffic_note": traffic_note,
        "ms_per_frame": round(net_loop_ms, 4), "samples_per_frame": st["samples"],
        # the fused launch runs on `fused_grid` of the chip's CUs (two frames' launches side by side): per CU-time the network phases are measured against the
        # stand-alone kernel, which has all 256 CUs.  frac x 256 / workgroups, only when every network tile of the frame runs inside the fused launch
        "frac_per_cu_time": (round(net_loop_gbs / HBM_PEAK_GBS * 256.0 / float(h.opt["fused_grid"]), 4)
                             if (ff >= 0 and folded and h.opt.get("fused_grid") and per_trip <= 1) else None),
        "frac_per_cu_time_note": ("network.frac is bytes / (the launch's duration x the network phases' share of wave time) of a launch on fused_grid = "
                                  f"{h.opt.get('fused_grid')} of 256 CUs; a launch of that width doing nothing but network tiles at the stand-alone kernel's per-CU rate "
                                  "(all_samples_one_launch, all 256 CUs) would read frac_of_hbm_peak x fused_grid / 256 — that is the ceiling network.frac is to be held "
                                  "against, and frac_per_cu_time = frac x 256 / fused_grid is the figure comparable with the stand-alone kernel's"),
        "definition": ("network time of a frame = the network launches of the per-trip trips (HIP events) + the fused launch's duration x the share of its waves' time "
                       "spent in network phases (phase clocks), both of a blocking render; achieved = samples x bytes_per_sample / that" if ff >= 0 else
                       "network launches inside the render loop, HIP events"),
        "per_trip_launch_ms": [round(float(v), 4) for v in net_ms[:per_trip]], "samples_in_per_trip_launches": net_samples_head,
        "mfma_view": {"flop_per_sample": MLP_FLOP_PER_SAMPLE, "TFLOPs": round(net_loop_tf, 2),
                      "frac_of_mfma_peak": round(net_loop_tf / (F16_MFMA_PEAK_TF if fp16 else F32_MFMA_PEAK_TF), 4),
                      "peak_used": "fp16 dense MFMA 2.5 PF" if fp16 else ("fp32-input MFMA 157.3 TF (the kernel computes fp32-accurate products out of " + ("3 fp16" if x_form else "6 bf16") + " MFMAs)")},
        "all_samples_one_launch": {"launch_ms_fp32": round(t_net, 4), "launch_ms_fp16": round(t_net_h, 4), "achieved_GBps": round(bps * B / (t_used * 1e-3) / 1e9, 1),
                                   "frac_of_hbm_peak": round(bps * B / (t_used * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
    }
    extra = {
        "network": network,
        "hash_lookup": {"kernel": "k_grid_encode<2> (stand-alone hash-grid lookup, output [L,B,C] like the reference kernel)",
                        "achieved_GBps": round(grid_gbs, 1), "frac_of_hbm_peak": round(grid_gbs / HBM_PEAK_GBS, 4), "launch_ms": round(t_grid, 4),
                        "bytes_per_sample": HASH_BYTES_PER_SAMPLE, "launch_ms_direct_BLC_output": round(t_grid_bl, 4)},
        "breakdown_ms": {"stepforward_alone": round(t_sim, 4), "stepforward_persistent_alone": (round(t_sim_coop, 4) if t_sim_coop else None),
                         "render_frame_eager": round(t_frame, 4), "fused_from_trip": ff,
                         "march_per_launch_group": [round(float(v), 4) for v in march_ms[:(ff + 1 if ff >= 0 else real)]],
                         "network_per_trip_launch": [round(float(v), 4) for v in net_ms[:per_trip]],
                         "in_pipeline_march_per_launch_group": [round(float(v), 4) for v in g_march[:(ff + 1 if ff >= 0 else real)]] if g_march is not None else None,
                         "in_pipeline_network_per_trip_launch": [round(float(v), 4) for v in g_net[:per_trip]] if g_net is not None else None,
                         "local_global_iters_per_s": round(opt["sim_iters"] / (t_sim * 1e-3), 1)},
    }
    return st, roofline, extra


def graph_stamps(make_harness, args):
    """Time stamps (one-lane kernels reading the 100 MHz clock) captured inside the pipeline's render graphs around each launch group, while `lanes` frames,
    the simulator and the frame copies run concurrently: the durations of the mode that produced `value` (HIP events recorded in a graph cannot be timed).
    Returns ((march_ms per launch group, network_ms per trip launch), the render options the pipeline picked)."""
    h = make_harness()
    h.capture_pipelined(lanes=args.lanes, depth=args.depth, n_trips=args.trips, _time_trips=True, copy_on=args.copy_on)
    for _ in range(4 * args.lanes * args.depth):
        h.step_pipelined()
    h.drain_pipeline()
    m_all, n_all = [], []
    for ws in range(args.lanes * args.depth):
        a, b = h.model.trip_times(slot=ws)
        m_all.append(a)
        n_all.append(b)
    n = min(len(a) for a in m_all)
    kw = {k: h._pipe_backend.kw.get(k) for k in ("fused_from", "fused_grid", "fused_whole", "fused_fold", "march_throughput", "march_throughput_trips")}
    res = (np.median(np.array([a[:n] for a in m_all]), axis=0), np.median(np.array([b[:n] for b in n_all]), axis=0)), kw
    del h
    torch.cuda.empty_cache()
    return res


def pipelined_extras(make_harness, args, steps):
    """Short measurements beside the headline (N = 1): device-resident rate, one-frame-at-a-time latency, kernel durations inside the pipelined graphs."""
    res = {}

    def rate(h, n):
        for _ in range(2 * args.lanes * args.depth + 4):
            h.step_pipelined()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            h.step_pipelined()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        h.drain_pipeline()
        return dt / n * 1e3
    h = make_harness()
    h.capture_pipelined(lanes=args.lanes, depth=args.depth, n_trips=args.trips, copy_out=False)
    ms = rate(h, steps)
    res["device_resident"] = {"steps_per_s": round(1e3 / ms, 2), "ms_per_step": round(ms, 4), "note": "same pipeline without the D2H of image/depth/depth_0"}
    del h
    torch.cuda.empty_cache()
    if args.lanes != 2 and args.config == "chair" and args.sigma_gain == 1.0:
        # what a rank of a multi-GPU job runs (two render lanes beside the simulator and the RCCL stream): the per-rank rate behind `predicted_scaling`.
        # In a process of its own (see the sigma-gain sweep below: stream-to-queue mapping of a process that has already created several pipelines)
        import subprocess
        cmd = [sys.executable, os.path.abspath(__file__), "--lanes", "2", "--steps", str(max(100, steps)), "--warmup", "20", "--no-extras", "--no-cpu-baseline",
               "--depth", str(args.depth), "--copy-on", args.copy_on]
        try:
            o_ = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
            d = json.loads([ln for ln in o_.stdout.splitlines() if ln.startswith("{")][-1])
            res["two_lanes"] = {"steps_per_s": d["value"], "ms_per_step": d["ms_per_step"], "verified": d.get("verified"), "launch": d["config"]["launch"],
                                "note": "the same bench line with --lanes 2 (incl. D2H)"}
        except Exception as e:  # noqa: BLE001 — measurement only
            res["two_lanes_error"] = f"{type(e).__name__}: {str(e)[:200]}"
    h = make_harness()
    h.capture_pipelined(lanes=1, depth=2, n_trips=args.trips, sim_ahead=1, copy_on=args.copy_on)
    ms = rate(h, steps)
    res["latency_ms_per_step"] = round(ms, 4)
    res["latency_note"] = "lanes = 1: one render at a time (the next substep overlaps it), incl. D2H — the GUI-equivalent frame time"
    del h
    torch.cuda.empty_cache()
    if args.config == "chair" and args.sigma_gain == 1.0:
        # how steps/s moves with the samples per frame: the synthetic checkpoint's density scaled (fewer / more samples before a ray saturates).
        # Each point in a process of its own: HIP multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues, and in a process that has already
        # created a few harnesses two busy lanes of a new one can land on the same queue (measured: 547 instead of 1 537 steps/s at gain 3).
        import subprocess
        sweep = []
        for g in (0.3, 3.0):
            cmd = [sys.executable, os.path.abspath(__file__), "--sigma-gain", str(g), "--steps", str(max(100, steps)), "--warmup", "20", "--no-extras",
                   "--no-cpu-baseline", "--lanes", str(args.lanes), "--depth", str(args.depth), "--copy-on", args.copy_on]
            try:
                out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
                d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
                sweep.append({"sigma_gain": g, "steps_per_s": d["value"], "samples_per_frame": d["config"]["samples_per_frame"],
                              "trips_per_frame": d["config"]["trips_per_frame"], "mean_samples_per_hit_ray": d["config"]["mean_samples_per_hit_ray"]})
            except Exception as e:  # noqa: BLE001 — measurement only
                sweep.append({"sigma_gain": g, "error": f"{type(e).__name__}: {str(e)[:200]}"})
        res["sigma_gain_sweep"] = {"points": sweep, "note": "the same bench line at other densities of the synthetic checkpoint (gain 1 is `value` itself)"}
    if args.config == "chair" and args.sigma_gain == 1.0:
        # the other single-GPU configurations of BASELINE.json (configs[2] trex option set, configs[4] stress) through the same command, each in a process of
        # its own: `value` and the dominant kernel's roofline block, so that the driver's record of the default run covers every single-GPU config
        import subprocess
        oc = {}
        # ... and SURVEY 8d's second pass of config 2: the chair under a constant update_force on the middle integration point
        runs = [("trex", ["--config", "trex"]), ("stress", ["--config", "stress"]), ("chair_forced", ["--force", "300", "100", "-200"])]
        for cfg, extra_args in runs:
            cmd = [sys.executable, os.path.abspath(__file__)] + extra_args + ["--steps", "100", "--warmup", "20", "--no-extras", "--no-cpu-baseline"]
            try:
                o_ = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
                d = json.loads([ln for ln in o_.stdout.splitlines() if ln.startswith("{")][-1])
                r = d["roofline"]
                oc[cfg] = {"value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "verified": d.get("verified"), "workload": d["config"]["workload"],
                           "samples_per_frame": d["config"]["samples_per_frame"], "trips_per_frame": d["config"]["trips_per_frame"], "launch": d["config"]["launch"],
                           "roofline": {k: r.get(k) for k in ("kernel", "achieved", "frac", "launch_ms", "launch_ms_alone", "frac_alone", "ms_per_frame", "traffic", "first_trips_march")},
                           "network_frac": d["network"]["frac"], "render_frame_eager_ms": d["breakdown_ms"]["render_frame_eager"],
                           "substep_ms_alone": d["breakdown_ms"]["stepforward_alone"]}
                if cfg in ("trex", "stress"):   # the two-lane rate of a rank of a multi-GPU job on this configuration, for its predicted_scaling below
                    o2 = subprocess.run(cmd + ["--lanes", "2"], capture_output=True, text=True, timeout=600)
                    d2 = json.loads([ln for ln in o2.stdout.splitlines() if ln.startswith("{")][-1])
                    oc[cfg]["two_lanes_steps_per_s"] = d2["value"]
            except Exception as e:  # noqa: BLE001 — measurement only
                oc[cfg] = {"error": f"{type(e).__name__}: {str(e)[:200]}"}
        res["other_configs"] = oc
    torch.cuda.empty_cache()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", choices=("chair", "stress", "trex"), default="chair")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the short additional measurements after the timed region")
    ap.add_argument("--eager", action="store_true", help="launch every kernel from the host instead of replaying captured HIP graphs")
    ap.add_argument("--prime", type=int, default=40, help="untimed priming steps right after the graph captures, before the warm-up steps (see the timed loop)")
    ap.add_argument("--trips", type=int, default=None, help="render-loop trips baked into the captured graphs (default: what one eager frame needs + 2; a frame "
                    "that needs more is continued when it is retired)")
    ap.add_argument("--copy-on", choices=("copy", "lane", "sim", "host"), default="host", help="stream of the per-frame D2H: a copy stream of its own, or the frame's render stream")
    ap.add_argument("--lanes", type=int, default=3,
                    help="render streams (3 render streams + the simulator stream = the 4 compute pipes of an XCD, more streams only time-slice)")
    ap.add_argument("--depth", type=int, default=2, help="workspaces per render stream")
    ap.add_argument("--single-graph", action="store_true", help="whole step as ONE captured graph (sim on a forked stream), one frame at a time")
    ap.add_argument("--no-d2h", action="store_true", help="leave the outputs on the device (then `value` is the device-resident rate and says so)")
    ap.add_argument("--whole-frame", action="store_true", help="--config stress: render the frame in one shot instead of ray batches of 4096")
    ap.add_argument("--cpu-budget", type=float, default=20.0)
    ap.add_argument("--dedicated-sim", choices=("auto", "on", "off"), default="auto",
                    help="N > 1: the sim owner only simulates and broadcasts, the other ranks render (auto: from 3 ranks on, frames.dedicated_sim_default)")
    ap.add_argument("--force", type=float, nargs=3, default=None, metavar=("FX", "FY", "FZ"),
                    help="constant update_force on the middle integration point (SURVEY 8d, config 2 second pass); default: per config")
    ap.add_argument("--sim-on-lanes", action="store_true", help="experiment: no simulator stream, substep g rides on render lane g %% lanes (frames.FramePipeline.sim_on_lanes)")
    ap.add_argument("--probe", choices=("none", "no-substep", "sim-priority", "sim-cus", "render-excl"), default="none",
                    help="diagnosis (value is then NOT the benchmark): the pipeline without the substep's launches / with the simulator stream at high priority / on 16 CUs of its own")
    ap.add_argument("--form", choices=("auto", "whole", "fold", "plain", "trips"), default="auto",
                    help="form of a frame's launches in the pipeline (auto: what harness.capture_pipelined picks from a warm-up frame): the whole frame in the fused "
                         "launch / the first trip's network + composite + compaction folded into it / later trips fused only / trip-by-trip launches")
    ap.add_argument("--parallelism", choices=("frame", "tile"), default="frame",
                    help="N > 1: 'frame' = whole frames round-robin over the ranks (throughput; BASELINE configs[3]); 'tile' = every frame split into 8 x 8 pixel tiles "
                         "over ALL ranks + one all_gather_into_tensor (latency; frames.TileParallel; SURVEY 8e's alternative).  With N = 1 'tile' runs the same "
                         "eager one-frame-at-a-time path on one rank (PN_FORCE_DIST=1: inside a one-rank RCCL group, collectives included)")
    ap.add_argument("--sigma-gain", type=float, default=1.0, help="scales the synthetic checkpoint's density (samples per frame fall as it rises)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    rccl_ranks = None
    force_dist = world == 1 and os.environ.get("PN_FORCE_DIST", "") == "1"   # a process group of ONE rank (tests: RCCL itself on a one-GPU box)
    if force_dist:
        os.environ.setdefault("MASTER_PORT", "29577")
    if world > 1 or force_dist:
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
        rccl_ranks = dist.get_world_size()
    else:
        dev_index = 0
        torch.cuda.set_device(0)
    dev = torch.device("cuda", dev_index)

    from pienerf_amd.harness import SimRenderHarness
    opt, cloud, ckpt, pose, force, workload = make_config(args.config, args.sigma_gain)
    staged = args.config == "stress" and not args.whole_frame
    if staged:
        opt["ray_batch"] = int(opt.get("max_ray_batch", 4096))  # BASELINE configs[4]: "4096 rays/batch" (get_opts.py:24)
    if args.force is not None:
        force = np.asarray(args.force, dtype=np.float64)

    def make_harness():
        hh = SimRenderHarness(opt, cloud=cloud, ckpt=ckpt, device=dev)
        hh.pose = pose
        if force is not None and (world == 1 or rank == 0):  # Simulator.update_force (solver.py:578-588): the dragged-point load of the GUI
            hh.sim.update_force(hh.sim.n_IP // 2, force)
        return hh
    h = make_harness()
    copy_out = not args.no_d2h

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()
        h.wait_frame_copies()   # the copier thread's SDMA copies are on no stream: the clock stops when the last frame enqueued has landed in host memory

    frames_done = [0]
    tile_mode = args.parallelism == "tile"
    if tile_mode:
        # every rank renders 1/world of each frame's 8 x 8 tiles from the owner's broadcast DOF snapshot; one all-gather hands every rank the frame.
        # Launches are eager, one frame at a time: this form divides a frame's LATENCY by the rank count (the frame-parallel form only adds throughput)
        if world > 1:
            from pienerf_amd.frames import broadcast_tensors
            m = h.model
            broadcast_tensors([m.encoder.embeddings.data, m.density_bitfield] + [l.weight.data for l in list(m.sigma_net) + list(m.color_net)], src=0)
            m._net_sig = None
        h.capture_tile_parallel(_force_collectives=force_dist)

        def run_steps(n):
            for _ in range(n):
                o_ = h.step_tile_parallel()
                if copy_out and rank == 0:
                    h.to_host(o_)
            frames_done[0] += n
        launch = f"eager launches, one frame at a time, each frame's 8 x 8 pixel tiles interleaved over {world} rank(s) + one all_gather_into_tensor"
    elif world == 1:
        if args.eager:
            def run_steps(n):
                for _ in range(n):
                    o_ = h.step()
                    if copy_out:
                        h.to_host(o_)
            launch = "eager (one kernel launch per host call)"
        elif args.single_graph:
            # the whole step (get_rays, get_IP_info, stepforward on a forked stream, render prologue + loop trips + epilogue) is one
            # captured HIP graph; each replay first completes the previous frame (continuing it if it ran out of trips)
            h.capture(n_trips=args.trips or 8)

            def run_steps(n):
                for _ in range(n):
                    o_ = h.step_graph()
                    if copy_out:
                        h.to_host(o_)
            launch = f"one hip graph per step, {args.trips or 8} trips"
        else:
            # --config stress: opt["ray_batch"] = 4096 (set above) — the frame's 157 ray batches keep their own trip schedules inside the same
            # launches (pn_render_opts.ray_batch), so the staged frame runs on the same pipeline as the frame in one piece
            probe_kw = {"no-substep": dict(_probe_no_substep=True), "sim-priority": dict(sim_priority=-1), "sim-cus": dict(sim_cus=int(os.environ.get("PN_PROBE_SIM_CUS", "16"))),
                        "render-excl": dict(sim_cus=-int(os.environ.get("PN_PROBE_RENDER_EXCL", "32")))}.get(args.probe, {})
            form_render_kw = {"whole": dict(fused_from=0, fused_whole=True, fused_fold=False), "fold": dict(fused_from=1, fused_whole=False, fused_fold=True),
                              "plain": dict(fused_from=1, fused_whole=False, fused_fold=False), "trips": dict(fused_from=-1, fused_whole=False, fused_fold=False)}.get(args.form)
            if form_render_kw is not None:
                probe_kw["render_kw"] = form_render_kw
            if args.sim_on_lanes:
                probe_kw["sim_on_lanes"] = True
            h.capture_pipelined(lanes=args.lanes, depth=args.depth, n_trips=args.trips, copy_out=copy_out, copy_on=args.copy_on, **probe_kw)
            args.trips = h._pipe_backend.trips
            run_steps = lambda n: [h.step_pipelined() for _ in range(n)]
            launch = (f"hip graphs, {args.trips} trips, {args.lanes} render streams x {args.depth} workspaces, "
                      + (f"march pass 1 of the first {h._pipe_backend.kw.get('march_throughput_trips', 1)} trip(s) in its throughput form (one lane per ray, "
                         f"{h._pipe_backend.kw.get('march_throughput')} rounds; pn_render_opts.throughput / throughput_trips), " if args.lanes > 1 else "")
                      + ("" if staged else "alive list in 16 x 4 pixel tiles (pn_render_opts.ray_tile_w), ")
                      + (f"loop trips from trip {h._pipe_backend.kw.get('fused_from')} on as ONE persistent launch"
                         + (" (first trip included: fused_whole)" if h._pipe_backend.kw.get("fused_whole") else "")
                         + (" with the first trip's network / composite / compaction inside it (fused_fold)" if h._pipe_backend.kw.get("fused_fold") else "")
                         + f" on {h._pipe_backend.kw.get('fused_grid') or 'all'} CUs (pn_render_opts.fused_from / fused_grid), " if (h._pipe_backend.kw.get("fused_from") or 0) >= 0 and not staged else "")
                      + "simulator running ahead, D2H on "
                      + {"copy": "a copy stream", "lane": "the frame's render stream", "sim": "the simulator stream", "host": "no stream (copier thread + SDMA through the HSA runtime)"}[args.copy_on]
                      + (f"; rays in batches of {opt['ray_batch']} with per-batch trip schedules (max_ray_batch), all batches in the same launches" if staged else ""))
    else:
        from pienerf_amd.frames import broadcast_tensors
        m = h.model
        broadcast_tensors([m.encoder.embeddings.data, m.density_bitfield] + [l.weight.data for l in list(m.sigma_net) + list(m.color_net)], src=0)
        m._net_sig = None
        # ROCm time-slices badly once more than 4 hardware queues are busy (launch structure): with the simulator stream
        # and the RCCL communication stream that leaves 2 render lanes per rank; a rank renders only every world-th frame anyway
        args.lanes = min(args.lanes, 2)
        dedicated = {"auto": None, "on": True, "off": False}[args.dedicated_sim]
        h.capture_frame_parallel(lanes=args.lanes, depth=args.depth, n_trips=args.trips or 8, dedicated_sim=dedicated, copy_out=copy_out, copy_on=args.copy_on)
        args.trips = h._pipe_backend.trips

        def run_steps(n):
            for _ in range(n * world):
                frames_done[0] += len(h.step_frame_parallel())
        launch = f"hip graphs, {args.trips} trips, {args.lanes} render streams x {args.depth} workspaces per rank"

    with torch.no_grad():
        # (1) exactly what the command line says, right after the graph captures: W untimed warm-up steps, K timed steps -> `value_unprimed`.
        barrier()
        run_steps(args.warmup)
        barrier()
        t0 = time.perf_counter()
        run_steps(args.steps)
        barrier()
        elapsed_unprimed = time.perf_counter() - t0
        # (2) the same again once the pipeline has been running for a while -> `value` (steady state).  The first replays of the lanes x depth
        # render graphs, the first touches of the pinned buffers and the clocks coming up from the host-bound capture phase make a short run slower:
        # measured in round 2 with 5 / 20 / 40 untimed steps in front, 20 timed steps took 25.2 / 22.2 / 18.7 ms.  Both figures are in the line.
        run_steps(args.prime)
        barrier()
        run_steps(args.warmup)
        barrier()
        t0 = time.perf_counter()
        run_steps(args.steps)
        barrier()
        elapsed = time.perf_counter() - t0
        continued = 0
        verified = None
        if tile_mode:
            pass
        elif world > 1 or not (args.eager or args.single_graph):  # the last frames in flight are retired here
            frames_done[0] += len(h.drain_pipeline())
            continued = h._pipe_backend.continued
            # ... and the last frame the pipeline delivered is rendered again, launch by launch, from the state its workspace holds: bit for bit
            if h._pipe.last_ws is not None:
                verified = h.verify_last_frame()
        elif args.single_graph:
            h._check_previous_graph_frame()
            continued = getattr(h, "graph_continued", 0)
    per_rank_frames = None
    if world > 1:
        t = torch.tensor([elapsed, elapsed_unprimed], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed, elapsed_unprimed = float(t[0].item()), float(t[1].item())
        fr = torch.zeros(world, dtype=torch.int64, device=dev)
        fr[rank] = frames_done[0]
        torch.distributed.all_reduce(fr)
        per_rank_frames = [int(v) for v in fr.tolist()]

    if rank == 0:
        del_h = h
        pipelined = world == 1 and not (args.eager or args.single_graph or tile_mode)
        form_kw = {k: h._pipe_backend.kw.get(k) for k in ("fused_from", "fused_grid", "fused_whole", "fused_fold", "march_throughput", "march_throughput_trips")} if hasattr(h, "_pipe_backend") else None
        with torch.no_grad():
            graph_ms = None
            if pipelined:
                try:
                    graph_ms, form_kw = graph_stamps(make_harness, args)
                except Exception as e:  # noqa: BLE001 — measurement only
                    print(f"graph stamps failed: {type(e).__name__}: {str(e)[:200]}", file=sys.stderr)
            hk = make_harness() if (world > 1 or not args.eager) else h   # a fresh eager harness for the per-kernel report
            opt["_config_name"] = args.config if args.sigma_gain == 1.0 else "other"
            st, roofline, extra = kernel_report(hk, opt, dev, form_kw, graph_ms)
        res = {
            "metric": "sim+render steps/s @800x800 chair" if args.config == "chair" else f"sim+render steps/s, {args.config} configuration",
            "value": round(args.steps * (1 if tile_mode else world) / elapsed, 3), "unit": "steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True,
            "value_unprimed": round(args.steps * (1 if tile_mode else world) / elapsed_unprimed, 3),
            "value_note": (f"`value`: K = {args.steps} timed steps after {args.prime} priming + W = {args.warmup} warm-up steps of the already running pipeline (steady state); "
                           f"`value_unprimed`: the first K timed steps after only the W warm-up steps, right behind the graph captures"),
            "verified": (bool(verified["ok"]) if verified else None),
            "verified_note": (f"frame {verified['frame']} (the last one the timed pipeline delivered, as copied to host memory) == a blocking launch-by-launch render of the "
                              f"same integration-point state and pose, bit for bit on image and depth_0: max abs difference {verified.get('max_abs_diff')}" if verified else
                              "not a pipelined run"),
            "scaling": ("strong" if tile_mode else "weak"), "vs_baseline": None, "dtype": ("f16 tables+MLP / f32 march / f64 sim" if opt.get("fp16") else "f32 render / f64 sim"), "data": "synthetic",
            "config": {"workload": workload + (f", constant force {[float(v) for v in force]} on IP {hk.sim.n_IP // 2}" if force is not None else ", gravity only")
                       + ("" if copy_out else " [--no-d2h: outputs left on the device]") + ("" if args.probe == "none" else f" [--probe {args.probe}: a diagnosis run, not the benchmark]"),
                       "rays": opt["W"] * opt["H"], "n_IP": hk.sim.n_IP, "n_kernels": hk.sim.n_k, "n_points": int(len(cloud["pos"])),
                       "samples_per_frame": st["samples"], "trips_per_frame": st["trips"], "d2h_bytes_per_step": (opt["W"] * opt["H"] * 20 if copy_out else 0),
                       "frames_continued_past_captured_trips": continued, "launch": launch, "prime_steps": args.prime, "sigma_gain": args.sigma_gain,
                       "hit_rays": st["hit_rays"], "mean_samples_per_hit_ray": round(st["samples"] / max(1, st["hit_rays"]), 2),
                       "ranks": world, "dist_backend": (os.environ.get("PN_DIST_BACKEND", "nccl") if world > 1 else None), "rccl_ranks": rccl_ranks,
                       "frames_per_rank": per_rank_frames, "dedicated_sim": (bool(del_h._pipe.dedicated) if (world > 1 and not tile_mode) else None),
                       "parallelism": (f"tile-parallel x{world} ({rccl_ranks} RCCL ranks{', collectives forced' if force_dist else ''}): every frame's 8 x 8 pixel tiles interleaved over the "
                                       f"ranks, dof snapshot broadcast + one all_gather_into_tensor per frame; ms_per_step is a frame's latency") if tile_mode else
                                      (f"frame-parallel x{world} ({rccl_ranks} RCCL ranks), dof snapshots broadcast over RCCL, "
                                       + ("rank 0 simulates only, frames round-robin over the other ranks" if del_h._pipe.dedicated else "frames round-robin over all ranks")
                                       + f", frames per rank {per_rank_frames}") if world > 1 else "single GPU"},
            "roofline": roofline,
        }
        res.update(extra)
        # what bounds the frame-parallel job: the sim owner's substep rate — the simulator is time-sequential — against N (or N - 1
        # with a dedicated owner) ranks rendering at the single-GPU rate
        t_launch, t_coop = extra["breakdown_ms"]["stepforward_alone"], extra["breakdown_ms"]["stepforward_persistent_alone"]
        t_sub = min(t_launch, t_coop) if t_coop else t_launch   # (since round 5 the launch form — 21 launches in its cell form — is the faster one on the chair)
        res["frame_parallel_ceiling"] = {"substep_ms_alone": t_sub, "owner_frames_per_s": round(1e3 / t_sub, 1),
                                         "substep_form": "persistent kernel" if (t_coop and t_coop < t_launch) else "launch form (cells: calc_elastic + collect_rhs_IP as one launch)",
                                         "owner_frames_per_s_launch_form": round(1e3 / t_launch, 1),
                                         "owner_frames_per_s_persistent_form": (round(1e3 / t_coop, 1) if t_coop else None),
                                         "physical_ceiling_note": "near-linear scaling to 8 GPUs (7 rendering ranks x the two-lane rate) would need a substep of ~0.07 ms, i.e. ~3 us per "
                                                                  "dependent phase of its 10 local/global iterations — below what one launch boundary or one device-wide exchange costs "
                                                                  "(~4.5 us): the simulator is time-sequential (Amdahl), see DESIGN.md 6",
                                         "note": "steps/s of an N-GPU frame-parallel job <= min(owner_frames_per_s [owner dedicated: its substep has the GPU to itself], "
                                                 "renderers x the single-GPU render rate); with the owner also rendering its substep shares the GPU and is ~1.8x slower"}
        if world == 1 and not args.no_extras and not (args.eager or args.single_graph or tile_mode):
            with torch.no_grad():
                res.update(pipelined_extras(make_harness, args, max(40, min(args.steps, 120))))
        if world == 1 and "two_lanes" in res:
            # what `python bench.py --gpus N` should print on one node, from this GPU's measurements: N = 2: both ranks render, the owner's
            # substeps share its GPU with its renders (measured x1.8 slower than alone); N >= 3: rank 0 only simulates (launch form) and broadcasts each
            # <= 82 KB snapshot, N - 1 ranks render with two lanes each; the broadcast (~20 us over xGMI) overlaps on the communication stream
            r2, own = res["two_lanes"]["steps_per_s"], res["frame_parallel_ceiling"]["owner_frames_per_s_launch_form"]
            def predict(v1, r2_, own_):
                return {"steps_per_s": {"1": v1, "2": round(min(own_ / 1.8, 2 * r2_), 1), "4": round(min(own_, 3 * r2_), 1), "8": round(min(own_, 7 * r2_), 1)},
                        "bound": {"2": "renderers" if 2 * r2_ < own_ / 1.8 else "sim owner (shared GPU)", "4": "renderers" if 3 * r2_ < own_ else "sim owner",
                                  "8": "renderers" if 7 * r2_ < own_ else "sim owner"}}
            res["predicted_scaling"] = dict(predict(res["value"], r2, own),
                                            note="prediction, not a measurement (no multi-GPU node was available to the builder): min(sim owner's substep rate, rendering ranks x the "
                                                 "two-lane single-GPU rate); the simulator is time-sequential, so the owner's substep rate caps the job whatever N")
            for cfg in ("trex", "stress"):
                oc_ = res.get("other_configs", {}).get(cfg, {})
                if "two_lanes_steps_per_s" in oc_ and oc_.get("substep_ms_alone"):
                    res["predicted_scaling"][cfg] = predict(oc_["value"], oc_["two_lanes_steps_per_s"], round(1e3 / oc_["substep_ms_alone"], 1))
        if staged and not args.no_extras:  # the same configuration with the frame rendered in one shot (what render_deformed does with these options in the reference)
            with torch.no_grad():
                opt.pop("ray_batch")
                hw = make_harness()
                opt["ray_batch"] = hw_batch = int(opt.get("max_ray_batch", 4096))
                hw.capture_pipelined(lanes=args.lanes, depth=args.depth, n_trips=None, copy_out=copy_out)
                for _ in range(12):
                    hw.step_pipelined()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(60):
                    hw.step_pipelined()
                torch.cuda.synchronize()
                res["whole_frame_pipelined"] = {"steps_per_s": round(60 / (time.perf_counter() - t1), 2), "trips_captured": hw._pipe_backend.trips,
                                                "note": "same scene, options and precision with the 800x800 frame rendered in one shot by the frame pipeline"}
                hw.drain_pipeline()
        if not args.no_cpu_baseline and world == 1:  # a reported baseline, timed on rank 0 at N = 1 only
            res["cpu_baseline"] = cpu_baseline(opt, cloud, ckpt, pose, force, args.cpu_budget)
        print(json.dumps(res))
    if world > 1 or force_dist:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
