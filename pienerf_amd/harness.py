"""Headless counterpart of the reference's per-frame harness.

One ``step()`` = what ``NeRFSimGUI.test_step`` -> ``Trainer.test_gui`` -> ``Trainer.test_step`` do for one GUI frame
(nerf/gui.py:556-645, nerf/trainer.py:284-329,531-602), minus the window:

    rays = get_rays(pose, intrinsics, H, W)            # trainer.py:543
    IP_pos, IP_F, IP_dF = solver.get_IP_info()         # trainer.py:303   (state BEFORE the step: render lags sim by one frame)
    model.p_def / IP_F / IP_dF = ...                   # trainer.py:304-306
    solver.stepforward()                               # trainer.py:308
    model.render_deformed(rays_o, rays_d, **vars(opt)) # trainer.py:316-318

Start-up mirrors main_gui.py:26-56.
"""
import contextlib
import gc
import os

import numpy as np
import torch

from . import scene
from .nerf.network import NeRFNetwork
from .nerf.utils import get_rays
from .simulator.solver import Simulator


@contextlib.contextmanager
def _capture(graph, **kw):
    """torch.cuda.graph with the cyclic garbage collector held off: a collection that runs DURING a stream capture may free another harness
    (workspaces, streams, graphs of an earlier capture) — hipFree / hipGraphDestroy inside a capture abort the process."""
    gc.collect()
    was = gc.isenabled()
    gc.disable()
    try:
        with torch.cuda.graph(graph, capture_error_mode="thread_local", **kw):
            yield
    finally:
        if was:
            gc.enable()


# pn_render_opts.throughput_trips: a trip marches in the throughput form (one lane per ray) when it has at least this many alive rays — 2048 waves,
# two per SIMD (measured: trex option set, second trip of 246 k rays: 1 191 -> 1 321 steps/s; chair, second trip of 41 k rays: 1 650 -> 1 480)
THROUGHPUT_FORM_MIN_RAYS = 131072

class SimRenderHarness:
    def __init__(self, opt=None, cloud=None, ckpt=None, device="cuda", overlap_sim=True):
        self.opt = dict(opt or scene.default_opt())
        o = self.opt
        self.device = torch.device(device)
        self.cloud = cloud if cloud is not None else scene.make_chair_points(hgs=o["hash_grid_size"], bound=o["bound"])
        self.ckpt = ckpt if ckpt is not None else scene.make_checkpoint(bound=o["bound"])
        # main_gui.py:26-34
        self.model = NeRFNetwork(encoding="hashgrid", bound=o["bound"], cuda_ray=True, density_scale=1, min_near=o["min_near"],
                                 density_thresh=o["density_thresh"], bg_radius=o["bg_radius"]).to(self.device)
        self.model.load_checkpoint_dict(self.ckpt)
        self.model.eval()
        # main_gui.py:39-48
        self.sim = Simulator(dt=o["sim_dt"], iters=o["sim_iters"], bbox=torch.tensor([2.0 * o["bound"]] * 3), dx=o["sim_dx"], stiff=o["sim_stiff"],
                             base=torch.tensor([-o["bound"]] * 3), device=self.device)
        c = self.cloud
        self.sim.InitializeFromArrays(c["pos"], c["mass"], c["mu"], c["lam"], c["pin"])
        # main_gui.py:50-56
        IP_pos, IP_F, IP_dF = self.sim.get_IP_info()
        m = self.model
        m.p_ori, m.p_def, m.IP_F, m.IP_dF = IP_pos, IP_pos, IP_F, IP_dF
        m.IP_dx = self.sim.dx * 1.05
        self.frame = 0
        self.pose = scene.orbit_pose(o["radius"])
        self.intrinsics = scene.orbit_intrinsics(o["W"], o["H"], o["fovy"])
        self._pose_dev = None
        self.overlap_sim = overlap_sim
        if overlap_sim:
            self._sim_stream = torch.cuda.Stream(self.device)
            self._ip_ready = torch.cuda.Event()
            self._sim_done = torch.cuda.Event()
            self._sim_done.record(torch.cuda.current_stream(self.device))
            self.sim.force_stream = self._sim_stream  # update_force / clear_force are ordered between two substeps (solver.py:578-593)

    def synchronize(self):
        torch.cuda.synchronize(self.device)
        self._check_persistent_substep()

    def _amp(self):
        """Trainer.test_gui renders inside ``torch.cuda.amp.autocast(enabled=self.fp16)`` (trainer.py:561).  main_gui.py builds its Trainer
        without fp16 (fp32 render, the default here); opt['fp16'] = True is main_train.py's Trainer(fp16=opt.fp16) and BASELINE configs[4]:
        fp16 hash tables + half Linear layers (renderer.rund_cuda reads the autocast state)."""
        return torch.autocast("cuda", dtype=torch.float16, enabled=bool(self.opt.get("fp16", False)))

    def render_kwargs(self):
        """**vars(opt) as the reference passes it (trainer.py:318); renderer reads these by name."""
        return dict(self.opt)

    @torch.no_grad()
    def step(self, pose=None, intrinsics=None, W=None, H=None, simulate=True, collect_stats=False, fused=True, render_kw=None):
        """render_kw: options of THIS render on top of self.opt (the pipeline's probes of a launch form pass theirs here instead of editing self.opt)."""
        o = self.opt
        W, H = W or o["W"], H or o["H"]
        pose = self.pose if pose is None else pose
        intrinsics = self.intrinsics if intrinsics is None else intrinsics
        key = np.asarray(pose, np.float32).tobytes()
        if self._pose_dev is None or self._pose_dev[0] != key:  # trainer.py:541 uploads the pose every frame; re-upload only when it changes
            self._pose_dev = (key, torch.from_numpy(np.asarray(pose, np.float32)).unsqueeze(0).to(self.device))
        rays = get_rays(self._pose_dev[1], intrinsics, H, W, -1)
        m = self.model
        if simulate:
            main = torch.cuda.current_stream(self.device)
            if self.overlap_sim:
                main.wait_event(self._sim_done)            # the previous substep must have finished before dof is read
            IP_pos, IP_F, IP_dF = self.sim.get_IP_info()
            m.p_def, m.IP_F, m.IP_dF = IP_pos, IP_F, IP_dF
            if self.overlap_sim:
                # the substep only feeds the NEXT frame (render lags sim by one frame, trainer.py:300-318), so it runs on a
                # side stream concurrently with this frame's render
                self._ip_ready.record(main)
                self._sim_stream.wait_event(self._ip_ready)
                with torch.cuda.stream(self._sim_stream):
                    self.sim.stepforward()
                    self._sim_done.record(self._sim_stream)
            else:
                self.sim.stepforward()
            self.frame += 1
        kw = self.render_kwargs()
        kw.update(render_kw or {})
        kw["collect_stats"] = collect_stats
        with self._amp():
            if fused:
                out = m.render_deformed(rays["rays_o"], rays["rays_d"], staged=True, bg_color=None, perturb=False, **kw)
            else:
                out = m.rund_cuda_ops(rays["rays_o"], rays["rays_d"], bg_color=None, perturb=False, **kw)
        return {"image": out["image"].reshape(-1, H, W, 3), "depth": out["depth"].reshape(-1, H, W), "depth_0": out["depth_0"].reshape(-1, H, W),
                "rays_o": rays["rays_o"], "rays_d": rays["rays_d"]}

    # ------------------------------------------------------------------ whole step as one HIP graph
    def _step_body(self, n_trips, W, H):
        """get_rays -> get_IP_info -> { stepforward || render_deformed } with fork/join on the current stream (capturable)."""
        m = self.model
        main = torch.cuda.current_stream(self.device)
        rays = get_rays(self._graph_pose, self.intrinsics, H, W, -1)
        IP_pos, IP_F, IP_dF = self.sim.get_IP_info()
        m.p_def, m.IP_F, m.IP_dF = IP_pos, IP_F, IP_dF
        self._sim_stream.wait_stream(main)
        with torch.cuda.stream(self._sim_stream):
            self.sim.stepforward()
        kw = self.render_kwargs()
        kw["async_trips"] = n_trips
        with self._amp():
            out = m.render_deformed(rays["rays_o"], rays["rays_d"], staged=True, bg_color=None, perturb=False, **kw)
        main.wait_stream(self._sim_stream)
        # every tensor the graph touches is returned (and so stays referenced): memory freed after capture would go back to the
        # graph's pool and could be handed out again
        return {"image": out["image"].reshape(-1, H, W, 3), "depth": out["depth"].reshape(-1, H, W), "depth_0": out["depth_0"].reshape(-1, H, W),
                "weights_sum": out["weights_sum"], "rays_o": rays["rays_o"], "rays_d": rays["rays_d"], "_ip": (IP_pos, IP_F, IP_dF)}

    @torch.no_grad()
    def capture(self, n_trips=8, W=None, H=None):
        """Captures one whole step (~85 kernel launches on two streams) into a HIP graph.  `n_trips` render-loop trips are
        baked in (the chair needs 5); step_graph() verifies afterwards that no ray was left alive."""
        o = self.opt
        W, H = W or o["W"], H or o["H"]
        if not hasattr(self, "_sim_stream"):
            self._sim_stream = torch.cuda.Stream(self.device)
        self._graph_pose = torch.from_numpy(np.asarray(self.pose, np.float32)).unsqueeze(0).to(self.device)
        keep = (self.sim.dof.clone(), self.sim.dof_vel.clone())
        warm = torch.cuda.Stream(self.device)
        warm.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(warm):
            for _ in range(2):  # warm-up outside capture: lazily created handles, allocator pools
                self._step_body(n_trips, W, H)
        torch.cuda.current_stream(self.device).wait_stream(warm)
        torch.cuda.synchronize(self.device)
        self._graph = torch.cuda.CUDAGraph()
        with _capture(self._graph):
            self._graph_out = self._step_body(n_trips, W, H)
        self.sim.dof.copy_(keep[0])       # warm-up advanced the simulator; capture itself executes nothing
        self.sim.dof_vel.copy_(keep[1])
        self.sim.reset_warm_start()       # ... and its SVD warm start: a replay from here has the bits of a simulator that never warmed up
        self._graph_trips = n_trips
        self._graph_done = None
        self._graph_form_epoch = self._net_form_epoch()
        return self

    def _net_form_epoch(self):
        """How often the fp32 network switched between its fp16 hi/lo and bf16 forms (include/pienerf_hip.h: pn_net_form_epoch).  The per-layer scales
        live in device memory and follow an in-place weight refresh; the form is baked into every captured launch."""
        from ._lib import lib
        net = getattr(self.model, "_net", None)
        return int(lib().pn_net_form_epoch(net)) if net is not None else 0

    def _check_net_form(self, captured, what):
        if self._net_form_epoch() != captured:
            raise RuntimeError(f"the network's weights were refreshed into another arithmetic form (pn_net_form) since {what} was captured: its graphs "
                               f"hold the old form's kernels — capture again")

    @torch.no_grad()
    def step_graph(self, pose=None):
        """Replays the captured step.  Outputs are static tensors (overwritten by the next replay); they are complete — including frames
        that needed more trips than were captured — after the NEXT call or ``finish_graph_frame()``."""
        if getattr(self, "_graph", None) is None:
            self.capture()
        self._check_net_form(self._graph_form_epoch, "the step")
        self._check_previous_graph_frame()
        if pose is not None:
            self._graph_pose.copy_(torch.from_numpy(np.asarray(pose, np.float32)).unsqueeze(0))
        self._graph.replay()
        self._graph_done = torch.cuda.Event()
        self._graph_done.record(torch.cuda.current_stream(self.device))
        self.frame += 1
        return self._graph_out

    def finish_graph_frame(self):
        """Waits for the last replayed step and completes it; returns its outputs."""
        self._check_previous_graph_frame()
        return self._graph_out

    def _check_previous_graph_frame(self):
        """The captured step bakes in a trip count; a frame that still had rays alive after them is finished here with further trips (the
        reference's loop just keeps going, renderer.py:836-891) before its outputs are handed on."""
        if getattr(self, "_graph_done", None) is not None:
            self._graph_done.synchronize()  # normally long complete: the host only ever runs one frame ahead
            st = self.model.render_status(synchronize=False)
            self._graph_done = None
            if st["alive_at_exit"] > 0:
                g = self._graph_out
                with self._amp():
                    self.model.render_continue(0, g["rays_o"], g["rays_d"], g, bg_color=None, **self.render_kwargs())
                self.graph_continued = getattr(self, "graph_continued", 0) + 1

    # ------------------------------------------------------------------ several frames in flight (one GPU, or frame-parallel over a node)
    def _cu_masked_stream(self, first_cu, n_cu, invert):
        import ctypes

        from ._lib import check, lib
        total = lib().pn_device_cu_count()
        h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            check(lib().pn_stream_create_cu_mask(total, first_cu, n_cu, int(invert), ctypes.byref(h)), "stream_create_cu_mask")
        self._raw_streams = getattr(self, "_raw_streams", []) + [h.value]
        return torch.cuda.ExternalStream(h.value, device=self.device)

    @torch.no_grad()
    def capture_pipelined(self, lanes=2, n_trips=8, W=None, H=None, sim_ahead=None, depth=2, sim_priority=0, sim_cus=0, copy_out=True, group=None,
                          frame_parallel=False, sim_owner=0, dedicated_sim=None, on_retire=None, copy_on="host", _probe_no_substep=False, _time_trips=False,
                          render_kw=None, _force_collectives=False, sim_on_lanes=False):
        """Throughput mode (pienerf_amd/frames.py: FramePipeline): `lanes` render streams with `depth` workspaces each, the simulator running
        `sim_ahead` frames ahead on dof snapshots, every frame's image / depth / depth_0 copied to pinned host memory (the reference's
        device->host boundary, trainer.py:589-592; copy_out=False leaves the results on the device) — copy_on="host": by the library's copier
        thread through the HSA runtime's SDMA engine, no stream involved (csrc/pn_copier.hip; "lane" / "copy" / "sim": as a hipMemcpyAsync
        on the frame's render stream / a stream of its own / the simulator stream).  Everything a frame
        launches is captured in HIP graphs — one for the substep, one per workspace for get_rays + the render with `n_trips` loop trips
        (None: measured on one eager frame, + 2); a frame that still has rays alive after its trips is continued when it is retired
        (renderer.py:836-891).  frame_parallel=True (or ``capture_frame_parallel``): the same pipeline over the ranks of `group` — the sim
        owner broadcasts every snapshot (<= 82 KB) over RCCL, rank frames.frame_owner(f) renders frame f.

        ``step_pipelined(pose=None)`` enqueues the next frame (with its own camera pose, trainer.py:541) and returns the frames it retired:
        [(frame index, {'image','depth','depth_0': numpy views of the pinned buffers, 'device': the device tensors})]; a frame comes back
        lanes*depth steps after it went in, ``drain_pipeline()`` returns the rest.  The numpy arrays stay valid for another lanes*depth steps
        (two pinned sets per workspace alternate); the device tensors are the graph's static outputs and are overwritten by the workspace's
        next frame, i.e. any time after this call returns: a consumer of the device copy hooks in with on_retire(frame, result), which runs
        when the frame is complete and before its workspace is handed to the next frame.  Every frame is rendered from the state before its own
        substep (trainer.py:300-318); ``self.sim.dof`` is `sim_ahead + 1` substeps in front of the last enqueued frame, and a force set with
        update_force() acts from the next substep that is enqueued (it is ordered on the simulator's stream)."""
        import torch.distributed as dist

        from .frames import FramePipeline
        o = self.opt
        W, H = W or o["W"], H or o["H"]
        on = bool(frame_parallel) and dist.is_available() and dist.is_initialized()
        world = dist.get_world_size(group) if on else 1
        rank = dist.get_rank(group) if on else 0
        if n_trips is None:  # calibrate: the trips one eager frame needs from the current state and pose, plus a margin
            self.step(simulate=False, collect_stats=True, W=W, H=H)
            n_trips = int(self.model.last_stats["trips"]) + 2  # every captured trip costs five launches whether it has rays or not
        from .frames import dedicated_sim_default
        if (world > 1 and (dedicated_sim_default(world) if dedicated_sim is None else bool(dedicated_sim)) and rank == sim_owner
                and dist.get_backend(group) == "nccl" and os.environ.get("PN_SIM_COOP", "") == "1"):
            # this GPU only simulates: with PN_SIM_COOP=1 the substep's local/global iterations run as one persistent kernel on (almost) every CU
            # (csrc/pn_sim.hip: k_substep_coop; 0.24 instead of 0.28 ms).  Opt-in: it has never run beside RCCL's kernels (one-GPU test box), and a
            # persistent kernel that cannot get all its CUs ends with a flag and an invalid state instead of a slower step.  Never beside renders:
            # its workgroups and those of the fused composite/compaction would wait for each other.  (A gloo group is the one-GPU dry run.)
            self.sim.enable_persistent()
        be = _HipBackend(self, lanes, depth, int(n_trips), W, H, sim_priority, sim_cus, copy_out, group if on else None,
                         (dist.get_global_rank(group, sim_owner) if (on and group is not None) else sim_owner), _probe_no_substep, _time_trips, copy_on,
                         render_kw=render_kw, distributed=on)
        self._pipe = FramePipeline(be, world=world, rank=rank, lanes=lanes, depth=depth, ahead=(lanes if sim_ahead is None and world == 1 else sim_ahead),
                                   sim_owner=sim_owner, dedicated_sim=dedicated_sim, copy_out=copy_out, on_retire=on_retire,
                                   force_collectives=bool(_force_collectives) and on, sim_on_lanes=sim_on_lanes)
        self.sim.force_hooks = (self._pipe.before_force, self._pipe.after_force) if self._pipe.sim_on_lanes else None
        self._pipe_backend = be
        self._pipe_form_epoch = self._net_form_epoch()
        self.model._in_flight = lambda pipe=self._pipe: sum(f is not None for f in pipe.pending)  # weight refreshes need a drained pipeline (network._net_handle)
        return self

    def capture_frame_parallel(self, lanes=2, n_trips=8, group=None, sim_owner=0, dedicated_sim=None, **kw):
        """Multi-GPU form (BASELINE.json configs[3]): every rank calls step_frame_parallel() once per GLOBAL frame."""
        return self.capture_pipelined(lanes=lanes, n_trips=n_trips, group=group, frame_parallel=True, sim_owner=sim_owner, dedicated_sim=dedicated_sim, **kw)

    @torch.no_grad()
    def step_pipelined(self, pose=None):
        self._check_net_form(self._pipe_form_epoch, "the pipeline")
        out = self._pipe.step(pose)
        self.frame = self._pipe.frame
        return out

    step_frame_parallel = step_pipelined

    @property
    def substeps_enqueued(self):
        """Simulator substeps enqueued so far in pipelined mode (= frames enqueued + sim_ahead once running)."""
        return self._pipe.substeps_enqueued

    @torch.no_grad()
    def drain_pipeline(self):
        """Retires every frame in flight (continuing any that ran out of trips) and returns them; the device is idle afterwards."""
        out = self._pipe.drain()
        torch.cuda.synchronize(self.device)
        self._check_persistent_substep()
        return out

    def wait_frame_copies(self):
        """Pipelined mode: host-waits for the device AND for the copier thread's frame copies (which are on no stream)."""
        torch.cuda.synchronize(self.device)
        be = getattr(self, "_pipe_backend", None)
        if be is not None and hasattr(be, "wait_copies"):
            be.wait_copies()

    @torch.no_grad()
    def verify_last_frame(self):
        """After drain_pipeline(): renders the LAST frame this rank's pipeline enqueued once more — a blocking render, launch by launch, from the
        integration-point state and pose its workspace still holds — and compares it bit for bit with what the pipeline delivered (the arrays in
        pinned host memory when the frames are copied out, else the workspace's device buffers).  Every form of the frame (captured graph, several
        frames in flight, fused launches, one lane per ray or windows) produces the same bits by construction; a difference means a race between the
        pipeline's streams, a frame copied before it was finished, or a broken kernel.  Returns dict(ok, frame, max_abs_diff)."""
        pipe, be, m = self._pipe, self._pipe_backend, self.model
        ws = pipe.last_ws
        if ws is None:
            return {"ok": False, "frame": None, "why": "no frame was enqueued"}
        torch.cuda.synchronize(self.device)
        res = be.result(ws)
        keep = (m.p_def, m.IP_F, m.IP_dF)
        try:
            m.p_def, m.IP_F, m.IP_dF = be.ip[ws]
            rays = get_rays(be.pose_dev[ws], self.intrinsics, be.H, be.W, -1)
            kw = dict(self.render_kwargs())
            kw.pop("ray_batch", None)
            if be.kw.get("ray_batch"):
                kw["ray_batch"] = be.kw["ray_batch"]   # batches keep their own trip schedules: part of what the frame IS, not of how it is launched
            with self._amp():
                out = m.render_deformed(rays["rays_o"], rays["rays_d"], staged=True, bg_color=None, perturb=False, **kw)
            torch.cuda.synchronize(self.device)
            worst, ok = 0.0, True
            for k in ("image", "depth_0"):
                want = out[k].reshape(-1).cpu().numpy()
                got = (res[k] if k in res else res["device"][k].cpu().numpy()).reshape(-1)
                ok = ok and bool(np.array_equal(want, got))
                worst = max(worst, float(np.abs(want - got).max()))
        finally:
            m.p_def, m.IP_F, m.IP_dF = keep
        return {"ok": ok, "frame": int(pipe.last_frame), "max_abs_diff": worst}

    def _check_persistent_substep(self):
        """A persistent substep whose workgroups could not all become resident ends with a flag instead of hanging (csrc/pn_sim.hip); its DOFs
        are garbage.  Checked where the host synchronises anyway."""
        if self.sim.persistent and self.sim.persistent_timed_out():
            raise RuntimeError("the persistent substep (pn_sim_stepforward_coop) timed out at a device-wide barrier: the simulator state is invalid "
                               "(another kernel held CUs for seconds, or two persistent substeps ran at once); use Simulator(persistent=False)")

    # ------------------------------------------------------------------ one frame split over the ranks (interactive latency)
    @torch.no_grad()
    def capture_tile_parallel(self, group=None, sim_owner=0, tile=8, W=None, H=None, _force_collectives=False):
        """Ray-tile-parallel form ("alternative for interactive latency"; frames.TileParallel): every rank renders an
        interleaved 1/world of the 8 x 8 pixel tiles of the SAME frame from the sim owner's broadcast DOF snapshot, and one all-gather
        (20 B x N / world per rank over RCCL) gives every rank the whole frame.  Unlike the frame-parallel pipeline, whose throughput is
        capped by the time-sequential simulator, this divides the render LATENCY of a frame by the rank count.  Launches are eager."""
        from .frames import TileParallel
        o, m, dev = self.opt, self.model, self.device
        W, H = W or o["W"], H or o["H"]
        ip = tuple(torch.empty((self.sim.n_IP, c), dtype=torch.float32, device=dev) for c in (3, 9, 27))
        self._tile_pose = torch.from_numpy(np.asarray(self.pose, np.float32)).unsqueeze(0).to(dev)

        def render_subset(idx):
            rays = get_rays(self._tile_pose, self.intrinsics, H, W, -1)
            sel = idx.clamp(min=0)
            self.sim.get_IP_info(out=ip)
            m.p_def, m.IP_F, m.IP_dF = ip
            with self._amp():
                out = m.render_deformed(rays["rays_o"][:, sel].contiguous(), rays["rays_d"][:, sel].contiguous(), bg_color=None, perturb=False, frame_slot=800,
                                        **self.render_kwargs())
            return torch.cat([out["image"].view(-1, 3), out["depth"].view(-1, 1), out["depth_0"].view(-1, 1)], dim=1)

        def set_dof(t):
            self.sim.dof.copy_(t)
        self._tile = TileParallel(W, H, render_subset, lambda: self.sim.dof, set_dof, self.sim.stepforward, sim_owner=sim_owner, group=group, tile=tile, device=dev,
                                  force_collectives=_force_collectives)
        self._tile_WH = (W, H)
        return self

    @torch.no_grad()
    def step_tile_parallel(self, pose=None):
        """One sim+render step; every rank gets the full frame: {'image' [1,H,W,3], 'depth' [1,H,W], 'depth_0' [1,H,W]} on the device."""
        if pose is not None:
            self._tile_pose.copy_(torch.from_numpy(np.asarray(pose, np.float32)).view(1, 4, 4).to(self.device))
        full = self._tile.step()
        W, H = self._tile_WH
        self.frame += 1
        return {"image": full[:, :3].reshape(1, H, W, 3), "depth": full[:, 3].reshape(1, H, W), "depth_0": full[:, 4].reshape(1, H, W)}

    # ------------------------------------------------------------------ a frame rendered in ray batches (BASELINE.json configs[4])
    def capture_staged(self, batch=None, **kw):
        """The frame rendered in ray batches of `batch` rays (opt max_ray_batch = 4096, get_opts.py:24; renderer.py:562-576's staging loop):
        every batch keeps its own trip schedule (n_step = max(min(N_b // n_alive_b, 8), 1), its own max_steps count), exactly as if the
        batches were rendered one after the other — but all batches advance inside the same launches (pn_render_opts.ray_batch: rays are
        independent, the alive list stays sorted by ray id, a batch is a contiguous run of it).  So the staged frame IS a pipelined frame:
        same graphs, same lanes, same D2H; `kw` goes to capture_pipelined, frames come back through step_pipelined() / drain_pipeline().
        (Rounds 1-2 replayed one captured launch chain per batch: ~10 000 launches per 800x800 frame, 20x slower than the frame in one piece.)"""
        rk = dict(kw.pop("render_kw", None) or {}, ray_batch=int(batch or self.opt.get("max_ray_batch", 4096)))  # this pipeline's renders only: self.opt stays as it was
        return self.capture_pipelined(render_kw=rk, **kw)

    # names of rounds 1-2 (one captured launch chain per batch), kept as aliases of the pipeline's calls
    def step_staged(self, pose=None):
        return self.step_pipelined(pose)

    def finish_staged(self):
        return self.drain_pipeline()

    def to_host(self, out):
        """The reference's device->host boundary (trainer.py:589-592)."""
        return {k: out[k][0].detach().cpu().numpy() for k in ("image", "depth", "depth_0")}


class _HipStream:
    def __init__(self, s):
        self.s = s

    def wait(self, ev):
        self.s.wait_event(ev.e)


class _HostCopyStream:
    """Stands where frames.FramePipeline expects the stream of the frame copies when they are done by the library's host-side copier
    (pn_copier, copy_on="host"): `wait(event)` names the event the next copy has to follow, `_HipBackend.copy_out` submits the copy behind it,
    an event `recorded` here carries the copy's ticket and can only be waited for by the host (which is all the pipeline does with it)."""

    def __init__(self, backend):
        self.backend, self.after, self.ticket = backend, None, None

    def wait(self, ev):
        self.after = ev


class _HipEvent:
    def __init__(self):
        self.e = torch.cuda.Event()
        self.copy = None  # (copier handle, ticket) when the event marks the end of a host-side frame copy

    def record(self, stream):
        if isinstance(stream, _HostCopyStream):
            self.copy = (stream.backend.copier, stream.ticket)
        else:
            self.copy = None
            self.e.record(stream.s)

    def host_wait(self):
        if self.copy is not None:
            from ._lib import check, lib
            check(lib().pn_copier_wait(self.copy[0], self.copy[1]), "copier_wait")
        else:
            self.e.synchronize()


class _HipBackend:
    """The device side of frames.FramePipeline on one MI355X: torch streams and events, the substep and one render per workspace captured
    as HIP graphs, RCCL broadcasts of the dof snapshots, D2H into pinned buffers."""

    def __init__(self, h, lanes, depth, n_trips, W, H, sim_priority, sim_cus, copy_out, group, src, probe_no_substep, time_trips=False, copy_on="host",
                 render_kw=None, distributed=False):
        # stream of the per-frame D2H.  gfx950 runs 4 hardware queues concurrently and time-slices beyond that: with 3 render lanes +
        # the simulator stream a copy stream of its own is a fifth busy queue (measured: 808 steps/s against 1016 with the copy on the frame's own
        # lane).  "copy": a stream of its own; "lane": the frame's render stream; "sim": the simulator stream; "host": no stream at all — the
        # library's copier thread waits for the render's event and hands the copy to the HSA runtime (SDMA), see csrc/pn_copier.hip
        self.copy_on = copy_on
        self.copier = None
        if copy_on == "host" and copy_out:
            import ctypes

            from ._lib import check, lib
            hc = ctypes.c_void_p()
            check(lib().pn_copier_create(ctypes.byref(hc)), "copier_create")
            self.copier = hc
        self.h, self.lanes, self.depth, self.trips, self.W, self.H, self.group, self.src = h, lanes, depth, n_trips, W, H, group, src
        dev, m, sim = h.device, h.model, h.sim
        self.continued = 0
        if sim_cus > 0:
            # compute-unit partition: the substep's chain of small dependent launches gets `sim_cus` CUs of its own (spread over the XCDs),
            # the render lanes the rest, so a substep never waits for a render wave to release registers
            self._streams = {"sim": h._cu_masked_stream(0, sim_cus, invert=False)}
            lane_streams = [h._cu_masked_stream(0, sim_cus, invert=True) for _ in range(lanes)]
        elif sim_cus < 0:
            # the converse: the render lanes stay off -sim_cus CUs, the simulator may use every CU — its launches always find those free, and whatever else is
            self._streams = {"sim": torch.cuda.Stream(dev, priority=sim_priority)}
            lane_streams = [h._cu_masked_stream(0, -sim_cus, invert=True) for _ in range(lanes)]
        else:
            self._streams = {"sim": torch.cuda.Stream(dev, priority=sim_priority)}
            lane_streams = [torch.cuda.Stream(dev) for _ in range(lanes)]
        for i, s in enumerate(lane_streams):
            self._streams[f"lane{i}"] = s
        self._streams["comm"], self._streams["copy"] = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
        sim.force_stream = self._streams["sim"]   # a force change is enqueued between two substeps of the simulator stream
        self.snap = {}
        n_ws, n_IP = lanes * depth, sim.n_IP
        kw = dict(h.render_kwargs(), async_trips=n_trips)
        kw.update(render_kw or {})  # options of THIS pipeline's renders (capture_staged: ray_batch)
        # several frames in flight: the first trip's march pass in its throughput form (pn_render_opts.throughput: one lane per ray, no speculative
        # evaluation; the same samples bit for bit, +8 % steps/s with three lanes) — with one lane the frame's own latency is what counts
        kw.setdefault("march_throughput", int(os.environ.get("PN_HARNESS_THROUGHPUT", "64")) if lanes > 1 else 0)   # (the environment variable: experiments)
        # ... and each frame's fused launch (pn_render_opts.fused_from) on half of the CUs: the launches of the frames in flight run side by side and the
        # other kernels find CUs with free LDS (pn_render_opts.fused_grid; three lanes on the chair: 1 721 -> 1 937 steps/s)
        if lanes > 1:
            cus = torch.cuda.get_device_properties(dev).multi_processor_count
            kw.setdefault("fused_grid", max(cus * 5 // 8 if lanes == 2 else cus // 2, 1))   # (two lanes, whole-frame launches: 128 / 160 / 192 workgroups: 1 717 / 1 776 / 1 689 steps/s)
        self.kw = kw
        if "fused_from" not in kw:
            # pn_render_opts.fused_from: the loop trips from the first one with n_step == 8 on (n_alive <= N / 8: it stays 8 for the rest of the frame) run
            # as ONE persistent launch (csrc/pn_trips_fused.h) — read off the trip records of a blocking frame from the current state and pose.  A later
            # frame that still has more rays alive at that trip is finished when it is retired (frames.FramePipeline: continue), only slower
            h.step(simulate=False, collect_stats=True, W=W, H=H)
            recs = m.trip_records(slot=0, max_trips=64)
            kw["fused_from"] = next((i for i, r in enumerate(recs) if i >= 1 and r[1] == 8), -1) if not kw.get("ray_batch") else -1
        if "fused_whole" not in kw:
            # pn_render_opts.fused_whole: the frame's first trip in that launch too (where at most N / 8 rays have anything to march: the blocking frame
            # below tries it and reports the trip at which the fused launch took over).  Measured on the chair: better with two frames in flight
            # (1 700-1 740 against 1 560 steps/s), worse with three (1 730 against 1 930) or one at a time — include/pienerf_hip.h
            whole = lanes == 2 and kw["fused_from"] == 1
            if whole:
                h.step(simulate=False, collect_stats=True, W=W, H=H, render_kw=dict(kw, fused_whole=True, fused_from=0, async_trips=0))
                whole = m.fused_clocks(slot=0)["first_trip"] == 0
            kw["fused_whole"] = bool(whole)
            if whole:
                kw["fused_from"] = 0
        if "fused_fold" not in kw:
            # pn_render_opts.fused_fold: the first trip's network, composite and compaction inside the fused launch (its march stays launches of its own): four
            # launches fewer on a lane's chain.  Where the blocking frame below finds it applicable (at most N / 8 rays with a sample on the first trip)
            # (measured on the chair, alternating runs on one box: three lanes 1 960 against 1 925 steps/s, --steps 20: 1 801 against 1 790; one frame at a
            # time the launch with the first trip's tiles in front of its rounds is longer than what it replaces, 0.884 against 0.859 ms per step)
            fold = lanes >= 3 and (not kw.get("fused_whole")) and kw["fused_from"] == 1
            if fold:
                h.step(simulate=False, collect_stats=True, W=W, H=H, render_kw=dict(kw, fused_fold=True, async_trips=0))
                fold = m.fused_clocks(slot=0)["mode"] == 2
            kw["fused_fold"] = bool(fold)
        if distributed:
            # every rank of a frame-parallel job runs the launch form rank `src` picked (each probed its own warm-up frame: the same state and pose, but
            # nothing guarantees the same answer at a threshold) — round-4 advisor; gated on the job being distributed, NOT on `group`: the default
            # process group is group=None (round-5 advisor: with `if group is not None` this never ran in a real job)
            from .frames import agree_on_launch_form
            agree_on_launch_form(kw, src=src, group=group, device=dev)
        pose0 = torch.from_numpy(np.asarray(h.pose, np.float32)).unsqueeze(0)
        self.pose_dev = [pose0.to(dev) for _ in range(n_ws)]
        self.pose_pin = [pose0.clone().pin_memory() for _ in range(n_ws)]
        self.pose_key = [pose0.numpy().tobytes()] * n_ws
        self.ip = [tuple(torch.empty((n_IP, c), dtype=torch.float32, device=dev) for c in (3, 9, 27)) for _ in range(n_ws)]
        keep = (sim.dof.clone(), sim.dof_vel.clone())
        main = torch.cuda.current_stream(dev)
        for ws in range(n_ws):  # warm-up of every workspace outside capture (creates the pn_frame workspaces, the fp16 tables, ...)
            s = self._streams[f"lane{ws // depth}"]
            s.wait_stream(main)
            with torch.cuda.stream(s):
                for _ in range(2):
                    sim.get_IP_info(out=self.ip[ws])
                    sim.stepforward()
                    m.p_def, m.IP_F, m.IP_dF = self.ip[ws]
                    rays = get_rays(self.pose_dev[ws], h.intrinsics, H, W, -1)
                    with h._amp():
                        m.render_deformed(rays["rays_o"], rays["rays_d"], staged=True, bg_color=None, perturb=False, **dict(kw, frame_slot=ws))
            torch.cuda.synchronize(dev)
            if ws == 0 and kw.get("march_throughput") and "march_throughput_trips" not in kw:
                # the throughput form for every leading trip that still has rays enough to fill the GPU with ONE lane per ray (the trex option set's
                # second trip: 246 k rays x 3 samples; the chair's: 41 k x 8, faster in the windows) — from this warm-up frame's trip records; the
                # samples do not depend on the form, so a scene that changes later only loses speed
                n_lpr = 1
                for rec in m.trip_records(slot=0)[1:]:
                    if rec[0] < THROUGHPUT_FORM_MIN_RAYS:
                        break
                    n_lpr += 1
                kw["march_throughput_trips"] = n_lpr
        # capture_error_mode="thread_local": with a process group alive, RCCL's watchdog thread queries events while we capture;
        # in the default "global" mode any HIP call from another thread invalidates the capture
        self.sim_graph = torch.cuda.CUDAGraph()
        with _capture(self.sim_graph, stream=self._streams["sim"]):
            if not probe_no_substep:
                sim.stepforward()
            else:
                sim.dof_vel.mul_(1.0)
        self.graph, self.rays, self.out, self.host, self.packed = [], [], [], [], []
        N = W * H
        for ws in range(n_ws):
            s = self._streams[f"lane{ws // depth}"]
            m.p_def, m.IP_F, m.IP_dF = self.ip[ws]  # the render graph of this workspace reads its own IP buffers
            if time_trips:  # measurement (bench.py): time stamps around each trip's march / network launches become nodes of the captured graph
                m.march_counters(2, slot=ws)
            # image | depth | depth_0 packed in one device buffer -> ONE device-to-host copy per frame (20 B per pixel, trainer.py:589-592)
            pk = torch.empty(N * 5, dtype=torch.float32, device=dev)
            ob = {"image": pk[:3 * N].view(N, 3), "depth": pk[3 * N:4 * N], "depth_0": pk[4 * N:], "weights_sum": torch.empty(N, dtype=torch.float32, device=dev)}
            g = torch.cuda.CUDAGraph()
            with _capture(g, stream=s):
                rays = get_rays(self.pose_dev[ws], h.intrinsics, H, W, -1)
                with h._amp():
                    out = m.render_deformed(rays["rays_o"], rays["rays_d"], staged=True, bg_color=None, perturb=False, out_buffers=ob, **dict(kw, frame_slot=ws))
            self.graph.append(g)
            # every tensor the graphs touch stays referenced: a tensor freed after capture goes back to the graph's memory pool
            self.rays.append(rays)
            self.out.append(out)
            self.packed.append(pk)
            # two pinned sets per workspace, used alternately: the arrays handed out when a frame is retired stay valid until the workspace has
            # been through another whole frame (the next frame's D2H goes into the other set)
            self.host.append([torch.empty(N * 5, dtype=torch.float32).pin_memory() for _ in range(2)] if copy_out else None)
        self.host_gen = [0] * n_ws
        torch.cuda.synchronize(dev)
        if self.copier is not None:
            # the first SDMA copy into a pinned buffer maps it for the engine (milliseconds): once per buffer here, not inside the pipeline's first frames
            import ctypes

            from ._lib import check, lib
            for ws in range(n_ws):
                for hb in self.host[ws]:
                    t = ctypes.c_uint64()
                    check(lib().pn_copier_submit(self.copier, ctypes.c_void_p(hb.data_ptr()), ctypes.c_void_p(self.packed[ws].data_ptr()),
                                                 hb.numel() * hb.element_size(), None, ctypes.byref(t)), "copier_submit")
                    check(lib().pn_copier_wait(self.copier, t.value), "copier_wait")
        sim.dof.copy_(keep[0])      # warm-up advanced the simulator; capture itself executes nothing
        sim.dof_vel.copy_(keep[1])
        sim.reset_warm_start()      # ... and its SVD warm start: a replay from here has the bits of a simulator that never warmed up
        torch.cuda.synchronize(dev)

    # ---- streams / events
    def stream(self, name):
        if name == "copy" and self.copier is not None:
            return _HostCopyStream(self)
        return _HipStream(self._streams[name])

    def __del__(self):
        if getattr(self, "copier", None) is not None:
            try:
                from ._lib import lib
                lib().pn_copier_destroy(self.copier)
            except Exception:  # noqa: BLE001 — interpreter shutdown: the import machinery is gone, the process is about to end anyway
                pass
            self.copier = None

    def event(self):
        return _HipEvent()

    def _snap(self, slot):
        if slot not in self.snap:
            self.snap[slot] = torch.empty_like(self.h.sim.dof)
        return self.snap[slot]

    # ---- device operations
    def snapshot(self, s, slot):
        with torch.cuda.stream(s.s):
            self._snap(slot).copy_(self.h.sim.dof)

    def substep(self, s):
        with torch.cuda.stream(s.s):
            self.sim_graph.replay()

    def broadcast(self, s, slot, src):
        import torch.distributed as dist
        with torch.cuda.stream(s.s):
            dist.broadcast(self._snap(slot), src=self.src, group=self.group)

    def render(self, s, frame, ws, slot, pose):
        with torch.cuda.stream(s.s):
            # the camera of this frame (trainer.py:541): the given pose, else the harness's current one.  Every workspace keeps its own device
            # copy (the graph reads it), so it is uploaded whenever it differs from what THIS workspace last rendered — staged in pinned memory,
            # in stream order before the graph reads it
            p = np.ascontiguousarray(self.h.pose if pose is None else pose, dtype=np.float32).reshape(1, 4, 4)
            key = p.tobytes()
            if self.pose_key[ws] != key:
                self.pose_key[ws] = key
                self.pose_pin[ws].copy_(torch.from_numpy(p))
                self.pose_dev[ws].copy_(self.pose_pin[ws], non_blocking=True)
            self.h.sim.get_IP_info(dof=self._snap(slot), out=self.ip[ws])
            self.graph[ws].replay()

    def copy_out(self, s, ws, same_buffer=False):
        if not same_buffer:
            self.host_gen[ws] ^= 1
        dst, src = self.host[ws][self.host_gen[ws]], self.packed[ws]
        if isinstance(s, _HostCopyStream):
            import ctypes

            from ._lib import check, lib
            t = ctypes.c_uint64()
            after = ctypes.c_void_p(s.after.e.cuda_event) if s.after is not None else None
            check(lib().pn_copier_submit(self.copier, ctypes.c_void_p(dst.data_ptr()), ctypes.c_void_p(src.data_ptr()), src.numel() * src.element_size(), after,
                                         ctypes.byref(t)), "copier_submit")
            s.ticket = t.value
            self.last_ticket = t.value
            return
        with torch.cuda.stream(s.s):
            dst.copy_(src, non_blocking=True)

    def wait_copies(self):
        """[host] Blocks until every frame copy submitted so far has landed in host memory.  The copier thread's SDMA copies are on no stream:
        torch.cuda.synchronize() does not wait for them (bench.py stops its clock behind this)."""
        if self.copier is not None and getattr(self, "last_ticket", None) is not None:
            from ._lib import check, lib
            check(lib().pn_copier_wait(self.copier, self.last_ticket), "copier_wait")

    # ---- host side of a retired workspace
    def complete(self, ws):
        return self.h.model.render_status(synchronize=False, slot=ws)["alive_at_exit"] == 0

    def finish(self, ws):
        """The captured trips were not enough for this frame: keep going on the workspace's lane until no ray is alive (blocking), then copy
        the finished frame out again."""
        h, m = self.h, self.h.model
        s = self._streams[f"lane{ws // self.depth}"]
        with torch.cuda.stream(s):
            with h._amp():
                m.render_continue(ws, self.rays[ws]["rays_o"], self.rays[ws]["rays_d"], self.out[ws], bg_color=None, **self.kw)
            if self.host[ws] is not None:
                self.copy_out(_HipStream(s), ws, same_buffer=True)
        s.synchronize()
        self.continued += 1
        if self.continued == 8:
            # the launch form baked into the graphs (fused_from / fused_whole / fused_fold, the trip count) came from ONE warm-up frame; a scene or pose
            # that has moved away from it makes every frame end here, blocking — correct, only slow (round-4 advisor): say so once
            import warnings
            warnings.warn("8 frames of this pipeline had rays alive behind their captured trips and were finished by blocking renders: the launch form picked "
                          "at capture no longer fits the scene / pose — capture_pipelined() again from the current state", RuntimeWarning, stacklevel=2)

    def result(self, ws):
        o = self.out[ws]
        H, W, N = self.H, self.W, self.H * self.W
        dev = {"image": o["image"].view(-1, H, W, 3), "depth": o["depth"].view(-1, H, W), "depth_0": o["depth_0"].view(-1, H, W)}
        if self.host[ws] is None:
            return {"device": dev}
        hb = self.host[ws][self.host_gen[ws]].numpy()
        return {"image": hb[:3 * N].reshape(H, W, 3), "depth": hb[3 * N:4 * N].reshape(H, W), "depth_0": hb[4 * N:].reshape(H, W), "device": dev}
