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
import numpy as np
import torch

from . import scene
from .nerf.network import NeRFNetwork
from .nerf.utils import get_rays
from .simulator.solver import Simulator


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

    def _amp(self):
        """Trainer.test_gui renders inside ``torch.cuda.amp.autocast(enabled=self.fp16)`` (trainer.py:561).  main_gui.py builds its Trainer
        without fp16 (fp32 render, the default here); opt['fp16'] = True is main_train.py's Trainer(fp16=opt.fp16) and BASELINE configs[4]:
        fp16 hash tables + half Linear layers (renderer.rund_cuda reads the autocast state)."""
        return torch.autocast("cuda", dtype=torch.float16, enabled=bool(self.opt.get("fp16", False)))

    def render_kwargs(self):
        """**vars(opt) as the reference passes it (trainer.py:318); renderer reads these by name."""
        return dict(self.opt)

    @torch.no_grad()
    def step(self, pose=None, intrinsics=None, W=None, H=None, simulate=True, collect_stats=False, fused=True):
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
        with torch.cuda.graph(self._graph, capture_error_mode="thread_local"):
            self._graph_out = self._step_body(n_trips, W, H)
        self.sim.dof.copy_(keep[0])       # warm-up advanced the simulator; capture itself executes nothing
        self.sim.dof_vel.copy_(keep[1])
        self._graph_trips = n_trips
        self._graph_done = None
        return self

    @torch.no_grad()
    def step_graph(self, pose=None):
        """Replays the captured step.  Outputs are static tensors (overwritten by the next replay)."""
        if getattr(self, "_graph", None) is None:
            self.capture()
        self._check_previous_graph_frame()
        if pose is not None:
            self._graph_pose.copy_(torch.from_numpy(np.asarray(pose, np.float32)).unsqueeze(0))
        self._graph.replay()
        self._graph_done = torch.cuda.Event()
        self._graph_done.record(torch.cuda.current_stream(self.device))
        self.frame += 1
        return self._graph_out

    def _check_previous_graph_frame(self):
        if getattr(self, "_graph_done", None) is not None:
            self._graph_done.synchronize()  # normally long complete: the host only ever runs one frame ahead
            st = self.model.render_status(synchronize=False)
            self._graph_done = None
            if st["alive_at_exit"] > 0:
                raise RuntimeError(f"captured step ran {self._graph_trips} render trips but {st['alive_at_exit']} rays were still alive: "
                                   "re-capture with more trips (harness.capture(n_trips=...))")

    # ------------------------------------------------------------------ several frames in flight on one GPU
    @torch.no_grad()
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
    def capture_pipelined(self, lanes=2, n_trips=8, W=None, H=None, sim_ahead=None, sim_priority=0, sim_cus=0, _probe_no_substep=False, _extra_slots=0):
        """Throughput mode: `lanes` renders in flight on their own streams, the simulator running `sim_ahead` frames ahead.

        A render is a chain of short latency-bound launches whose tails leave most of the 256 CUs idle, and the substep is a
        serial chain of ~30 small launches; neither fills the GPU alone, so frames are software-pipelined:
          * simulator stream: for frame g, `snap[g % slots] <- dof` (33 KB: the state frame g is rendered from), then the
            captured substep graph.  It depends on nothing but itself, so it never waits for a render;
          * lane stream l = f % lanes: update_F(snap[f % slots]) into the lane's IP buffers, then the lane's captured render
            graph, after the snapshot's event.
        The host enqueues frame f only after frame f - lanes completed (its outputs and buffers are reused), and keeps the
        simulator `sim_ahead` (default: lanes) frames ahead of that.  Every frame is still rendered from the state before
        its own substep (trainer.py:300-318); `self.sim.dof` is `sim_ahead + 1` substeps ahead of the last enqueued frame, and a
        force set with update_force() acts from the next substep that is enqueued."""
        o, m, dev = self.opt, self.model, self.device
        W, H = W or o["W"], H or o["H"]
        ahead = lanes if sim_ahead is None else int(sim_ahead)
        slots = lanes + ahead + 1 + int(_extra_slots)
        self._pipe = dict(lanes=lanes, ahead=ahead, slots=slots, W=W, H=H, trips=n_trips, ren_graph=[], out=[], stream=[], done=[], pending=[], ip=[],
                          keepalive=[], sim_next=0)
        p = self._pipe
        self._graph_pose = torch.from_numpy(np.asarray(self.pose, np.float32)).unsqueeze(0).to(dev)
        if sim_cus > 0:
            # compute-unit partition: the substep's ~30 small dependent launches get `sim_cus` CUs of their own (spread over the
            # XCDs), the render lanes the rest, so a substep never waits for a render wave to release registers
            p["sim_stream"] = self._cu_masked_stream(0, sim_cus, invert=False)
            streams = [self._cu_masked_stream(0, sim_cus, invert=True) for _ in range(lanes)]
        else:
            p["sim_stream"] = torch.cuda.Stream(dev, priority=sim_priority)
            streams = [torch.cuda.Stream(dev) for _ in range(lanes)]
        self.sim.force_stream = p["sim_stream"]  # a force change is enqueued between two substeps of the simulator stream
        p["snap"] = [torch.empty_like(self.sim.dof) for _ in range(slots)]
        p["snap_ready"] = [torch.cuda.Event() for _ in range(slots)]
        keep = (self.sim.dof.clone(), self.sim.dof_vel.clone())
        kw = self.render_kwargs()
        kw["async_trips"] = n_trips
        main = torch.cuda.current_stream(dev)
        n_IP = self.sim.n_IP
        for lane in range(lanes):  # warm-up of every lane outside capture (creates the per-lane frame workspaces)
            s = streams[lane]
            s.wait_stream(main)
            ip = tuple(torch.empty((n_IP, c), dtype=torch.float32, device=dev) for c in (3, 9, 27))
            p["ip"].append(ip)
            with torch.cuda.stream(s):
                for _ in range(2):
                    self.sim.get_IP_info(out=ip)
                    self.sim.stepforward()
                    m.p_def, m.IP_F, m.IP_dF = ip
                    rays = get_rays(self._graph_pose, self.intrinsics, H, W, -1)
                    with self._amp():
                        m.render_deformed(rays["rays_o"], rays["rays_d"], staged=True, bg_color=None, perturb=False, **dict(kw, frame_slot=lane))
            torch.cuda.synchronize(dev)
        # capture_error_mode="thread_local": with a process group alive, RCCL's watchdog thread queries events while we capture;
        # in the default "global" mode any HIP call from another thread invalidates the capture
        gs = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gs, stream=p["sim_stream"], capture_error_mode="thread_local"):
            if not _probe_no_substep:
                self.sim.stepforward()
            else:
                self.sim.dof_vel.mul_(1.0)
        p["sim_graph"] = gs
        for lane in range(lanes):
            kw_l = dict(kw, frame_slot=lane)
            s = streams[lane]
            m.p_def, m.IP_F, m.IP_dF = p["ip"][lane]  # the render graph of this lane reads the lane's own IP buffers
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=s, capture_error_mode="thread_local"):
                rays = get_rays(self._graph_pose, self.intrinsics, H, W, -1)
                with self._amp():
                    out = m.render_deformed(rays["rays_o"], rays["rays_d"], staged=True, bg_color=None, perturb=False, **kw_l)
            p["ren_graph"].append(gr)
            # every tensor the graphs touch stays referenced: a tensor freed after capture goes back to the graph's memory pool
            # and may be handed out again
            p["keepalive"].append((rays, out))
            p["out"].append({"image": out["image"].reshape(-1, H, W, 3), "depth": out["depth"].reshape(-1, H, W),
                             "depth_0": out["depth_0"].reshape(-1, H, W)})
            p["stream"].append(s)
            ev = torch.cuda.Event()
            ev.record(main)
            p["done"].append(ev)
            p.setdefault("done_spare", []).append(torch.cuda.Event())
            p["pending"].append(False)
        torch.cuda.synchronize(dev)
        self.sim.dof.copy_(keep[0])
        self.sim.dof_vel.copy_(keep[1])
        torch.cuda.synchronize(dev)
        return self

    @torch.no_grad()
    def step_pipelined(self):
        """Enqueues frame `self.frame` on its lane and returns that lane's (static) outputs; they are complete once the lane's
        `done` event has fired (synchronize() / the next use of the lane)."""
        p = self._pipe
        lane = self.frame % p["lanes"]
        sim_s, ren_s = p["sim_stream"], p["stream"][lane]
        # simulator: substeps up to frame + ahead.  Snapshot slot g % slots was last read by frame g - slots = frame - lanes - 1,
        # which the host saw complete at the end of the previous call, so the simulator stream waits for nothing.
        with torch.cuda.stream(sim_s):
            while p["sim_next"] <= self.frame + p["ahead"]:
                slot = p["sim_next"] % p["slots"]
                p["snap"][slot].copy_(self.sim.dof)
                p["snap_ready"][slot].record(sim_s)
                p["sim_graph"].replay()
                p["sim_next"] += 1
        # this frame goes onto its lane BEFORE the host waits for the lane's previous frame (stream order keeps them apart on
        # the GPU): the ~0.15 ms the host needs to wake up and enqueue ~100 launches is then hidden behind that frame's tail
        # instead of leaving the lane empty
        slot = self.frame % p["slots"]
        prev_done, had_prev = p["done"][lane], p["pending"][lane]
        p["done"][lane], p["done_spare"][lane] = p["done_spare"][lane], prev_done
        ren_s.wait_event(p["snap_ready"][slot])
        with torch.cuda.stream(ren_s):
            self.sim.get_IP_info(dof=p["snap"][slot], out=p["ip"][lane])
            p["ren_graph"][lane].replay()
            p["done"][lane].record(ren_s)
        p["pending"][lane] = True
        self.frame += 1
        if had_prev:  # the host runs at most `lanes` completed-or-running frames plus one queued frame per lane ahead
            prev_done.synchronize()
            # the status words of the previous frame are read microseconds after it completed; the frame just enqueued overwrites
            # them only at ITS end, a render later
            st = self.model.render_status(synchronize=False, slot=lane)
            if st["alive_at_exit"] > 0:
                raise RuntimeError(f"pipelined step ran {p['trips']} render trips but {st['alive_at_exit']} rays were still alive")
        return p["out"][lane]

    # ------------------------------------------------------------------ frame-parallel over the GPUs of a node
    @torch.no_grad()
    def capture_frame_parallel(self, lanes=3, n_trips=8, group=None, sim_owner=0, dedicated_sim=None):
        """Multi-GPU form of capture_pipelined (BASELINE.json configs[3]): every rank calls step_frame_parallel()
        once per GLOBAL frame f.  The sim owner advances the simulator (running ahead on dof snapshots, exactly as on one GPU)
        and every snapshot is broadcast over `group` on a communication stream of its own — <= 82 KB per frame, the only
        exchange; rank frames.frame_owner(f) renders frame f on one of its lanes from its copy of snapshot f (round-robin over all
        ranks; with `dedicated_sim` — default from 3 ranks on — over every rank but the owner, which then only simulates).  Neither the
        substeps nor the broadcasts ever wait for a render, so the ranks' renders overlap freely; the job is bounded by the
        owner's substep rate (the simulator is time-sequential and does not shard)."""
        import torch.distributed as dist
        on = dist.is_available() and dist.is_initialized()
        world = dist.get_world_size(group) if on else 1
        rank = dist.get_rank(group) if on else 0
        self.capture_pipelined(lanes=lanes, n_trips=n_trips, sim_ahead=world * lanes, _extra_slots=world * lanes)
        p = self._pipe
        S = p["slots"]
        from .frames import dedicated_sim_default
        dedicated = dedicated_sim_default(world) if dedicated_sim is None else bool(dedicated_sim and world > 1)
        p.update(world=world, rank=rank, owner=sim_owner, group=group, bc_next=0, comm=torch.cuda.Stream(self.device), dedicated=dedicated,
                 renderers=(world - 1 if dedicated else world), my_frames=0,
                 src=(dist.get_global_rank(group, sim_owner) if (on and group is not None) else sim_owner),
                 bc_done=[torch.cuda.Event() for _ in range(S)], ip_done=[torch.cuda.Event() for _ in range(S)],
                 bc_used=[False] * S, ip_used=[False] * S)
        return self

    @torch.no_grad()
    def step_frame_parallel(self):
        """One global frame.  Returns the lane's (static) outputs on the rank that renders it, None elsewhere."""
        import torch.distributed as dist
        p = self._pipe
        from .frames import frame_owner
        f, world, rank, S = self.frame, p["world"], p["rank"], p["slots"]
        mine = frame_owner(f, world, p["owner"], p["dedicated"]) == rank
        lane = p["my_frames"] % p["lanes"]
        sim_s, comm = p["sim_stream"], p["comm"]
        if rank == p["owner"]:
            with torch.cuda.stream(sim_s):
                while p["sim_next"] <= f + p["ahead"]:
                    slot = p["sim_next"] % S
                    if p["bc_used"][slot]:
                        sim_s.wait_event(p["bc_done"][slot])   # the slot's previous snapshot has been sent ...
                    if p["ip_used"][slot]:
                        sim_s.wait_event(p["ip_done"][slot])   # ... and consumed by this rank's own render
                    p["snap"][slot].copy_(self.sim.dof)
                    p["snap_ready"][slot].record(sim_s)
                    p["sim_graph"].replay()
                    p["sim_next"] += 1
        if world > 1:
            with torch.cuda.stream(comm):
                while p["bc_next"] <= f + p["ahead"]:  # same order on every rank
                    slot = p["bc_next"] % S
                    if rank == p["owner"]:
                        comm.wait_event(p["snap_ready"][slot])
                    elif p["ip_used"][slot]:
                        comm.wait_event(p["ip_done"][slot])    # this rank's render has read the slot's previous snapshot
                    dist.broadcast(p["snap"][slot], src=p["src"], group=p["group"])
                    p["bc_done"][slot].record(comm)
                    p["bc_used"][slot] = True
                    p["bc_next"] += 1
        out = None
        if mine:
            slot = f % S
            ren_s = p["stream"][lane]
            prev_done, had_prev = p["done"][lane], p["pending"][lane]
            p["done"][lane], p["done_spare"][lane] = p["done_spare"][lane], prev_done
            ren_s.wait_event(p["bc_done"][slot] if world > 1 else p["snap_ready"][slot])
            with torch.cuda.stream(ren_s):  # enqueued before the host waits for the lane's previous frame (see step_pipelined)
                self.sim.get_IP_info(dof=p["snap"][slot], out=p["ip"][lane])
                p["ip_done"][slot].record(ren_s)
                p["ip_used"][slot] = True
                p["ren_graph"][lane].replay()
                p["done"][lane].record(ren_s)
            p["pending"][lane] = True
            p["my_frames"] += 1
            out = p["out"][lane]
            if had_prev:
                prev_done.synchronize()
                st = self.model.render_status(synchronize=False, slot=lane)
                if st["alive_at_exit"] > 0:
                    raise RuntimeError(f"frame-parallel step ran {p['trips']} render trips but {st['alive_at_exit']} rays were still alive")
        elif p["dedicated"] and rank == p["owner"] and world > 1:
            # a rank that never renders has nothing that paces its host: wait until this frame's snapshot has been delivered, so that
            # the owner stays at most `ahead` frames in front of the slowest receiver instead of enqueueing the whole job at once
            p["bc_done"][f % S].synchronize()
        self.frame += 1
        return out

    @property
    def substeps_enqueued(self):
        """Simulator substeps enqueued so far in pipelined mode (= frames rendered + sim_ahead + 1 once running)."""
        return self._pipe["sim_next"]

    def drain_pipeline(self):
        """Waits for every frame in flight and verifies each lane's last render completed."""
        p = self._pipe
        torch.cuda.synchronize(self.device)
        for lane in range(p["lanes"]):
            if p["pending"][lane]:
                st = self.model.render_status(synchronize=False, slot=lane)
                p["pending"][lane] = False
                if st["alive_at_exit"] > 0:
                    raise RuntimeError(f"pipelined step ran {p['trips']} render trips but {st['alive_at_exit']} rays were still alive")

    def to_host(self, out):
        """The reference's device->host boundary (trainer.py:589-592)."""
        return {k: out[k][0].detach().cpu().numpy() for k in ("image", "depth", "depth_0")}
