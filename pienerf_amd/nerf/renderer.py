"""Deformed-space renderer with the reference's surface (nerf/renderer.py:74-113, 587-599, 755-907).

``NeRFRenderer.render_deformed(rays_o, rays_d, **vars(opt))`` is what ``Trainer.test_step`` calls
(nerf/trainer.py:316-318); the caller sets ``p_ori, p_def, IP_F, IP_dF, IP_dx`` on the model
(main_gui.py:52-56, trainer.py:303-306).  Two implementations of ``rund_cuda`` are provided:

  * ``rund_cuda``      — one C call (pn_render_deformed): the whole loop runs on the GPU with a device-side
                         (n_alive, n_step) record; no per-trip host synchronisation.
  * ``rund_cuda_ops``  — the reference's Python loop verbatim in structure, on the drop-in ops
                         (near_far_from_aabb / get_pnts_in_grids / march_rays_quadratic_bending / network /
                         composite_rays / compaction).  Used by the parity tests and as documentation of semantics.
"""
import ctypes as C
import math

import torch
import torch.nn as nn

from .. import raymarching
from .._lib import RenderOpts, check, lib, ptr, require_gpu, stream_ptr
from .utils import get_pnts_in_grids


class NeRFRenderer(nn.Module):
    def __init__(self, bound=1, cuda_ray=False, density_scale=1, min_near=0.2, density_thresh=0.01, bg_radius=-1):
        super().__init__()
        self.bound = bound
        self.cascade = 1 + math.ceil(math.log2(bound))
        self.grid_size = 128
        self.density_scale = density_scale
        self.min_near = min_near
        self.density_thresh = density_thresh
        self.bg_radius = bg_radius
        aabb = torch.FloatTensor([-bound, -bound, -bound, bound, bound, bound])
        self.register_buffer("aabb_train", aabb)
        self.register_buffer("aabb_infer", aabb.clone())
        self.cuda_ray = cuda_ray
        if cuda_ray:  # same state-dict keys as renderer.py:94-111
            self.register_buffer("density_grid", torch.zeros([self.cascade, self.grid_size ** 3]))
            self.register_buffer("density_bitfield", torch.zeros(self.cascade * self.grid_size ** 3 // 8, dtype=torch.uint8))
            self.register_buffer("step_counter", torch.zeros(16, 2, dtype=torch.int32))
            self.mean_density = 0
            self.iter_density = 0
            self.mean_count = 0
            self.local_step = 0
        self._frame = None
        self._frames = {}
        self.last_stats = None

    def forward(self, x, d):
        raise NotImplementedError()

    def _net_handle(self, half=False):
        raise NotImplementedError()

    @staticmethod
    def _autocast_half():
        return False

    # ------------------------------------------------------------------ entry point (renderer.py:587-599)
    def render_deformed(self, rays_o, rays_d, staged=False, max_ray_batch=4096, **kwargs):
        """renderer.py:587-599.  `staged` / `max_ray_batch` are accepted and unused exactly as in the reference, whose render_deformed calls
        rund_cuda on the whole ray set whatever they say (only the non-cuda_ray `render` stages, :562-576)."""
        if not self.cuda_ray:
            raise RuntimeError("render_deformed: only the cuda_ray path (main_gui.py / main_render.py with -O) is implemented")
        return self.rund_cuda(rays_o, rays_d, **kwargs)

    def _ip_state(self, device):
        return [t.to(device=device, dtype=torch.float32).contiguous() for t in (self.p_def, self.p_ori, self.IP_F, self.IP_dF)]

    def _frame_handle(self, N, n_vtx, hgs, slot=0, cells=None):
        """Per-frame workspace (pn_frame).  `slot` selects one of several independent workspaces so that frames can be in
        flight concurrently on different streams (harness.capture_pipelined)."""
        # spatial-hash capacity: the IP cloud lives inside the simulation box (2.04*bound per side, solver.py:24-32); x2 margin per axis
        side = int(math.ceil(2.2 * float(self.bound) / hgs)) + 2
        cells = side ** 3 if cells is None else int(cells)
        key = (N, n_vtx, cells)
        cur = self._frames.get(slot)
        if cur is None or cur[1] != key:
            if cur is not None:
                lib().pn_frame_destroy(cur[0])
            h = C.c_void_p()
            check(lib().pn_frame_create(C.byref(h), N, n_vtx, cells), "frame_create")
            self._frames[slot] = (h, key)
        if slot == 0:
            self._frame = self._frames[0][0]
        return self._frames[slot][0]

    # ------------------------------------------------------------------ fused loop
    def _deformed_opts(self, dt_gamma, bg_scalar, max_steps, T_thresh, kwargs, n_rays=0):
        o = RenderOpts()
        o.max_iter_num = int(kwargs.get("max_iter_num"))
        o.hash_grid_size = float(kwargs.get("hash_grid_size"))
        o.num_seek_IP = int(kwargs.get("num_seek_IP"))
        o.IP_dx = float(self.IP_dx)
        o.cut = int(bool(kwargs.get("cut")))
        cb = kwargs.get("cut_bounds") or [0.0] * 6
        for i in range(6):
            o.cut_bounds[i] = float(cb[i])
        o.bound = float(kwargs.get("bound", self.bound))
        o.min_near = float(self.min_near)
        o.dt_gamma = float(dt_gamma)
        o.max_steps = int(max_steps)
        o.T_thresh = float(T_thresh)
        o.cascade = int(self.cascade)
        o.grid_size = int(self.grid_size)
        o.density_scale = float(self.density_scale)
        o.bg_color = float(bg_scalar)
        o.fp16 = int(self._autocast_half())  # Trainer.test_gui renders under autocast(enabled=self.fp16) (trainer.py:561)
        o.reuse_tables = int(bool(kwargs.get("reuse_tables")))  # extension: later ray batches of the same frame keep the first batch's tables
        o.ray_batch = int(kwargs.get("ray_batch") or 0)  # extension: per-batch trip schedules inside one set of launches (pn_render_opts.ray_batch)
        o.throughput = int(kwargs.get("march_throughput") or 0)  # extension: one lane per ray in the first trip's pass 1 (pn_render_opts.throughput)
        o.throughput_trips = int(kwargs.get("march_throughput_trips") or 0)  # ... and in this many leading trips (0: the first only)
        # extension: walk a whole image's rays in 16 x 4 pixel tiles (pn_render_opts.ray_tile_w; results do not depend on it).  The reference hands
        # **vars(opt) to the renderer (trainer.py:318), so W / H arrive by name; only a ray set of exactly W * H rays is taken for the image
        tw = kwargs.get("ray_tile_w")
        if tw is None and n_rays and kwargs.get("W") and kwargs.get("H") and int(kwargs["W"]) * int(kwargs["H"]) == n_rays:
            tw = kwargs["W"]
        o.ray_tile_w = int(tw or 0)
        o.fused_from = int(kwargs.get("fused_from") or 0)  # extension: first loop trip of the one-launch form (pn_render_opts.fused_from; 0: trip 1, < 0: never)
        o.fused_grid = int(kwargs.get("fused_grid") or 0)  # extension: workgroups of that launch (pn_render_opts.fused_grid; 0: one per CU)
        o.fused_fold = int(bool(kwargs.get("fused_fold")))  # extension: ... the first trip's network + composite + compaction folded in (pn_render_opts.fused_fold)
        o.fused_whole = int(bool(kwargs.get("fused_whole")))  # extension: ... the frame's first trip included, where it applies (pn_render_opts.fused_whole)
        return o

    def rund_cuda(self, rays_o, rays_d, dt_gamma=0, bg_color=None, perturb=False, max_steps=1024, T_thresh=1e-2, **kwargs):
        if perturb:
            return self.rund_cuda_ops(rays_o, rays_d, dt_gamma, bg_color, perturb, max_steps, T_thresh, **kwargs)
        prefix = rays_o.shape[:-1]
        rays_o = rays_o.to(torch.float32).contiguous().view(-1, 3)
        rays_d = rays_d.to(torch.float32).contiguous().view(-1, 3)
        require_gpu(rays_o, rays_d)
        N, device = rays_o.shape[0], rays_o.device
        if self.bg_radius > 0:
            raise RuntimeError("background model (bg_radius > 0) is not on the simulate-and-render path")
        if bg_color is None:
            bg_color = 1
        # a tensor background ([3] or [N,3], e.g. the GUI's bg_color tensor, gui.py:590): the driver composites over 0 and the blend
        # image + (1 - weights_sum) * bg (renderer.py:896) is applied afterwards with the same two roundings
        bg_tensor = bg_color.to(device=device, dtype=torch.float32) if torch.is_tensor(bg_color) else None
        p_def, p_ori, F_IP, dF_IP = self._ip_state(device)
        assert p_def.shape == p_ori.shape and p_ori.shape[0] > 0  # renderer.py:816-817
        n_vtx = p_ori.shape[0]
        o = self._deformed_opts(dt_gamma, 0.0 if bg_tensor is not None else bg_color, max_steps, T_thresh, kwargs, n_rays=N)
        ob = kwargs.get("out_buffers")  # extension: caller-owned outputs (the frame pipeline packs image | depth | depth_0 into one buffer -> one D2H)
        if ob is not None:
            image, depth, depth_0, weights_sum = ob["image"], ob["depth"], ob["depth_0"], ob["weights_sum"]
            assert image.shape == (N, 3) and depth.shape == (N,) and all(t.is_contiguous() and t.dtype == torch.float32 for t in (image, depth, depth_0, weights_sum))
        else:
            image = torch.empty(N, 3, dtype=torch.float32, device=device)
            depth = torch.empty(N, dtype=torch.float32, device=device)
            depth_0 = torch.empty(N, dtype=torch.float32, device=device)
            weights_sum = torch.empty(N, dtype=torch.float32, device=device)
        async_trips = int(kwargs.get("async_trips") or 0)
        frame, net = self._frame_handle(N, n_vtx, o.hash_grid_size, int(kwargs.get("frame_slot") or 0)), self._net_handle(half=bool(o.fp16))
        if async_trips > 0:
            # non-blocking: a fixed number of trips, no host synchronisation (legal under HIP-graph capture); completion is
            # checked later with render_status(), a frame that ran out of trips is finished with render_continue()
            check(lib().pn_render_deformed_async(frame, net, C.byref(o), ptr(rays_o), ptr(rays_d), N, ptr(p_def), ptr(p_ori), ptr(F_IP), ptr(dF_IP),
                                                 n_vtx, ptr(self.density_bitfield), ptr(image), ptr(depth), ptr(depth_0), ptr(weights_sum),
                                                 async_trips, stream_ptr()), "render_deformed_async")
        else:
            stats = (C.c_int64 * 5)() if kwargs.get("collect_stats") else None
            check(lib().pn_render_deformed(frame, net, C.byref(o), ptr(rays_o), ptr(rays_d), N, ptr(p_def), ptr(p_ori), ptr(F_IP), ptr(dF_IP), n_vtx,
                                           ptr(self.density_bitfield), ptr(image), ptr(depth), ptr(depth_0), ptr(weights_sum), stats, stream_ptr()),
                  "render_deformed")
            if stats is not None:
                self._set_stats(stats)
        if bg_tensor is not None:
            image = image + (1 - weights_sum).unsqueeze(-1) * bg_tensor
        return {"depth": depth.view(*prefix), "image": image.view(*prefix, 3), "depth_0": depth_0.view(*prefix), "weights_sum": weights_sum}

    def render_continue(self, slot, rays_o, rays_d, out, n_trips=0, dt_gamma=0, bg_color=None, max_steps=1024, T_thresh=1e-2, static=False, **kwargs):
        """Finishes the frame last rendered on workspace `slot` with a fixed trip count that turned out too small (render_status reports
        rays alive at exit): more trips of the same loop — n_trips of them, or (0) until no ray is alive — and the epilogue again, into the
        SAME output tensors `out` (the dict the render returned).  The reference's loop has no trip limit but max_steps (renderer.py:836-891);
        this is how the captured, fixed-length forms keep that semantics.  Blocking when n_trips == 0."""
        rays_o = rays_o.to(torch.float32).contiguous().view(-1, 3)
        rays_d = rays_d.to(torch.float32).contiguous().view(-1, 3)
        N = rays_o.shape[0]
        if static:
            o = RenderOpts()
            o.max_iter_num, o.hash_grid_size, o.num_seek_IP, o.IP_dx, o.cut = 1, 1.0, 1, 0.0, 0
            o.bound, o.min_near, o.dt_gamma, o.max_steps, o.T_thresh = float(self.bound), float(self.min_near), float(dt_gamma), int(max_steps), float(T_thresh)
            o.cascade, o.grid_size, o.density_scale, o.bg_color = int(self.cascade), int(self.grid_size), float(self.density_scale), float(1 if bg_color is None else bg_color)
            o.fp16 = int(self._autocast_half())
        else:
            o = self._deformed_opts(dt_gamma, 1 if bg_color is None else bg_color, max_steps, T_thresh, kwargs, n_rays=N)
        image, depth, ws = out["image"].view(-1, 3), out["depth"].view(-1), out["weights_sum"].view(-1)
        depth_0 = out["depth_0"].view(-1) if "depth_0" in out else torch.empty_like(depth)
        assert image.is_contiguous() and image.shape[0] == N
        stats = (C.c_int64 * 5)() if int(n_trips) == 0 else None
        check(lib().pn_render_continue(self._frames[slot][0], self._net_handle(half=bool(o.fp16)), C.byref(o), ptr(rays_o), ptr(rays_d), N,
                                       ptr(self.density_bitfield), ptr(image), ptr(depth), ptr(depth_0), ptr(ws), stats, int(n_trips), int(bool(static)),
                                       stream_ptr()), "render_continue")
        if stats is not None:
            self._set_stats(stats)
        return out

    def _set_stats(self, stats):
        self.last_stats = dict(trips=int(stats[0]), samples=int(stats[1]), err=int(stats[2]), alive_at_exit=int(stats[3]), unfinished=int(stats[4]))
        if stats[2]:
            raise RuntimeError(f"render_deformed: device error flags {int(stats[2])} (1: sample cell outside the spatial hash, "
                               "2: IP outside it, 4: spatial-hash capacity exceeded, 8: candidate-list capacity exceeded, "
                               "16: the fused composite/compaction gave up waiting for an earlier chunk, 32: the fused launch found its kernarg segment laid "
                               "out otherwise than csrc/pn_trips_fused.h: fused_karg_fresh assumes)")

    def march_counters(self, enable, read=False, slot=0):
        """Measurement hook on workspace `slot`: 1 = device-side work counters of the march kernel (iterations, candidates, warps, samples);
        2 = HIP events around each trip's march and network launches (see trip_times)."""
        out = (C.c_uint64 * 4)() if read else None
        check(lib().pn_frame_march_counters(self._frames[slot][0], int(enable), out, stream_ptr()), "march_counters")
        return None if out is None else dict(iterations=int(out[0]), candidates=int(out[1]), warps=int(out[2]), samples=int(out[3]))

    def fused_clocks(self, slot=0, reset=False):
        """Phase clocks of the fused later-trips launches on `slot` (march_counters(4)): dict of cycles summed over waves + the trip the last render
        switched to the fused launch at (-1: it did not)."""
        out, first = (C.c_uint64 * 16)(), C.c_int(-1)
        check(lib().pn_frame_fused_clocks(self._frames[slot][0], out, C.byref(first), int(bool(reset)), stream_ptr()), "fused_clocks")
        return dict(refill=int(out[0]), march=int(out[1]), windows=int(out[2]), network=int(out[3]), composite=int(out[4]), wave_rounds=int(out[5]),
                    waves=int(out[6]), lifetime_ticks=int(out[7]), max_rounds=int(out[8]), max_lifetime_ticks=int(out[9]), first_trip=int(first.value),
                    # whole-frame form (fused_from = 0): the first trip's one-lane march, its 64-lane windows, its network, its composite + hand-over,
                    # the wait at the workgroup barrier behind it
                    a_march=int(out[10]), a_windows=int(out[11]), a_network=int(out[12]), a_composite=int(out[13]), a_barrier=int(out[14]),
                    mode=int(out[15]))   # 0: later trips only, 1: whole frame, 2: first trip's network / composite / compaction folded in

    def trip_records(self, slot=0, max_trips=16):
        """Diagnostics: [(n_alive, n_step, step_base, n_samples, n_emitted, n_tail)] per trip of the last render on `slot`."""
        rec, tail = (C.c_int * (5 * max_trips))(), (C.c_int * max_trips)()
        check(lib().pn_frame_trip_records(self._frames[slot][0], rec, tail, max_trips, stream_ptr()), "trip_records")
        return [tuple(rec[5 * i:5 * i + 5]) + (tail[i],) for i in range(max_trips) if rec[5 * i] > 0]

    def trip_times(self, slot=0):
        """Per-trip (march_ms, network_ms) of the last render on `slot` made while event timing was enabled (HIP events on the launch stream;
        for a captured render: the records of the last replay)."""
        a, b, n = (C.c_float * 64)(), (C.c_float * 64)(), C.c_int(0)
        check(lib().pn_frame_trip_times(self._frames[slot][0], a, b, 64, C.byref(n), stream_ptr()), "trip_times")
        return [float(a[i]) for i in range(n.value)], [float(b[i]) for i in range(n.value)]

    def render_status(self, synchronize=True, slot=0):
        """Outcome of the last render on frame workspace `slot`: dict(trips, samples, err, alive_at_exit).
        After an async render, alive_at_exit > 0 means the enqueued trips were not enough."""
        stats = (C.c_int64 * 5)()
        check(lib().pn_render_status(self._frames[slot][0], stats, int(bool(synchronize)), stream_ptr()), "render_status")
        self._set_stats(stats)
        return self.last_stats

    # ------------------------------------------------------------------ static render + training state
    def reset_extra_state(self):
        """renderer.py:125-135."""
        if not self.cuda_ray:
            return
        self.density_grid.zero_()
        self.mean_density = 0
        self.iter_density = 0
        self.step_counter.zero_()
        self.mean_count = 0
        self.local_step = 0

    def run_cuda(self, rays_o, rays_d, dt_gamma=0, bg_color=None, perturb=False, force_all_rays=False, max_steps=1024, T_thresh=1e-2, **kwargs):
        """NeRFRenderer.run_cuda (nerf/renderer.py:267-387), the static / undeformed render.

        train():  march_rays_train -> network (differentiable) -> composite_rays_train, the step counter feeding ``mean_count`` (:293-341).
        eval():   ONE call of the frame driver pn_render_static — near / far from aabb_infer, then trips of march / network / composite /
                  stable compaction driven by a device-side trip record, with one 16-byte read-back per batch of 8 trips (the reference
                  synchronises the host on every trip, :351-380).  Per-ray background colours and ``perturb`` take the op-by-op loop
                  ``run_cuda_ops`` (same kernels, host-driven)."""
        if not self.training and not perturb and not torch.is_tensor(bg_color):
            return self._run_static_fused(rays_o, rays_d, dt_gamma, 1 if bg_color is None else bg_color, max_steps, T_thresh, **kwargs)
        return self.run_cuda_ops(rays_o, rays_d, dt_gamma, bg_color, perturb, force_all_rays, max_steps, T_thresh, **kwargs)

    def _run_static_fused(self, rays_o, rays_d, dt_gamma, bg_color, max_steps, T_thresh, **kwargs):
        prefix = rays_o.shape[:-1]
        rays_o = rays_o.to(torch.float32).contiguous().view(-1, 3)
        rays_d = rays_d.to(torch.float32).contiguous().view(-1, 3)
        require_gpu(rays_o, rays_d)
        if self.bg_radius > 0:
            raise RuntimeError("background model (bg_radius > 0) is not built")
        N, device = rays_o.shape[0], rays_o.device
        o = RenderOpts()
        o.max_iter_num, o.hash_grid_size, o.num_seek_IP, o.IP_dx, o.cut = 1, 1.0, 1, 0.0, 0
        o.bound, o.min_near, o.dt_gamma, o.max_steps, o.T_thresh = float(self.bound), float(self.min_near), float(dt_gamma), int(max_steps), float(T_thresh)
        o.cascade, o.grid_size, o.density_scale, o.bg_color = int(self.cascade), int(self.grid_size), float(self.density_scale), float(bg_color)
        o.fp16 = int(self._autocast_half())
        aabb = (C.c_float * 6)(*self._aabb_infer_host())
        image, depth, depth_0, ws = (torch.empty(N, 3, dtype=torch.float32, device=device), torch.empty(N, dtype=torch.float32, device=device),
                                     torch.empty(N, dtype=torch.float32, device=device), torch.empty(N, dtype=torch.float32, device=device))
        frame = self._frame_handle(N, 1, 1.0, int(kwargs.get("frame_slot") or 0), cells=1)
        async_trips = int(kwargs.get("async_trips") or 0)
        stats = (C.c_int64 * 5)() if not async_trips else None
        check(lib().pn_render_static(frame, self._net_handle(half=bool(o.fp16)), C.byref(o), ptr(rays_o), ptr(rays_d), N, aabb, ptr(self.density_bitfield),
                                     ptr(image), ptr(depth), ptr(depth_0), ptr(ws), stats, async_trips, stream_ptr()), "render_static")
        if stats is not None:
            self._set_stats(stats)
        return {"depth": depth.view(*prefix), "image": image.view(*prefix, 3), "weights_sum": ws}

    def _aabb_infer_host(self):
        """aabb_infer as six Python floats, read back from the device only when the buffer changed (never inside a stream capture)."""
        key = (self.aabb_infer.data_ptr(), self.aabb_infer._version)
        if getattr(self, "_aabb_cache", (None, None))[0] != key:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("aabb_infer changed: render one frame outside stream capture first")
            self._aabb_cache = (key, [float(v) for v in self.aabb_infer.tolist()])
        return self._aabb_cache[1]

    def run_cuda_ops(self, rays_o, rays_d, dt_gamma=0, bg_color=None, perturb=False, force_all_rays=False, max_steps=1024, T_thresh=1e-2, **kwargs):
        """run_cuda op by op on the drop-in ops (the training branch; the eval branch with perturb / tensor backgrounds; parity tests)."""
        shape = rays_o.shape[:-1]
        o3, d3 = rays_o.contiguous().view(-1, 3), rays_d.contiguous().view(-1, 3)
        nears, fars = raymarching.near_far_from_aabb(o3, d3, self.aabb_train if self.training else self.aabb_infer, self.min_near)
        if self.bg_radius > 0:
            raise RuntimeError("background model (bg_radius > 0) is not built")
        bg = 1 if bg_color is None else bg_color

        def shade(xyzs, dirs):
            sig, rgb = self(xyzs, dirs)
            return self.density_scale * sig, rgb

        if self.training:
            slot = self.step_counter[self.local_step % 16]
            slot.zero_()
            self.local_step += 1
            xyzs, dirs, deltas, rays = raymarching.march_rays_train(o3, d3, self.bound, self.density_bitfield, self.cascade, self.grid_size, nears, fars, slot,
                                                                    self.mean_count, perturb, 128, force_all_rays, dt_gamma, max_steps)
            ws, depth, image = raymarching.composite_rays_train(*shade(xyzs, dirs), deltas, rays, T_thresh)
            self.last_stats = dict(trips=1, samples=int(xyzs.shape[0]), err=0, alive_at_exit=0)
        else:
            with torch.no_grad():
                n_rays, dev = o3.shape[0], o3.device
                ws, depth, image = (torch.zeros(n_rays, device=dev), torch.zeros(n_rays, device=dev), torch.zeros(n_rays, 3, device=dev))
                alive, t_now = torch.arange(n_rays, dtype=torch.int32, device=dev), nears.clone()
                done_steps = trips = samples = 0
                while done_steps < max_steps and alive.shape[0] > 0:
                    k = alive.shape[0]
                    per_ray = max(min(n_rays // k, 8), 1)
                    xyzs, dirs, deltas = raymarching.march_rays(k, per_ray, alive, t_now, o3, d3, self.bound, self.density_bitfield, self.cascade,
                                                                self.grid_size, nears, fars, 128, perturb if done_steps == 0 else False, dt_gamma, max_steps)
                    raymarching.composite_rays(k, per_ray, alive, t_now, *shade(xyzs, dirs), deltas, ws, depth, image, T_thresh)
                    alive = raymarching.compact_rays(alive)
                    samples += int((deltas[:, 0] != 0).sum())
                    done_steps += per_ray
                    trips += 1
                self.last_stats = dict(trips=trips, samples=samples, err=0, alive_at_exit=int(alive.shape[0]))
        image = image + (1 - ws).unsqueeze(-1) * bg
        depth = torch.clamp(depth - nears, min=0) / (fars - nears)
        return {"depth": depth.view(*shape), "image": image.view(*shape, 3), "weights_sum": ws}

    def render(self, rays_o, rays_d, staged=False, max_ray_batch=4096, **kwargs):
        """renderer.py:552-585 (cuda_ray never stages)."""
        if not self.cuda_ray:
            raise RuntimeError("render: only the cuda_ray path is built")
        return self.run_cuda(rays_o, rays_d, **kwargs)

    # ------------------------------------------------------------------ density-grid state on the device (pn_grid_state.hip)
    @torch.no_grad()
    def mark_untrained_grid(self, poses, intrinsic, S=64):
        """renderer.py:390-452: cells of the density grid that no training camera sees get density -1 (never sampled, never updated).
        One launch — a lane per (cascade, cell) loops over the poses — instead of the reference's five nested Python loops; returns the
        number of unseen cells (the reference prints it).  ``S`` (the reference's block / pose-batch size) has no role here."""
        if not self.cuda_ray:
            return
        cams = torch.as_tensor(poses).to(device=self.density_grid.device, dtype=torch.float32).contiguous().view(-1, 4, 4)
        require_gpu(cams, self.density_grid)
        fx, fy, cx, cy = (float(v) for v in intrinsic)
        n_unseen = torch.zeros(1, dtype=torch.int32, device=cams.device)
        check(lib().pn_mark_untrained_grid(ptr(cams), cams.shape[0], fx, fy, cx, cy, self.cascade, self.grid_size, float(self.bound),
                                           ptr(self.density_grid), ptr(n_unseen), stream_ptr()), "mark_untrained_grid")
        return int(n_unseen.item())

    @torch.no_grad()
    def update_extra_state(self, decay=0.95, S=128):
        """renderer.py:454-549.  Density grid: sigma at one jittered sample per cell — every cell for the first 16 calls, afterwards H^3/4
        uniform cells + H^3/4 draws from the occupied cells per cascade — EMA-max into the grid, mean of the non-negative part, bitfield at
        min(mean, density_thresh); then ``mean_count`` from the step counters.  Everything runs on the device in a handful of launches
        (pn_density_cells_* -> pn_nerf_sigma -> pn_density_grid_update); the only host read-back is the final mean (the reference's
        ``.item()``), and the occupied-cell list of the partial sweep is compacted on the device instead of through torch.nonzero."""
        if not self.cuda_ray:
            return
        grid = self.density_grid
        dev, H, n_cas = grid.device, self.grid_size, self.cascade
        require_gpu(grid)
        cells = H ** 3
        net, half = self._net_handle(half=self._autocast_half()), int(self._autocast_half())
        tmp = torch.empty_like(grid)
        if self.iter_density < 16:
            pts = torch.empty(n_cas * cells, 3, dtype=torch.float32, device=dev)
            check(lib().pn_density_cells_full(n_cas, H, float(self.bound), ptr(torch.rand(n_cas * cells, 3, device=dev)), ptr(pts), stream_ptr()), "density_cells_full")
            check(lib().pn_nerf_sigma(net, ptr(pts), n_cas * cells, float(self.density_scale), ptr(tmp), half, stream_ptr()), "nerf_sigma")
        else:
            n = cells // 4
            scratch = torch.empty(int(lib().pn_density_partial_scratch_ints(H)), dtype=torch.int32, device=dev)
            pts = torch.empty(2 * n, 3, dtype=torch.float32, device=dev)
            idx = torch.empty(2 * n, dtype=torch.int32, device=dev)
            sig = torch.empty(2 * n, dtype=torch.float32, device=dev)
            for cas in range(n_cas):
                draws = torch.randint(0, H, (n, 3), device=dev, dtype=torch.int32)
                check(lib().pn_density_cells_partial(cas, H, float(self.bound), n, ptr(draws), ptr(torch.rand(n, device=dev)), ptr(torch.rand(2 * n, 3, device=dev)),
                                                     ptr(grid[cas]), ptr(tmp[cas]), ptr(scratch), ptr(idx), ptr(pts), stream_ptr()), "density_cells_partial")
                check(lib().pn_nerf_sigma(net, ptr(pts), 2 * n, float(self.density_scale), ptr(sig), half, stream_ptr()), "nerf_sigma")
                check(lib().pn_density_scatter(2 * n, ptr(idx), ptr(sig), ptr(tmp[cas]), stream_ptr()), "density_scatter")
        partial = torch.empty((n_cas * cells + 255) // 256, dtype=torch.float64, device=dev)
        mean_thresh = torch.empty(2, dtype=torch.float32, device=dev)
        check(lib().pn_density_grid_update(n_cas * cells, ptr(grid), ptr(tmp), float(decay), float(self.density_thresh), ptr(self.density_bitfield),
                                           ptr(partial), ptr(mean_thresh), stream_ptr()), "density_grid_update")
        self.mean_density = float(mean_thresh[0].item())
        self.iter_density += 1
        counted = min(16, self.local_step)
        if counted > 0:  # :545-547
            self.mean_count = int(self.step_counter[:counted, 0].sum().item() / counted)
        self.local_step = 0

    # ------------------------------------------------------------------ op-by-op loop (reference structure, renderer.py:755-907)
    def rund_cuda_ops(self, rays_o, rays_d, dt_gamma=0, bg_color=None, perturb=False, max_steps=1024, T_thresh=1e-2, **kwargs):
        dtype = torch.float32
        prefix = rays_o.shape[:-1]
        rays_o = rays_o.contiguous().view(-1, 3)
        rays_d = rays_d.contiguous().view(-1, 3)
        N, device = rays_o.shape[0], rays_o.device
        max_iter_num, hgs, bound = kwargs.get("max_iter_num"), kwargs.get("hash_grid_size"), kwargs.get("bound")
        cut = kwargs.get("cut")
        cut_bounds = torch.tensor(kwargs.get("cut_bounds") or [0.0] * 6, dtype=dtype, device=device)
        p_def, p_ori, F_IP, dF_IP = self._ip_state(device)
        bmin, bmax = p_def.min(axis=0).values, p_def.max(axis=0).values
        if cut:
            bmin = -bound * torch.ones(3, dtype=dtype, device=device)
            bmax = bound * torch.ones(3, dtype=dtype, device=device)
        marg = 1e-3
        bbmin = bmin - marg * torch.ones(3, dtype=dtype, device=device)
        bbmax = bmax + marg * torch.ones(3, dtype=dtype, device=device)
        resolution = torch.ceil((bbmax - bbmin) / hgs).to(torch.int32)
        aabb = torch.cat((bbmin, bbmax), dim=0)
        nears, fars = raymarching.near_far_from_aabb(rays_o, rays_d, aabb, self.min_near)
        if bg_color is None:
            bg_color = 1
        weights_sum = torch.zeros(N, dtype=dtype, device=device)
        depth = torch.zeros(N, dtype=dtype, device=device)
        image = torch.zeros(N, 3, dtype=dtype, device=device)
        num_seek_IP = kwargs.get("num_seek_IP")
        n_vtx = p_ori.shape[0]
        n_grid = int(resolution[2] * resolution[1] * resolution[0])
        assert p_def.shape == p_ori.shape and n_vtx > 0
        pig_cnt, pig_bgn, pig_idx = get_pnts_in_grids(n_vtx, n_grid, p_def, bbmin, bbmax, hgs, resolution)
        rays_alive = torch.arange(N, dtype=torch.int32, device=device)
        rays_t = nears.clone()
        step, trips, samples = 0, 0, 0
        while step < max_steps:
            n_alive = rays_alive.shape[0]
            if n_alive <= 0:
                break
            n_step = max(min(N // n_alive, 8), 1)
            xyzs, dirs, deltas = raymarching.march_rays_quadratic_bending(
                pig_cnt, pig_bgn, pig_idx, n_vtx, n_grid, p_def, p_ori, F_IP, dF_IP, max_iter_num, bbmin, bbmax, hgs, resolution, num_seek_IP,
                self.IP_dx, cut, cut_bounds, n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, self.bound, self.density_bitfield, self.cascade,
                self.grid_size, nears, fars, 128, perturb if step == 0 else False, dt_gamma, max_steps)
            sigmas, rgbs = self(xyzs, dirs)
            sigmas = self.density_scale * sigmas
            raymarching.composite_rays(n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image, T_thresh)
            rays_alive = raymarching.compact_rays(rays_alive)  # == rays_alive[rays_alive >= 0]
            samples += int((deltas[:, 0] != 0).sum())
            step += n_step
            trips += 1
        self.last_stats = dict(trips=trips, samples=samples, err=0, alive_at_exit=int(rays_alive.shape[0]))
        depth_0 = depth
        image = image + (1 - weights_sum).unsqueeze(-1) * bg_color
        depth = torch.clamp(depth - nears, min=0) / (fars - nears)
        return {"depth": depth.view(*prefix), "image": image.view(*prefix, 3), "depth_0": depth_0.view(*prefix), "weights_sum": weights_sum}
