"""Frame pipeline of the simulate-and-render path: several frames in flight on one GPU, and frame-parallel over the GPUs of a node
(BASELINE.json configs[1] / configs[3]).

The simulator is time-sequential and tiny (<= 10 290 fp64 DOFs); a render is a chain of latency-bound launches that leaves most of the
chip idle; rendering is embarrassingly parallel over frames.  So the step is software-pipelined:

  * a *simulator stream* (on the sim owner) runs nothing but `snapshot[g % S] <- dof` + one substep, frame after frame, `ahead` frames in
    front of the renders; frame g is rendered from snapshot g = the state BEFORE substep g (trainer.py:300-318);
  * with more than one rank, every snapshot (<= 82 KB) is broadcast on a *communication stream* (RCCL; every peer is one xGMI hop) — the only
    per-frame exchange; the checkpoint is broadcast once (``broadcast_tensors``);
  * frame f is rendered by rank ``frame_owner(f)`` on render *lane* (stream) ``k % lanes`` and workspace ``(k // lanes) % depth`` of that
    lane, k = the rank's own frame counter: update_F(snapshot f) -> the captured render graph -> D2H of image / depth / depth_0 into pinned
    buffers on a copy stream (trainer.py:589-592).  A workspace is *retired* — host-waited, checked for rays still alive (then continued
    with more trips, renderer.py:836-891), handed out — right before it is reused, `lanes * depth` of the rank's frames later, by which time
    it has long completed: the host never waits for the GPU in steady state and never leaves a lane empty while it enqueues.

``FramePipeline`` is that schedule and nothing else: every device action goes through a small *backend* (streams, events, and the five
operations snapshot / substep / broadcast / render / copy-out).  ``pienerf_amd.harness`` supplies the HIP backend (torch streams, HIP
graphs, RCCL); ``SimulatedBackend`` below executes the same enqueue sequence on CPU tensors with FIFO "streams" run in a randomised but
dependency-respecting order and gloo broadcasts — what tests/test_frames_gloo.py uses to check, for 1-4 ranks, both placements and many
more frames than snapshot slots, that no slot is overwritten before its readers ran and that every frame sees exactly its own state.
"""
import random

import torch
import torch.distributed as dist


def agree_on_launch_form(kw, src=0, group=None, device=None):
    """Every rank of a frame-parallel job captures the launch form rank `src` picked: kw["fused_from" / "fused_whole" / "fused_fold"] are overwritten
    in place with src's (one 3-int broadcast; group=None is the default process group).  Each rank probes its own warm-up frame — same state and pose,
    but nothing guarantees the same answer at a threshold, and ranks running different launch forms would still render the same bits but not the same
    schedule the owner sized its lanes for."""
    import torch
    import torch.distributed as dist
    on_gpu = dist.get_backend(group) == "nccl"
    pick = torch.tensor([int(kw["fused_from"]), int(bool(kw["fused_whole"])), int(bool(kw["fused_fold"]))], dtype=torch.int64,
                        device=device if on_gpu else "cpu")
    dist.broadcast(pick, src=src, group=group)
    kw["fused_from"], kw["fused_whole"], kw["fused_fold"] = int(pick[0]), bool(pick[1]), bool(pick[2])
    return kw


def dedicated_sim_default(world_size):
    """Whether the sim owner should only simulate.  The job is bounded by the owner's substep rate (the simulator is time-sequential);
    a substep that shares its GPU with renders runs slower than alone, so from 3 ranks on — where the other
    ranks can absorb the owner's share of the frames — the owner renders nothing and the frames go round-robin over the rest."""
    return world_size >= 3


def frame_owner(frame, world_size, sim_owner=0, dedicated_sim=False):
    """Rank that renders `frame`: round-robin over all ranks, or over every rank but the sim owner."""
    if not dedicated_sim or world_size == 1:
        return frame % world_size
    k = frame % (world_size - 1)
    return k if k < sim_owner else k + 1


def broadcast_tensors(tensors, src=0, group=None):
    """One-off state broadcast (checkpoint: embeddings, MLP weights, density_bitfield; SURVEY.md §5 'Checkpoint / resume')."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    for t in tensors:
        dist.broadcast(t, src=src, group=group)


def tile_partition(W, H, world, tile=8):
    """Ray-tile-parallel split of ONE frame ("alternative for interactive latency"): the image is cut into tile x tile pixel
    blocks, block b (row-major) goes to rank b % world, so that every rank gets an interleaved 1/world of the object and of the background.
    Returns [world] int64 tensors of flat pixel (= ray) indices, each padded with -1 to the common length (all_gather needs equal sizes)."""
    ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    block = (ys // tile) * ((W + tile - 1) // tile) + (xs // tile)
    owner = (block % world).reshape(-1)
    parts = [torch.nonzero(owner == r).reshape(-1) for r in range(world)]
    n = max(p.numel() for p in parts)
    return [torch.cat([p, torch.full((n - p.numel(),), -1, dtype=torch.int64)]) for p in parts]


class TileParallel:
    """One frame rendered by all ranks together: rank r renders the rays ``tile_partition(...)[r]`` and an all-gather hands every rank the
    whole frame — latency scales down with the rank count, where the frame-parallel pipeline only scales throughput.  The sim owner broadcasts
    the DOF snapshot (<= 82 KB) first so that every rank renders the same state.

    render_subset(indices [n] int64, -1 = padding) -> float32 tensor [n, channels] on the communication device (image | depth | depth_0 = 5 channels)
    get_dof() / set_dof(t): the owner's state / installing the received state;  sim_step(): one substep on the owner."""

    def __init__(self, W, H, render_subset, get_dof, set_dof, sim_step, sim_owner=0, group=None, tile=8, device="cpu", force_collectives=False):
        on = dist.is_available() and dist.is_initialized()
        # force_collectives: a world of ONE rank still runs the broadcast and the all-gather (tests: RCCL itself on a one-GPU box)
        self.collectives = on and (dist.get_world_size(group) > 1 or bool(force_collectives))
        self.world = dist.get_world_size(group) if on else 1
        self.rank = dist.get_rank(group) if on else 0
        self.W, self.H, self.group, self.owner = W, H, group, sim_owner
        self.src = dist.get_global_rank(group, sim_owner) if (on and group is not None) else sim_owner
        self.parts = [p.to(device) for p in tile_partition(W, H, self.world, tile)]
        self.render_subset, self.get_dof, self.set_dof, self.sim_step = render_subset, get_dof, set_dof, sim_step
        self._recv = None
        self._gather = None

    def step(self):
        """Renders the current state (the state BEFORE this step's substep, trainer.py:300-318) and advances the simulator.  Returns the full
        frame [H*W, channels] on every rank."""
        if self.collectives:
            buf = self.get_dof() if self.rank == self.owner else (self._recv if self._recv is not None else torch.empty_like(self.get_dof()))
            self._recv = None if self.rank == self.owner else buf
            dist.broadcast(buf, src=self.src, group=self.group)
            if self.rank != self.owner:
                self.set_dof(buf)
        mine = self.render_subset(self.parts[self.rank]).contiguous()
        if self.collectives:
            if self._gather is None or self._gather.shape[1:] != mine.shape or self._gather.dtype != mine.dtype:
                self._gather = torch.empty((self.world,) + tuple(mine.shape), dtype=mine.dtype, device=mine.device)
            dist.all_gather_into_tensor(self._gather.view(-1, mine.shape[1]), mine, group=self.group)   # RCCL: ONE all-gather of 20 B x N / world per rank
            gathered = list(self._gather.unbind(0))
        else:
            gathered = [mine]
        full = torch.empty(self.W * self.H, mine.shape[1], dtype=mine.dtype, device=mine.device)
        for part, data in zip(self.parts, gathered):
            ok = part >= 0
            full[part[ok]] = data[ok]
        if self.rank == self.owner:
            self.sim_step()
        return full


class FramePipeline:
    """The enqueue schedule of the pipelined / frame-parallel step.  One instance per rank; every rank calls ``step()`` once per GLOBAL frame.

    backend interface (all calls only ENQUEUE work, except the three marked host):
        stream(name) -> s         named FIFO of device work: 'sim', 'comm', 'lane0'..., 'copy'
        event() -> e              e.record(s): marks a point of stream s;  s.wait(e): s does not pass until that point has run
        e.host_wait()             [host] blocks until the point has run
        snapshot(s, slot)         snap[slot] <- dof               substep(s)          one Simulator.stepforward
        broadcast(s, slot, src)   collective on snap[slot]
        render(s, frame, ws, slot, pose)   update_F(snap[slot]) + the render of workspace `ws` = (lane, sub)
        copy_out(s, ws)           D2H of the workspace's outputs
        complete(ws) -> bool      [host] did the render finish inside its trips?      finish(ws): [host] continue it until no ray is alive
        result(ws) -> object      [host] what step() hands back for a retired frame
    """

    def __init__(self, backend, world=1, rank=0, lanes=2, depth=2, ahead=None, sim_owner=0, dedicated_sim=None, copy_out=True, on_retire=None, force_collectives=False,
                 sim_on_lanes=False):
        self.b, self.world, self.rank, self.lanes, self.depth, self.owner = backend, int(world), int(rank), int(lanes), int(depth), int(sim_owner)
        self.dedicated = dedicated_sim_default(self.world) if dedicated_sim is None else bool(dedicated_sim and self.world > 1)
        self.ahead = self.world * self.lanes * self.depth if ahead is None else int(ahead)
        self.slots = self.ahead + self.world * self.lanes * self.depth + 1   # snapshot ring: reuse is guarded by events, the size only avoids stalls
        self.copy_out = copy_out
        # the snapshot broadcasts are skipped with a single rank — unless forced (tests: a world of ONE RCCL rank still runs every collective of the
        # N-rank schedule on the communication stream, beside the graph replays, so that the first N-GPU run is a measurement, not a first execution)
        self.collectives = self.world > 1 or bool(force_collectives)
        self.on_retire = on_retire   # called as on_retire(frame, result) the moment a frame is complete, BEFORE its workspace is reused
        b = backend
        self.s_sim, self.s_comm, self.s_copy = b.stream("sim"), b.stream("comm"), b.stream("copy")
        self.s_lane = [b.stream(f"lane{i}") for i in range(self.lanes)]
        S = self.slots
        self.snap_ready = [b.event() for _ in range(S)]      # owner: snapshot written (sim stream)
        self.bc_done = [b.event() for _ in range(S)]         # broadcast of the slot finished (comm stream)
        self.ip_done = [b.event() for _ in range(S)]         # this rank's render has consumed the slot (lane stream)
        self.bc_used, self.ip_used = [False] * S, [False] * S
        n_ws = self.lanes * self.depth
        self.render_done = [b.event() for _ in range(n_ws)]
        self.out_ready = [b.event() for _ in range(n_ws)]
        self.pending = [None] * n_ws                          # global frame index occupying the workspace
        self.frame = 0            # next global frame
        self.my_frames = 0        # frames this rank has rendered
        self.sim_next = 0         # next snapshot / substep the owner enqueues
        self.bc_next = 0          # next broadcast this rank enqueues
        self.retired = []         # (frame, result) of workspaces retired by the last step()
        self.last_ws, self.last_frame = None, None
        # sim_on_lanes (one rank): no simulator stream — substep g rides on render lane g % lanes, in front of that lane's next frame, chained to substep
        # g - 1 by an event.  The part runs four hardware queues side by side: the queue the simulator does not take is a fourth render lane.
        self.sim_on_lanes = bool(sim_on_lanes) and self.world == 1
        self.sub_done = [b.event() for _ in range(self.lanes + 2)] if self.sim_on_lanes else []
        self.force_done, self.force_pending = (b.event() if self.sim_on_lanes else None), False

    # a force change between two substeps (Simulator.update_force on the backend's 'sim' stream, which then carries nothing else): behind the last substep
    # enqueued, in front of the next one
    def before_force(self):
        if self.sim_on_lanes and self.sim_next > 0:
            self.s_sim.wait(self.sub_done[(self.sim_next - 1) % len(self.sub_done)])

    def after_force(self):
        if self.sim_on_lanes:
            self.force_done.record(self.s_sim)
            self.force_pending = True

    # ------------------------------------------------------------------ per-frame
    def _ws(self, k):
        lane = k % self.lanes
        return lane, lane * self.depth + (k // self.lanes) % self.depth

    def _advance_simulator(self, upto):
        b, S = self.b, self.slots
        while self.sim_next <= upto:
            g = self.sim_next
            slot = g % S
            s = self.s_lane[g % self.lanes] if self.sim_on_lanes else self.s_sim
            if self.sim_on_lanes:
                if g > 0:
                    s.wait(self.sub_done[(g - 1) % len(self.sub_done)])   # the simulator is time-sequential: behind the previous substep, wherever it ran
                if self.force_pending:
                    s.wait(self.force_done)
                    self.force_pending = False
            if self.bc_used[slot]:
                s.wait(self.bc_done[slot])    # the slot's previous snapshot has been sent ...
            if self.ip_used[slot]:
                s.wait(self.ip_done[slot])    # ... and consumed by this rank's own render
            b.snapshot(s, slot)
            self.snap_ready[slot].record(s)
            b.substep(s)
            if self.sim_on_lanes:
                self.sub_done[g % len(self.sub_done)].record(s)
            self.sim_next += 1

    def _broadcasts(self, upto):
        b, S = self.b, self.slots
        while self.bc_next <= upto:   # same order on every rank
            slot = self.bc_next % S
            if self.rank == self.owner:
                self.s_comm.wait(self.snap_ready[slot])
            elif self.ip_used[slot]:
                self.s_comm.wait(self.ip_done[slot])   # this rank's render has read the slot's previous snapshot
            b.broadcast(self.s_comm, slot, self.owner)
            self.bc_done[slot].record(self.s_comm)
            self.bc_used[slot] = True
            self.bc_next += 1

    def retire(self, ws):
        """[host] Completes the frame occupying workspace `ws` (if any) and returns (frame, result)."""
        f = self.pending[ws]
        if f is None:
            return None
        b = self.b
        (self.out_ready if self.copy_out else self.render_done)[ws].host_wait()
        if not b.complete(ws):          # rays were still alive after the captured trips: keep going like the reference's loop
            b.finish(ws)                # [host] blocking continuation on the workspace's lane + copy-out again
        self.pending[ws] = None
        res = b.result(ws)
        if self.on_retire is not None:
            self.on_retire(f, res)
        return f, res

    def step(self, pose=None):
        """Enqueues global frame `self.frame`.  Returns the list of (frame, result) this call retired on this rank (a frame comes back
        `lanes * depth` of the rank's frames after it was enqueued; ``drain()`` returns the rest)."""
        b, f, S = self.b, self.frame, self.slots
        self.retired = []
        if self.rank == self.owner:
            self._advance_simulator(f + self.ahead)
        if self.collectives:
            self._broadcasts(f + self.ahead)
        mine = frame_owner(f, self.world, self.owner, self.dedicated) == self.rank
        if mine:
            lane, ws = self._ws(self.my_frames)
            done = self.retire(ws)     # the workspace's previous frame: enqueued lanes*depth frames ago, normally long complete
            if done is not None:
                self.retired.append(done)
            slot = f % S
            s = self.s_lane[lane]
            s.wait(self.bc_done[slot] if self.collectives else self.snap_ready[slot])
            b.render(s, f, ws, slot, pose)
            self.ip_done[slot].record(s)     # recorded after the whole render: conservative (update_F alone reads the snapshot)
            self.ip_used[slot] = True
            self.render_done[ws].record(s)
            if self.copy_out:
                # on the copy stream (overlaps the lane's next frame) or, where a further busy hardware queue costs more than it brings, on the
                # frame's own lane right behind the render (backend.copy_on == "lane")
                where = getattr(b, "copy_on", "copy")   # "sim": the simulator's stream has room (a substep is a third of a step) and is a queue that exists anyway
                sc = s if where == "lane" else (self.s_sim if where == "sim" else self.s_copy)
                if sc is not s:
                    sc.wait(self.render_done[ws])
                b.copy_out(sc, ws)
                self.out_ready[ws].record(sc)
            self.pending[ws] = f
            self.last_ws, self.last_frame = ws, f   # (harness.verify_last_frame: the workspace still holds this frame's inputs after drain())
            self.my_frames += 1
        elif self.dedicated and self.rank == self.owner and self.world > 1:
            # a rank that never renders has nothing that paces its host: wait until this frame's snapshot has been delivered, so that the
            # owner stays at most `ahead` frames in front of the slowest receiver instead of enqueueing the whole job at once
            self.bc_done[f % S].host_wait()
        self.frame += 1
        return self.retired

    def drain(self):
        """[host] Retires every frame still in flight on this rank, in frame order."""
        out = [self.retire(ws) for ws in range(self.lanes * self.depth)]
        out = sorted(r for r in out if r is not None)
        self.retired = out
        return out

    @property
    def substeps_enqueued(self):
        return self.sim_next


# ---------------------------------------------------------------------------------------------------- CPU stand-in of the device side
class _SimEvent:
    def __init__(self, be):
        self.be, self.stream, self.pos = be, None, -1

    def record(self, stream):
        self.stream, self.pos = stream, len(stream.ops) + stream.base   # the point after everything enqueued so far
        stream.ops.append(("record", self, None))

    def host_wait(self):
        self.be.run_until(self)


class _SimStream:
    def __init__(self, name):
        self.name, self.ops, self.base, self.executed = name, [], 0, 0   # ops[i] has absolute index base + i; executed = absolute count run

    def wait(self, event):
        # like hipStreamWaitEvent: the wait refers to the event's record at THIS moment (a later re-record does not move it)
        self.ops.append(("wait", (event.stream, event.pos), None))


class SimulatedBackend:
    """Deferred, dependency-respecting execution of the pipeline's enqueue sequence on CPU tensors (the stand-in for HIP streams).

    Every stream is a FIFO; an operation runs only when the stream's earlier operations have run and the events it waits on have been
    recorded.  ``run_until(event)`` executes what the event needs, plus — seeded randomness — any other operation that happens to be
    ready, so different seeds exercise different legal interleavings.  The operations themselves check the pipeline's safety properties:
    a snapshot slot tagged with frame g may only be overwritten once every consumer of g has run, and a render must find its own frame's
    tag and data in the slot it reads."""

    def __init__(self, world, rank, n_dof=64, owner=0, seed=0, needs_more_trips=lambda frame: False, group=None):
        self.world, self.rank, self.owner, self.group = world, rank, owner, group
        self.rng = random.Random(seed * 1000 + rank)
        self.streams = {}
        self.dof = torch.arange(n_dof, dtype=torch.float64) * (1.0 if rank == owner else -7.0)   # only the owner's copy is ever advanced
        self.steps = 0
        self.snap, self.snap_tag = {}, {}            # slot -> tensor, slot -> frame whose state it holds
        self.readers_left = {}                       # (slot, frame) -> renders of that content enqueued but not yet run
        self.ws_out, self.ws_host, self.ws_trips_short = {}, {}, {}
        self.needs_more_trips = needs_more_trips
        self.log = []

    # ---- streams / events
    def stream(self, name):
        return self.streams.setdefault(name, _SimStream(name))

    def event(self):
        return _SimEvent(self)

    def _enqueue(self, s, what, fn):
        s.ops.append(("op", what, fn))

    def _ready(self, s):
        if not s.ops:
            return False
        kind, a, _ = s.ops[0]
        return kind != "wait" or a[0] is None or a[0].executed > a[1]

    def _run_one(self, s):
        kind, a, fn = s.ops.pop(0)
        s.base += 1
        if kind == "op":
            fn()
        s.executed += 1

    def run_until(self, event):
        guard = 0
        while not (event.stream is None or event.stream.executed > event.pos):
            ready = [s for s in self.streams.values() if self._ready(s)]
            assert ready, f"deadlock: nothing can run while waiting for a point of stream {event.stream.name}"
            # bias towards the awaited stream, but let anything legal happen
            pick = event.stream if (event.stream in ready and self.rng.random() < 0.5) else self.rng.choice(ready)
            self._run_one(pick)
            guard += 1
            assert guard < 10_000_000

    def flush(self):
        while True:
            ready = [s for s in self.streams.values() if self._ready(s)]
            if not ready:
                break
            self._run_one(self.rng.choice(ready))
        assert all(not s.ops for s in self.streams.values()), "operations left that can never run"

    # ---- the five device operations
    def snapshot(self, s, slot):
        g = self._sim_enq = getattr(self, "_sim_enq", 0)
        self._sim_enq += 1

        def run():
            prev = self.snap_tag.get(slot)
            assert self.steps == g, f"snapshot of frame {g} taken after {self.steps} substeps"   # the state BEFORE the frame's own substep (trainer.py:300-318)
            assert self.readers_left.get((slot, prev), 0) == 0, f"snapshot slot {slot} overwritten with frame {g} while frame {prev} still has readers"
            self.snap[slot] = self.dof.clone()
            self.snap_tag[slot] = g
        self._enqueue(s, f"snapshot {g}", run)

    def substep(self, s):
        def run():
            self.dof = self.dof * 1.01 + 0.5
            self.steps += 1
        self._enqueue(s, "substep", run)

    def broadcast(self, s, slot, src):
        g = self._bc_enq = getattr(self, "_bc_enq", 0)
        self._bc_enq += 1

        def run():
            if self.rank == src:
                assert self.snap_tag.get(slot) == g, f"broadcast of frame {g} finds frame {self.snap_tag.get(slot)} in slot {slot}"
                buf = self.snap[slot]
            else:
                prev = self.snap_tag.get(slot)
                assert self.readers_left.get((slot, prev), 0) == 0, f"slot {slot} received frame {g} while frame {prev} still has readers"
                buf = self.snap.setdefault(slot, torch.empty_like(self.dof))
            dist.broadcast(buf, src=src, group=self.group)   # gloo: blocks until every rank runs the same broadcast
            self.snap_tag[slot] = g
        self._enqueue(s, f"broadcast {g}", run)

    def render(self, s, frame, ws, slot, pose):
        key = (slot, frame)   # readers are counted per slot CONTENT: registered at enqueue time, released when the render has run
        self.readers_left[key] = self.readers_left.get(key, 0) + 1

        def run():
            assert self.snap_tag.get(slot) == frame, f"frame {frame} rendered from slot {slot} holding frame {self.snap_tag.get(slot)}"
            self.readers_left[key] -= 1
            short = bool(self.needs_more_trips(frame))
            self.ws_out[ws] = (frame, float(self.snap[slot].sum()), None if pose is None else float(pose), "partial" if short else "full")
            self.ws_trips_short[ws] = short
            self.log.append(frame)
        self._enqueue(s, f"render {frame}", run)

    def copy_out(self, s, ws):
        def run():
            self.ws_host[ws] = self.ws_out[ws]
        self._enqueue(s, f"copy_out {ws}", run)

    # ---- host side of a retired workspace
    def complete(self, ws):
        return not self.ws_trips_short[ws]

    def finish(self, ws):
        f, v, p, _ = self.ws_out[ws]
        self.ws_out[ws] = self.ws_host[ws] = (f, v, p, "full")
        self.ws_trips_short[ws] = False

    def result(self, ws):
        return self.ws_host.get(ws, self.ws_out.get(ws))
