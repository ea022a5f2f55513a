"""Frame-parallel simulate-and-render over the GPUs of one node (BASELINE.json configs[3]).

The simulator is time-sequential and tiny (<= 10 290 fp64 DOFs); rendering is embarrassingly parallel over frames.
So one rank (the *sim owner*) advances the elastodynamics and, per frame, broadcasts the kernel-DOF vector
``dof[30 n_k]`` (<= 82 KB) — the only per-frame exchange — over RCCL (``torch.distributed`` backend "nccl" on ROCm;
intra-node xGMI, one hop to every peer, so a direct broadcast, not a ring).  Every rank holds the checkpoint and the
shape functions, rebuilds ``(p_def, F, dF)`` locally from the received DOFs (pn_sim_update_F) and renders the frames
``f`` with ``frame_owner(f) == rank`` (round-robin over all ranks; from 3 ranks on, over every rank but the sim owner, which then
only simulates and broadcasts: ``dedicated_sim_default``).  Start-up state is made identical by a one-off broadcast of the checkpoint tensors
(or by deterministic re-initialisation on every rank).

The scheduling / exchange logic is backend-agnostic and is exercised on CPU with gloo (tests/test_frames_gloo.py);
the render and sim callables are injected.
"""
import torch
import torch.distributed as dist


def dedicated_sim_default(world_size):
    """Whether the sim owner should only simulate.  The job is bounded by the owner's substep rate (the simulator is time-sequential);
    a substep that shares its GPU with renders runs ~1.8x slower than alone, so from 3 ranks on — where the other
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


class FrameParallel:
    """Drives `n_frames` of sim+render across the ranks of `group`.

    sim_step():          advance the simulator by one substep (called on the sim owner only)
    get_dof() -> tensor: the owner's current DOF vector (flat fp64, on the communication device)
    set_dof(tensor):     install a received DOF vector on this rank
    render(frame):       render `frame` from this rank's current DOF state; the return value is collected
    """

    def __init__(self, sim_step, get_dof, set_dof, render, sim_owner=0, group=None, dedicated_sim=None):
        self.sim_step, self.get_dof, self.set_dof, self.render = sim_step, get_dof, set_dof, render
        self.sim_owner = sim_owner
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.dedicated_sim = dedicated_sim_default(self.world) if dedicated_sim is None else bool(dedicated_sim and self.world > 1)
        self._buf = None

    def run(self, n_frames, first_frame=0):
        """Frame f is rendered from the state BEFORE substep f (the reference's GUI shows the pre-step state,
        nerf/trainer.py:300-318).  Returns {frame: render result} for the frames this rank owns."""
        results = {}
        for f in range(first_frame, first_frame + n_frames):
            if self.world > 1:
                if self.rank == self.sim_owner:
                    buf = self.get_dof()
                else:
                    if self._buf is None:
                        self._buf = torch.empty_like(self.get_dof())
                    buf = self._buf
                dist.broadcast(buf, src=self.sim_owner, group=self.group)  # <= 82 KB: latency-bound, every peer is one xGMI hop
                if self.rank != self.sim_owner:
                    self.set_dof(buf)
            if frame_owner(f, self.world, self.sim_owner, self.dedicated_sim) == self.rank:
                results[f] = self.render(f)
            if self.rank == self.sim_owner:
                self.sim_step()
        return results

    def gather_frame_ids(self, results):
        """All-gather of which frames were rendered where (bookkeeping / tests)."""
        mine = sorted(results.keys())
        if self.world == 1:
            return [mine]
        out = [None] * self.world
        dist.all_gather_object(out, mine, group=self.group)
        return out
