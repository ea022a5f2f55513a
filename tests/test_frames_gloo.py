"""The frame pipeline's schedule (pienerf_amd/frames.py: FramePipeline — the class the HIP harness and bench.py --gpus N run) on CPU:
1-4 gloo ranks, both placements of the simulator, many more frames than snapshot slots, randomised legal interleavings of the simulated
streams.  The simulated backend asserts the safety properties while it runs (no snapshot slot overwritten or received into while a reader
is outstanding, every render finds its own frame's state); the tests check the results against the serial sequence."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _serial(n_frames, n_dof=64):
    dof = np.arange(n_dof, dtype=np.float64)
    want = []
    for _ in range(n_frames):
        want.append(dof.sum())
        dof = dof * 1.01 + 0.5
    return want


def _run_rank(rank, world, n_frames, lanes, depth, dedicated, seed, ahead=None, short=(), sim_on_lanes=False):
    from pienerf_amd.frames import FramePipeline, SimulatedBackend
    be = SimulatedBackend(world, rank, seed=seed, needs_more_trips=lambda f: f in short)
    pipe = FramePipeline(be, world=world, rank=rank, lanes=lanes, depth=depth, dedicated_sim=dedicated, ahead=ahead, sim_on_lanes=sim_on_lanes)
    got = []
    for f in range(n_frames):
        got += pipe.step(pose=float(f) * 0.25)    # a per-frame pose travels with the frame
    got += pipe.drain()
    be.flush()
    return pipe, be, got


def _worker(rank, world, port, n_frames, lanes, depth, dedicated, seed, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pienerf_amd.frames import broadcast_tensors, frame_owner
    ckpt = [torch.full((5,), float(rank)), torch.full((3, 3), float(rank) + 10)]
    broadcast_tensors(ckpt, src=0)
    assert all(float(t.flatten()[0]) in (0.0, 10.0) for t in ckpt)  # every rank now holds rank 0's "checkpoint"
    pipe, be, got = _run_rank(rank, world, n_frames, lanes, depth, dedicated, seed, short=(3, 10))
    frames = [f for f, _ in got]
    assert frames == sorted(frames) and frames == [f for f in range(n_frames) if frame_owner(f, world, 0, pipe.dedicated) == rank]
    assert all(r[0] == f and r[2] == f * 0.25 and r[3] == "full" for f, r in got)      # own frame, own pose, continued to completion if short
    if rank == 0:
        assert be.steps == n_frames + pipe.ahead and pipe.substeps_enqueued == n_frames + pipe.ahead
    else:
        assert be.steps == 0
    np.save(os.path.join(out_dir, f"r{rank}.npy"), np.array([[f, r[1]] for f, r in got]).reshape(-1, 2))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,lanes,depth,dedicated,seed", [(2, 2, 2, None, 0), (2, 1, 2, False, 1), (3, 2, 1, None, 2), (3, 2, 2, False, 3), (4, 2, 2, None, 4),
                                                               (4, 3, 1, False, 5)])
def test_frame_pipeline_multi_rank(tmp_path, world, lanes, depth, dedicated, seed):
    n_frames = 61   # several times the snapshot ring (2 * world * lanes * depth + 1 slots)
    mp.spawn(_worker, args=(world, _free_port(), n_frames, lanes, depth, dedicated, seed, str(tmp_path)), nprocs=world, join=True)
    got = np.concatenate([np.load(tmp_path / f"r{r}.npy") for r in range(world)])
    got = got[np.argsort(got[:, 0])]
    assert list(got[:, 0].astype(int)) == list(range(n_frames))
    assert np.allclose(got[:, 1], _serial(n_frames), rtol=0, atol=1e-9)      # frame f sees the state before substep f
    from pienerf_amd.frames import dedicated_sim_default
    if dedicated is None and dedicated_sim_default(world):
        assert len(np.load(tmp_path / "r0.npy")) == 0                        # the owner only simulates and broadcasts


@pytest.mark.parametrize("lanes,depth,ahead,seed", [(1, 2, None, 0), (3, 2, None, 1), (2, 1, 1, 2), (1, 1, 0, 3), (3, 1, 7, 4)])
def test_frame_pipeline_single_rank(lanes, depth, ahead, seed):
    n_frames = 40
    pipe, be, got = _run_rank(0, 1, n_frames, lanes, depth, None, seed, ahead=ahead, short=(5,))
    assert [f for f, _ in got] == list(range(n_frames))
    assert np.allclose([r[1] for _, r in got], _serial(n_frames), rtol=0, atol=1e-9)
    assert all(r[3] == "full" for _, r in got) and sorted(be.log) == list(range(n_frames))
    assert pipe.substeps_enqueued == n_frames + pipe.ahead
    # frames come back lanes * depth frames late, in order, and drain() returns the rest
    pipe2, _, _ = _run_rank(0, 1, 0, lanes, depth, None, seed, ahead=ahead)
    assert pipe2.drain() == []


@pytest.mark.parametrize("lanes,depth,ahead,seed", [(4, 2, None, 0), (3, 2, None, 1), (2, 1, 1, 2), (1, 1, 0, 3), (4, 1, 9, 4), (2, 2, 3, 5)])
def test_frame_pipeline_with_the_substeps_on_the_render_lanes(lanes, depth, ahead, seed):
    """FramePipeline(sim_on_lanes=True): no simulator stream; substep g is enqueued on lane g % lanes and chained to substep g - 1 by an event.  Under every
    legal interleaving of the lanes (the stand-in draws them at random) the snapshot of frame g is taken after exactly g substeps, no slot is overwritten
    while a render still reads it, and every frame sees the state before its own substep."""
    n_frames = 50
    pipe, be, got = _run_rank(0, 1, n_frames, lanes, depth, None, seed, ahead=ahead, short=(7,), sim_on_lanes=True)
    assert pipe.sim_on_lanes and not be.stream("sim").ops and be.stream("sim").executed == 0      # nothing ever ran on a simulator stream
    assert [f for f, _ in got] == list(range(n_frames))
    assert np.allclose([r[1] for _, r in got], _serial(n_frames), rtol=0, atol=1e-9)
    assert be.steps == n_frames + pipe.ahead and pipe.substeps_enqueued == n_frames + pipe.ahead


def test_schedule_helpers():
    from pienerf_amd.frames import dedicated_sim_default, frame_owner
    assert [dedicated_sim_default(w) for w in (1, 2, 3, 8)] == [False, False, True, True]
    assert [frame_owner(f, 4, 0, True) for f in range(7)] == [1, 2, 3, 1, 2, 3, 1]
    assert [frame_owner(f, 4, 2, True) for f in range(7)] == [0, 1, 3, 0, 1, 3, 0]
    assert [frame_owner(f, 4, 0, False) for f in range(5)] == [0, 1, 2, 3, 0]
    assert [frame_owner(f, 8) for f in range(10)] == [0, 1, 2, 3, 4, 5, 6, 7, 0, 1]


def test_simulated_backend_catches_a_missing_dependency():
    """The stand-in is only worth something if it fails when an ordering edge is missing: drop the wait that protects a snapshot slot
    from being overwritten before its render ran, and some interleaving must trip the assertion."""
    from pienerf_amd.frames import FramePipeline, SimulatedBackend

    class Broken(FramePipeline):
        def _advance_simulator(self, upto):
            self.ip_used = [False] * self.slots      # "forget" that this rank's renders read the slots
            super()._advance_simulator(upto)
    hit = 0
    for seed in range(20):
        be = SimulatedBackend(1, 0, seed=seed)
        pipe = Broken(be, lanes=2, depth=2, ahead=2)
        pipe.slots = 4                                # a deliberately tight ring
        try:
            for f in range(40):
                pipe.step()
            pipe.drain()
            be.flush()
        except AssertionError:
            hit += 1
    assert hit > 0


def _tile_worker(rank, world, port, W, H, n_frames, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pienerf_amd.frames import TileParallel
    state = {"dof": torch.arange(64, dtype=torch.float64) * (1.0 if rank == 0 else -3.0), "steps": 0}

    def render_subset(idx):   # "pixel" = f(ray index, state): 5 channels like image | depth | depth_0
        base = idx.clamp(min=0).to(torch.float32)[:, None] * torch.tensor([1.0, 2.0, 3.0, 0.5, 0.25])
        return (base + float(state["dof"].sum()) * 1e-3).to(torch.float32)

    def sim_step():
        state["dof"] = state["dof"] * 1.01 + 0.5
        state["steps"] += 1
    tp = TileParallel(W, H, render_subset, lambda: state["dof"], lambda t: state.__setitem__("dof", t.clone()), sim_step)
    frames = [tp.step().numpy().copy() for _ in range(n_frames)]
    assert state["steps"] == (n_frames if rank == 0 else 0)
    np.save(os.path.join(out_dir, f"t{rank}.npy"), np.stack(frames))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,W,H", [(2, 40, 24), (4, 50, 30), (3, 17, 9)])
def test_tile_parallel_frames(tmp_path, world, W, H):
    """Ray-tile-parallel rendering of one frame: interleaved 8 x 8 tiles per rank, DOF broadcast, all-gather — every rank ends up
    with the whole frame of the owner's state, also when the tile count does not divide evenly (padding)."""
    from pienerf_amd.frames import tile_partition
    parts = tile_partition(W, H, world)
    flat = torch.cat([p[p >= 0] for p in parts])
    assert sorted(flat.tolist()) == list(range(W * H)) and len({p.numel() for p in parts}) == 1
    assert all((p[p >= 0] // W // 8 * ((W + 7) // 8) + p[p >= 0] % W // 8) .remainder(world).eq(r).all() for r, p in enumerate(parts))
    n_frames = 4
    mp.spawn(_tile_worker, args=(world, _free_port(), W, H, n_frames, str(tmp_path)), nprocs=world, join=True)
    got = [np.load(tmp_path / f"t{r}.npy") for r in range(world)]
    dof = np.arange(64, dtype=np.float64)
    for f in range(n_frames):
        want = (np.arange(W * H, dtype=np.float32)[:, None] * np.float32([1.0, 2.0, 3.0, 0.5, 0.25]) + np.float32(dof.sum() * 1e-3)).astype(np.float32)
        for r in range(world):
            assert np.allclose(got[r][f], want, rtol=1e-6, atol=1e-4), (f, r)
        dof = dof * 1.01 + 0.5


def _form_worker(rank, world, port, src, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pienerf_amd.frames import agree_on_launch_form
    # what each rank's own warm-up frame "found": different answers at a threshold
    kw = [dict(fused_from=1, fused_whole=False, fused_fold=True, other="kept"), dict(fused_from=0, fused_whole=True, fused_fold=False, other="kept"),
          dict(fused_from=2, fused_whole=False, fused_fold=False, other="kept")][rank]
    agree_on_launch_form(kw, src=src, group=None)   # group=None = the default process group, what every real call site passes
    np.save(os.path.join(out_dir, f"f{rank}.npy"), np.array([kw["fused_from"], int(kw["fused_whole"]), int(kw["fused_fold"]), int(kw["other"] == "kept")]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,src", [(2, 0), (3, 1)])
def test_ranks_agree_on_the_owners_launch_form(tmp_path, world, src):
    """harness._HipBackend's broadcast of (fused_from, fused_whole, fused_fold) with the DEFAULT process group (group=None): every rank ends with
    rank src's choice.  Round 5 gated this on `group is not None`, which no real call site satisfies — the ranks kept their own answers."""
    mp.spawn(_form_worker, args=(world, _free_port(), src, str(tmp_path)), nprocs=world, join=True)
    got = [np.load(tmp_path / f"f{r}.npy").tolist() for r in range(world)]
    want = [[1, 0, 1, 1], [0, 1, 0, 1], [2, 0, 0, 1]][src]
    assert all(g == want for g in got), got
