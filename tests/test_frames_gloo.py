"""N > 1 path on CPU: world_size-2 gloo run of the frame-parallel driver (rank 0 simulates and broadcasts the DOF state,
frames are rendered round-robin)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_frames, out_dir, dedicated=None):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pienerf_amd.frames import FrameParallel, broadcast_tensors

    # stand-in simulator: a deterministic linear recurrence on a DOF vector; only rank 0's copy is ever advanced
    n = 30 * 7
    state = {"dof": torch.arange(n, dtype=torch.float64) * (1.0 if rank == 0 else -1.0), "steps": 0}

    def sim_step():
        state["dof"] = state["dof"] * 1.01 + 0.5
        state["steps"] += 1

    def render(frame):
        return float(state["dof"].sum())  # "image" = a checksum of the state the frame was rendered from

    ckpt = [torch.full((5,), float(rank)), torch.full((3, 3), float(rank) + 10)]
    broadcast_tensors(ckpt, src=0)
    assert all(float(t.flatten()[0]) in (0.0, 10.0) for t in ckpt)  # every rank now holds rank 0's "checkpoint"

    fp = FrameParallel(sim_step, lambda: state["dof"], lambda t: state.__setitem__("dof", t.clone()), render, dedicated_sim=dedicated)
    res = fp.run(n_frames)
    ids = fp.gather_frame_ids(res)
    np.save(os.path.join(out_dir, f"r{rank}.npy"), np.array([[f, v] for f, v in sorted(res.items())]).reshape(-1, 2))
    if rank == 0:
        assert state["steps"] == n_frames
        if fp.dedicated_sim:  # the owner renders nothing; frames go round-robin over the other ranks
            assert ids == [[]] + [list(range(r - 1, n_frames, world - 1)) for r in range(1, world)]
        else:
            assert ids == [list(range(r, n_frames, world)) for r in range(world)]
    else:
        assert state["steps"] == 0
    dist.barrier()
    dist.destroy_process_group()


def test_frame_parallel_two_ranks(tmp_path):
    n_frames = 7
    port = _free_port()
    mp.spawn(_worker, args=(2, port, n_frames, str(tmp_path)), nprocs=2, join=True)
    got = np.concatenate([np.load(tmp_path / "r0.npy"), np.load(tmp_path / "r1.npy")])
    got = got[np.argsort(got[:, 0])]
    assert list(got[:, 0].astype(int)) == list(range(n_frames))
    # serial reference: frame f sees the state before substep f
    dof = np.arange(30 * 7, dtype=np.float64)
    want = []
    for f in range(n_frames):
        want.append(dof.sum())
        dof = dof * 1.01 + 0.5
    assert np.allclose(got[:, 1], want, rtol=0, atol=1e-9)


def test_frame_parallel_three_ranks_dedicated_sim_owner(tmp_path):
    """From 3 ranks on the sim owner only simulates and broadcasts (frames.dedicated_sim_default); the frames are the serial sequence."""
    from pienerf_amd.frames import dedicated_sim_default, frame_owner
    assert [dedicated_sim_default(w) for w in (1, 2, 3, 8)] == [False, False, True, True]
    assert [frame_owner(f, 4, 0, True) for f in range(7)] == [1, 2, 3, 1, 2, 3, 1]
    assert [frame_owner(f, 4, 2, True) for f in range(7)] == [0, 1, 3, 0, 1, 3, 0]
    assert [frame_owner(f, 4, 0, False) for f in range(5)] == [0, 1, 2, 3, 0]
    n_frames = 8
    mp.spawn(_worker, args=(3, _free_port(), n_frames, str(tmp_path)), nprocs=3, join=True)
    got = np.concatenate([np.load(tmp_path / f"r{r}.npy") for r in range(3)])
    got = got[np.argsort(got[:, 0])]
    assert list(got[:, 0].astype(int)) == list(range(n_frames)) and len(np.load(tmp_path / "r0.npy")) == 0
    dof = np.arange(30 * 7, dtype=np.float64)
    want = []
    for f in range(n_frames):
        want.append(dof.sum())
        dof = dof * 1.01 + 0.5
    assert np.allclose(got[:, 1], want, rtol=0, atol=1e-9)


def test_single_process_fallthrough():
    from pienerf_amd.frames import FrameParallel, frame_owner
    log = []
    fp = FrameParallel(lambda: log.append("s"), lambda: torch.zeros(3, dtype=torch.float64), lambda t: None, lambda f: log.append(f) or f)
    res = fp.run(3)
    assert res == {0: 0, 1: 1, 2: 2} and log == [0, "s", 1, "s", 2, "s"]
    assert [frame_owner(f, 8) for f in range(10)] == [0, 1, 2, 3, 4, 5, 6, 7, 0, 1]
