"""N > 1 path on the GPU box: two processes (gloo, both on cuda:0 — the test box has one GPU; RCCL refuses two ranks on one
device) drive SimRenderHarness.capture_frame_parallel / step_frame_parallel.  Rank 0 owns the simulator and broadcasts the DOF
snapshots, frames are rendered round-robin from them; the images must be the eager single-process sequence."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

N_FRAMES = 9
SMALL = dict(sub_res=30, sim_dx=0.1, sim_iters=4)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _scene():
    from pienerf_amd import scene
    opt = scene.default_opt(sim_dx=SMALL["sim_dx"], sim_iters=SMALL["sim_iters"], W=64, H=64)
    cloud = scene.make_chair_points(sub_res=SMALL["sub_res"], hgs=opt["hash_grid_size"])
    ckpt = scene.make_checkpoint(bound=1.0, seed=0)
    return opt, cloud, ckpt


def _worker(rank, world, port, out_dir, dedicated=None):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pienerf_amd.harness import SimRenderHarness
    opt, cloud, ckpt = _scene()
    from pienerf_amd.frames import frame_owner
    h = SimRenderHarness(opt, cloud=cloud, ckpt=ckpt, device="cuda:0").capture_frame_parallel(lanes=2, n_trips=8, dedicated_sim=dedicated)
    p = h._pipe
    assert (p.world, p.rank) == (world, rank) and p.dedicated == bool(dedicated)
    got = {}
    for f in range(N_FRAMES):
        for idx, res in h.step_frame_parallel():
            got[idx] = res["image"].copy()
    for idx, res in h.drain_pipeline():
        got[idx] = res["image"].copy()
    assert sorted(got) == [f for f in range(N_FRAMES) if frame_owner(f, world, 0, bool(dedicated)) == rank]
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), **{str(k): v for k, v in got.items()})
    if rank == 0:
        assert h.substeps_enqueued == N_FRAMES + p.ahead
    else:
        assert h.substeps_enqueued == 0  # only the owner's simulator ever advances
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("dedicated", [False, True])
def test_frame_parallel_two_ranks_on_gpu(tmp_path, dedicated):
    """dedicated = True forces the >= 3-rank policy onto two ranks: rank 0 only simulates and broadcasts, rank 1 renders every frame."""
    import torch.multiprocessing as mp
    from pienerf_amd.harness import SimRenderHarness
    opt, cloud, ckpt = _scene()
    eager = SimRenderHarness(opt, cloud=cloud, ckpt=ckpt, device="cuda:0")
    want = [eager.step()["image"].clone().cpu().numpy() for _ in range(N_FRAMES)]
    eager.synchronize()
    del eager
    torch.cuda.empty_cache()
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), dedicated), nprocs=2, join=True)
    got = {}
    for r in range(2):
        with np.load(tmp_path / f"r{r}.npz") as z:
            assert not (dedicated and r == 0 and z.files)
            got.update({int(k): z[k] for k in z.files})
    assert sorted(got) == list(range(N_FRAMES))
    for f in range(N_FRAMES):
        assert np.abs(want[f] - got[f]).max() < 1e-5, f
    assert np.abs(want[0] - want[N_FRAMES - 1]).max() > 1e-3  # the object really moved


def test_frame_parallel_single_rank_equals_pipelined(tmp_path):
    """world = 1 (no process group): step_frame_parallel is step_pipelined with a deeper snapshot ring."""
    from pienerf_amd.harness import SimRenderHarness
    opt, cloud, ckpt = _scene()
    eager = SimRenderHarness(opt, cloud=cloud, ckpt=ckpt, device="cuda:0")
    fp = SimRenderHarness(opt, cloud=cloud, ckpt=ckpt, device="cuda:0").capture_frame_parallel(lanes=3, n_trips=8)
    want = [eager.step()["image"].clone() for _ in range(7)]
    eager.synchronize()
    got = []
    for f in range(7):
        got += [(i, r["image"].copy()) for i, r in fp.step_frame_parallel()]   # the pinned arrays are recycled: keep a copy
    got += [(i, r["image"].copy()) for i, r in fp.drain_pipeline()]
    assert [i for i, _ in got] == list(range(7))
    for f in range(7):
        assert np.array_equal(got[f][1], want[f][0].cpu().numpy()), f


def _tile_worker(rank, world, port, out_dir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pienerf_amd.harness import SimRenderHarness
    opt, cloud, ckpt = _scene()
    h = SimRenderHarness(opt, cloud=cloud, ckpt=ckpt, device="cuda:0", overlap_sim=False).capture_tile_parallel()
    frames = [h.step_tile_parallel()["image"][0].cpu().numpy() for _ in range(4)]
    np.save(os.path.join(out_dir, f"t{rank}.npy"), np.stack(frames))
    dist.barrier()
    dist.destroy_process_group()


def test_tile_parallel_two_ranks_on_gpu(tmp_path):
    """Ray-tile-parallel rendering (harness.capture_tile_parallel): two gloo ranks on the one GPU each render half of the 8 x 8 tiles of every frame from
    rank 0's broadcast DOFs; both end up with the whole frame, equal to the single-process eager frame bit for bit (rays are independent)."""
    import torch.multiprocessing as mp
    from pienerf_amd.harness import SimRenderHarness
    opt, cloud, ckpt = _scene()
    eager = SimRenderHarness(opt, cloud=cloud, ckpt=ckpt, device="cuda:0")
    want = [eager.step()["image"][0].clone().cpu().numpy() for _ in range(4)]
    eager.synchronize()
    del eager
    torch.cuda.empty_cache()
    mp.spawn(_tile_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        got = np.load(tmp_path / f"t{r}.npy")
        for f in range(4):
            assert np.array_equal(got[f], want[f]), (r, f)
    assert np.abs(want[0] - want[3]).max() > 1e-4
