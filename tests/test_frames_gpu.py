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


def _tile_rccl_worker(rank, port, out_dir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))   # "nccl" IS RCCL on ROCm
    from pienerf_amd.harness import SimRenderHarness
    opt, cloud, ckpt = _scene()
    h = SimRenderHarness(opt, cloud=cloud, ckpt=ckpt, device="cuda:0")
    h.capture_tile_parallel(_force_collectives=True)
    assert h._tile.collectives and h._tile.world == 1
    frames = [h.step_tile_parallel()["image"][0].cpu().numpy() for _ in range(4)]
    assert h._tile._gather is not None   # the all-gather ran
    np.save(os.path.join(out_dir, "tile_rccl.npy"), np.stack(frames))
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


@pytest.mark.parametrize("world,dedicated", [(2, "off"), (2, "on"), (3, "auto")])
def test_bench_multi_gpu_command_line_dry_run(tmp_path, world, dedicated):
    """`bench.py --gpus N` exactly as the driver launches it (torch.distributed.run, one rank per process), with PN_DIST_BACKEND=gloo so that the
    N ranks can share the test box's one GPU (RCCL refuses that): the frame-parallel path of BASELINE configs[3] end to end — process group,
    checkpoint broadcast, both placements of the simulator (owner renders too / owner only simulates), more frames than the snapshot ring holds,
    max-over-ranks timing, and ONE JSON line on rank 0 that carries the rank count, the frames every rank completed and the owner's ceiling."""
    import json
    import subprocess
    import sys

    from conftest import ROOT
    K, W, P = 10, 3, 4
    env = dict(os.environ, PN_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1", "--master-port",
           str(_free_port()), "bench.py", "--gpus", str(world), "--steps", str(K), "--warmup", str(W), "--prime", str(P), "--no-cpu-baseline", "--no-extras",
           "--dedicated-sim", dedicated]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    cfg = d["config"]
    assert d["n_gpus"] == world and cfg["ranks"] == world and cfg["rccl_ranks"] == world and cfg["dist_backend"] == "gloo"
    assert d["scaling"] == "weak" and d["value"] > 0 and d["value_unprimed"] > 0 and d["steps"] == K and d["warmup"] == W
    want_dedicated = {"off": False, "on": True, "auto": world >= 3}[dedicated]
    assert cfg["dedicated_sim"] == want_dedicated
    frames = cfg["frames_per_rank"]
    total = world * (2 * (W + K) + P)
    assert len(frames) == world and sum(frames) == total, (frames, total)
    if want_dedicated:
        assert frames[0] == 0 and min(frames[1:]) > 0
    else:
        assert min(frames) > 0 and max(frames) - min(frames) <= 1
    assert total > 2 * (2 * 2 + 2)                    # more frames than twice the snapshot ring (lanes x depth + ahead)
    ceil = d["frame_parallel_ceiling"]
    assert ceil["owner_frames_per_s"] > 0 and ceil["substep_ms_alone"] > 0.05     # (the ranks share ONE GPU here: the figure itself means nothing)
    assert d["roofline"]["frac"] > 0 and "parallelism" in cfg and f"{world} RCCL ranks" in cfg["parallelism"]


def _rccl_worker(rank, world, port, out_dir, persistent):
    """ONE RCCL rank on cuda:0 that runs every collective of the N-rank schedule (frames.FramePipeline(force_collectives=True))."""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))   # "nccl" IS RCCL on ROCm
    assert dist.get_backend() == "nccl"
    from pienerf_amd.frames import broadcast_tensors
    from pienerf_amd.harness import SimRenderHarness
    opt, cloud, ckpt = _scene()
    h = SimRenderHarness(opt, cloud=cloud, ckpt=ckpt, device="cuda:0")
    m = h.model
    # the one-off checkpoint broadcast of bench.py --gpus N: hash tables, density bitfield, every layer's weights
    before = m.encoder.embeddings.data.clone()
    broadcast_tensors([m.encoder.embeddings.data, m.density_bitfield] + [l.weight.data for l in list(m.sigma_net) + list(m.color_net)], src=0)
    m._net_sig = None
    assert torch.equal(before, m.encoder.embeddings.data)
    if persistent:
        assert h.sim.enable_persistent()   # the substep's iterations as ONE cooperative kernel (csrc/pn_sim.hip: k_substep_coop) beside RCCL's kernels
    h.capture_frame_parallel(lanes=2, n_trips=8, _force_collectives=True)   # graphs on 2 lanes + simulator stream + RCCL broadcasts on the comm stream + copier thread
    p = h._pipe
    assert p.world == 1 and p.collectives and h._pipe_backend.copier is not None
    got = {}
    for f in range(N_FRAMES):
        for idx, res in h.step_frame_parallel():
            got[idx] = res["image"].copy()
    for idx, res in h.drain_pipeline():
        got[idx] = res["image"].copy()
    assert p.bc_next == N_FRAMES + p.ahead    # one broadcast per snapshot, all of them enqueued
    assert h.verify_last_frame()["ok"]
    np.savez(os.path.join(out_dir, "rccl.npz"), **{str(k): v for k, v in got.items()})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("persistent", [False, True])
def test_rccl_snapshot_and_checkpoint_broadcasts_beside_graphs_and_copier(tmp_path, persistent):
    """RCCL itself (backend "nccl"), world size 1 — all this box has: the checkpoint broadcast and one snapshot broadcast per frame on the communication
    stream while the render graphs replay on two lanes, the simulator graph on its stream (both substep forms) and the copier thread moves the frames
    to host memory.  Frames equal the eager single-process sequence bit for bit."""
    import torch.multiprocessing as mp
    from pienerf_amd.harness import SimRenderHarness
    opt, cloud, ckpt = _scene()
    eager = SimRenderHarness(opt, cloud=cloud, ckpt=ckpt, device="cuda:0")
    want = [eager.step()["image"].clone().cpu().numpy() for _ in range(N_FRAMES)]
    eager.synchronize()
    del eager
    torch.cuda.empty_cache()
    mp.spawn(_rccl_worker, args=(1, _free_port(), str(tmp_path), persistent), nprocs=1, join=True)
    with np.load(tmp_path / "rccl.npz") as z:
        got = {int(k): z[k] for k in z.files}
    assert sorted(got) == list(range(N_FRAMES))
    for f in range(N_FRAMES):
        if persistent:   # the cooperative kernel sums the chunk sums in another order than the launch form: 1e-9 on the DOFs
            assert np.abs(want[f] - got[f]).max() < 1e-5, f
        else:
            assert np.abs(want[f] - got[f]).max() == 0.0, f


def test_bench_tile_parallel_command_line_with_rccl_world_of_one():
    """`bench.py --parallelism tile` (frames.TileParallel: every frame's 8 x 8 pixel tiles over the ranks, dof snapshot broadcast + ONE
    all_gather_into_tensor per frame) inside a one-rank RCCL group (PN_FORCE_DIST=1, collectives forced: RCCL itself runs every collective of the
    N-rank schedule — all this box has), and the two-rank schedule through gloo ranks sharing the GPU: one JSON line, `scaling` strong, frames counted."""
    import json
    import subprocess
    import sys

    from conftest import ROOT
    base = ["bench.py", "--parallelism", "tile", "--steps", "6", "--warmup", "2", "--prime", "2", "--no-cpu-baseline", "--no-extras"]
    env = dict(os.environ, PN_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable] + base, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["scaling"] == "strong" and d["value"] > 0 and d["config"]["rccl_ranks"] == 1
    assert "collectives forced" in d["config"]["parallelism"] and "all_gather_into_tensor" in d["config"]["parallelism"]
    env = dict(os.environ, PN_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(_free_port())] + base + ["--gpus", "2"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["frames_per_rank"] == [2 * (6 + 2) + 2] * 2 and "tile-parallel x2" in d["config"]["parallelism"]


def test_tile_parallel_frames_through_rccl_equal_the_eager_frames(tmp_path):
    """frames.TileParallel with its collectives forced inside a one-rank RCCL ("nccl") group: the broadcast of the dof snapshot and the
    all_gather_into_tensor of the rank's tiles run through RCCL, and the frames equal the single-process eager frames bit for bit."""
    import torch.multiprocessing as mp
    from pienerf_amd.harness import SimRenderHarness
    opt, cloud, ckpt = _scene()
    eager = SimRenderHarness(opt, cloud=cloud, ckpt=ckpt, device="cuda:0")
    want = [eager.step()["image"][0].clone().cpu().numpy() for _ in range(4)]
    eager.synchronize()
    del eager
    torch.cuda.empty_cache()
    mp.spawn(_tile_rccl_worker, args=(_free_port(), str(tmp_path)), nprocs=1, join=True)
    got = np.load(tmp_path / "tile_rccl.npy")
    for f in range(4):
        assert np.array_equal(got[f], want[f]), f
