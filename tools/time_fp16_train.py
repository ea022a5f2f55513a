import sys, time, torch, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from pienerf_amd import scene
from pienerf_amd.nerf.network import NeRFNetwork
from pienerf_amd.training import RayImageSet, Trainer
DEV = 'cuda:0'
Wd = 64
intr = scene.orbit_intrinsics(Wd, Wd, 50.0)
poses = np.stack([scene.orbit_pose(4.0, a, -20.0) for a in (0.0, 120.0, 240.0)]).astype(np.float32)
images = torch.rand(3, Wd, Wd, 3, device=DEV)
for fp16 in (False, True):
    torch.manual_seed(1)
    student = NeRFNetwork(encoding="hashgrid", bound=1.0, cuda_ray=True, density_thresh=10).to(DEV)
    data = RayImageSet(torch.from_numpy(poses).to(DEV), intr, images, generator=torch.Generator().manual_seed(2))
    tr = Trainer(student, dict(dt_gamma=0, max_steps=512, T_thresh=1e-2), lr=1e-2, iters=400, num_rays=2048, fp16=fp16)
    tr.train(data, 3)
    torch.cuda.synchronize(); t0 = time.time()
    tr.train(data, 20)
    torch.cuda.synchronize(); print('fp16', fp16, (time.time() - t0) / 20 * 1e3, 'ms per step')
    # pieces
    student.train()
    with torch.autocast('cuda', dtype=torch.float16, enabled=fp16):
        torch.cuda.synchronize(); t0 = time.time(); student.update_extra_state(); torch.cuda.synchronize(); print('  update_extra_state', (time.time() - t0) * 1e3)
        b = data.batch(2048)
        torch.cuda.synchronize(); t0 = time.time(); _, _, loss = tr.train_step(b); torch.cuda.synchronize(); print('  train_step fwd', (time.time() - t0) * 1e3)
    torch.cuda.synchronize(); t0 = time.time(); tr.scaler.scale(loss).backward(); torch.cuda.synchronize(); print('  backward', (time.time() - t0) * 1e3)
