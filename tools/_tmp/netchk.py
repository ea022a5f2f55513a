import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pienerf_amd import scene
from pienerf_amd.nerf.network import NeRFNetwork
ckpt = scene.make_checkpoint(bound=1.0, seed=0)
net = NeRFNetwork(encoding="hashgrid", bound=1.0, cuda_ray=True).to("cuda:0").load_checkpoint_dict(ckpt)
rng = np.random.default_rng(5)
for M in (4099, 100000, 1000003):
    x = torch.tensor((rng.random((M, 3)).astype(np.float32) * 2 - 1) * 0.9, device="cuda:0")
    d = rng.standard_normal((M, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=-1, keepdims=True); d = torch.tensor(d, device="cuda:0")
    with torch.no_grad():
        s2, c2 = net.forward_ops(x, d)
        outs = [net(x, d) for _ in range(4)]
    torch.cuda.synchronize()
    for i, (s, c) in enumerate(outs):
        es = (s / s2 - 1).abs(); ec = (c - c2).abs()
        same = bool((s == outs[0][0]).all() and (c == outs[0][1]).all())
        print(M, i, "finite", bool(torch.isfinite(s).all()), "sigma rel max %.2e" % es.max().item(), "n>1e-4:", int((es > 1e-4).sum()), "rgb abs %.2e" % ec.max().item(), "same_as_run0", same)
M = 100000
x = torch.tensor((rng.random((M, 3)).astype(np.float32) * 2 - 1) * 0.9, device="cuda:0")
d = rng.standard_normal((M, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=-1, keepdims=True); d = torch.tensor(d, device="cuda:0")
with torch.no_grad():
    s2, c2 = net.forward_ops(x, d)
    for r in range(3):
        s, c = net(x, d)
        bad = torch.nonzero((s / s2 - 1).abs() > 1e-4).reshape(-1).cpu().numpy()
        badc = torch.nonzero(((c - c2).abs() > 1e-4).any(-1)).reshape(-1).cpu().numpy()
        print("run", r, "bad sigma idx", bad[:40], "tiles", np.unique(bad // 32)[:20], "lane-in-tile", (bad % 32)[:40])
        print("      bad rgb idx", badc[:40])
