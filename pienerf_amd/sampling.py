"""Point sampling from a trained density field: the MI355X-side mirror of main_sample.py's AdaptiveUniformSampling
(/root/reference/main_sample.py:142-308) — SURVEY.md §8(f) rank 2, off the per-frame hot path.

A `sub_res`^3 lattice is evaluated through the fused network kernel (NeRFNetwork.density -> pn_nerf_density), lattice cells whose
density gradient is non-zero receive `int(cell_size * sub_coeff * res * |grad|)^3` extra points, everything above `density_threshold`
is kept and every kept point gets the volume `hgs^3 / (points in its spatial-hash cell)` (get_pnts_in_grids -> pn_pnts_in_grids).
The result is the `.ply` the simulator consumes (x, y, z, vp; README.md:106-108 turns vp into mass).

Same names and arithmetic as the reference (fp32 tensors, the float hash `g2*res*res + g1*res + g0`, `linspace(-bound, bound, res)`
lattice against a `2*bound/res` cell size).  Deliberate, documented differences — each one replaces undefined or racy behaviour:
  * `get_grid_coords` (main_sample.py:50-66) scatters out-of-range cells (a lattice point at +bound has g = res) into slots of other
    cells or past the end of the array; here only cells with 0 <= g < res are recorded;
  * `get_sub_grid` (:101-140) reads `grid_density[hash(g + 1)]` past the end of the array on the upper faces; such neighbours count 0;
  * `get_sub_bgn` (:74-82) hands out the output ranges by atomic race; here they are the exclusive prefix sum in cell order (same set
    of points, reproducible order);
  * `get_point_volumes` (:182-200) loops over all hash cells on the host; here it is one gather.
The uniform numbers of `torch.rand` (:268) can be passed in (`rand=`) so that a run is reproducible and testable.
"""
import os

import numpy as np
import torch

from ._lib import require_gpu
from .nerf.utils import get_pnts_in_grids


class AdaptiveUniformSampling:
    def __init__(self, opt, model, device="cuda:0"):
        self.device = torch.device(device)
        self.dtype = torch.float32
        self.opt = dict(opt)
        self.bound = float(opt["bound"])
        self.threshold = float(opt.get("density_threshold", 0.05))     # get_opts.py:77
        self.res = int(opt.get("sub_res", 20))                          # get_opts.py:79
        self.sub_coeff = float(opt.get("sub_coeff", 0.1))               # get_opts.py:78
        self.model = model.to(self.device)
        self.grid_size = 2 * self.bound / self.res                      # main_sample.py:155

    # ------------------------------------------------------------------ main_sample.py:164-180
    def get_density(self, x):
        x = x.to(self.device)
        density = self.model.density(x)["sigma"]
        return 1 - torch.exp(-density / 128.0)

    def p2g(self, x):
        return torch.floor((x + self.bound) / self.grid_size)

    def g2p(self, g):
        return g * self.grid_size - self.bound

    def hash_code_g(self, g):
        """Float hash of main_sample.py:45-46 for an integer cell tensor [...,3] -> int64 (fp32 arithmetic, exact below 2^24)."""
        gf = g.to(torch.float32)
        r = torch.tensor(float(self.res), dtype=torch.float32, device=g.device)
        return (gf[..., 2] * r * r + gf[..., 1] * r + gf[..., 0]).to(torch.int64)

    def lattice(self):
        """grid_pts [res^3,3] fp32 (main_sample.py:204-226); row n = i*res^2 + j*res + k holds (x_k, y_j, z_i)."""
        o, res = self.opt, self.res
        if o.get("cut", False):
            cb = list(o["cut_bounds"])
            for a in (0, 2, 4):
                cb[a] = max(cb[a], -self.bound)
            for a in (1, 3, 5):
                cb[a] = min(cb[a], self.bound)
            assert cb[0] < cb[1] and cb[2] < cb[3] and cb[4] < cb[5]
            xs, ys, zs = (torch.linspace(cb[2 * a], cb[2 * a + 1], res) for a in range(3))
        else:
            xs = ys = zs = torch.linspace(-self.bound, self.bound, res)
        zg, yg, xg = torch.meshgrid(zs, ys, xs, indexing="ij")
        return torch.stack([xg, yg, zg], dim=-1).reshape(-1, 3).to(self.device)

    # ------------------------------------------------------------------ main_sample.py:182-200
    def get_point_volumes(self, pts):
        pts = pts.to(self.device, torch.float32).contiguous()
        require_gpu(pts)
        n_vtx = pts.shape[0]
        marg = 1e-3
        bbmin = pts.min(dim=0).values - marg * torch.ones(3, dtype=torch.float32, device=self.device)
        bbmax = pts.max(dim=0).values + marg * torch.ones(3, dtype=torch.float32, device=self.device)
        hgs = float(self.opt["hash_grid_size"])
        resolution = torch.ceil((bbmax - bbmin) / hgs).to(torch.int32)
        n_grid = int(resolution[2] * resolution[1] * resolution[0])
        pig_cnt, pig_bgn, pig_idx = get_pnts_in_grids(n_vtx, n_grid, pts, bbmin, bbmax, hgs, resolution)
        vol = hgs ** 3 / pig_cnt.float()
        cell_of_slot = torch.repeat_interleave(torch.arange(n_grid, device=self.device), pig_cnt.long())   # slot -> cell (pig_bgn order)
        vols = torch.zeros(n_vtx, dtype=torch.float32, device=self.device)
        vols[pig_idx.long()] = vol[cell_of_slot]
        return vols

    # ------------------------------------------------------------------ main_sample.py:202-308
    @torch.no_grad()
    def sample(self, rand=None, generator=None):
        """-> (pts [n,3] fp32, vols [n] fp32) on the device.  `rand`: the [max_add,3] uniform numbers of main_sample.py:268 (more rows
        than needed are fine); default: torch.rand with `generator`."""
        res, dev = self.res, self.device
        n_grid = res ** 3
        grid_pts = self.lattice()
        assert grid_pts.shape[0] > 0, "No grid points, check params!"
        grid_density = self.get_density(grid_pts)
        # get_grid_coords: slot hash(g) <- g for every lattice point whose cell lies inside the lattice
        g = self.p2g(grid_pts).to(torch.int32)
        inside = ((g >= 0) & (g < res)).all(dim=1)
        grid_coords = torch.zeros((n_grid, 3), dtype=torch.int32, device=dev)
        grid_coords[self.hash_code_g(g[inside])] = g[inside]
        # get_sub_grid: forward-difference density gradient over the cell's 8 corners -> number of extra points per axis
        offs = torch.tensor([[0, 0, 0], [0, 0, 1], [0, 1, 0], [0, 1, 1], [1, 0, 0], [1, 0, 1], [1, 1, 0], [1, 1, 1]], dtype=torch.int32, device=dev)
        corner = grid_coords[:, None, :] + offs[None, :, :]                                  # [n_grid,8,3]
        h = self.hash_code_g(corner)
        d = torch.where(h < n_grid, grid_density[h.clamp(max=n_grid - 1)], torch.zeros((), dtype=torch.float32, device=dev))
        grad_x = d[:, 4] + d[:, 5] + d[:, 6] + d[:, 7] - (d[:, 0] + d[:, 1] + d[:, 2] + d[:, 3])
        grad_y = d[:, 2] + d[:, 3] + d[:, 6] + d[:, 7] - (d[:, 0] + d[:, 1] + d[:, 4] + d[:, 5])
        grad_z = d[:, 1] + d[:, 3] + d[:, 5] + d[:, 7] - (d[:, 0] + d[:, 2] + d[:, 4] + d[:, 6])
        grad_norm = torch.sqrt(grad_x * grad_x + grad_y * grad_y + grad_z * grad_z)
        sub_mins = self.g2p(corner[:, 0, :].to(torch.float32))
        sub_maxs = self.g2p(corner[:, 7, :].to(torch.float32))
        rf = torch.tensor(float(res), dtype=torch.float32, device=dev)
        sub_dims = ((sub_maxs - sub_mins)[:, 0] * self.sub_coeff * rf * grad_norm).to(torch.int32)
        flat = grad_norm == 0.0
        sub_dims[flat] = 0
        sub_mins[flat] = 0.0
        sub_maxs[flat] = 0.0
        # get_sub_bgn / get_pnts_add: cell gid receives the first sub_dims^3 uniform points, scaled into the cell
        cnt = sub_dims.long() ** 3
        tot = int(cnt.sum())
        max_add = int(sub_dims.max()) ** 3
        if rand is None:
            rand = torch.rand((max_add, 3), dtype=torch.float32, device=dev, generator=generator)
        rand = rand.to(dev, torch.float32)
        assert rand.shape[0] >= max_add, "not enough uniform numbers"
        assert tot > 0, "No boundary points sampled, check params!"
        owner = torch.repeat_interleave(torch.arange(n_grid, device=dev), cnt)               # output row -> cell, prefix-sum order
        sub_bgn = torch.cumsum(cnt, 0) - cnt
        local = torch.arange(tot, device=dev) - sub_bgn[owner]
        scale = sub_maxs - sub_mins
        pnts_add = scale[owner] * rand[local] + sub_mins[owner]
        # threshold lattice-cell centres + boundary points, then per-point volumes
        pts = torch.cat((pnts_add, grid_pts + 0.5 * 2 * self.bound / float(res)), dim=0)
        density = self.get_density(pts)
        pts = pts[density > self.threshold]
        assert pts.shape[0] > 0, "No points sampled, check params!"
        self.last = dict(grid_points=int(grid_pts.shape[0]), boundary_points=int(tot), kept=int(pts.shape[0]))
        return pts, self.get_point_volumes(pts)


def write_ply(filename, points, volumes, binary=True):
    """main_sample.py:14-23: vertex element with double x, y, z, vp."""
    pts = np.asarray(points.detach().cpu() if torch.is_tensor(points) else points, dtype=np.float64)
    vp = np.asarray(volumes.detach().cpu() if torch.is_tensor(volumes) else volumes, dtype=np.float64)
    rec = np.zeros(len(pts), dtype=[("x", "<f8"), ("y", "<f8"), ("z", "<f8"), ("vp", "<f8")])
    rec["x"], rec["y"], rec["z"], rec["vp"] = pts[:, 0], pts[:, 1], pts[:, 2], vp
    hdr = ["ply", "format binary_little_endian 1.0" if binary else "format ascii 1.0", f"element vertex {len(pts)}"]
    hdr += [f"property double {k}" for k in ("x", "y", "z", "vp")] + ["end_header"]
    os.makedirs(os.path.dirname(os.path.abspath(filename)), exist_ok=True)
    with open(filename, "wb") as f:
        f.write(("\n".join(hdr) + "\n").encode())
        if binary:
            f.write(rec.tobytes())
        else:
            for r in rec:
                f.write((" ".join(repr(float(v)) for v in r) + "\n").encode())


def simulator_cloud(points, volumes, density=1e3, lam=1e6, mu=1e6, pin_height=0.05):
    """README.md:106-108: the attributes the simulator reads — mass = density * vp, constant Lame parameters, the lowest
    `pin_height` of the cloud pinned."""
    pos = np.asarray(points.detach().cpu() if torch.is_tensor(points) else points, dtype=np.float64)
    vp = np.asarray(volumes.detach().cpu() if torch.is_tensor(volumes) else volumes, dtype=np.float64)
    y0 = pos[:, 1].min()
    return dict(pos=pos, mass=density * vp, mu=np.full(len(pos), float(mu)), lam=np.full(len(pos), float(lam)),
                pin=(pos[:, 1] < y0 + pin_height).astype(np.int32))


if __name__ == "__main__":
    import argparse
    from . import scene
    from .nerf.network import NeRFNetwork
    ap = argparse.ArgumentParser(description="sample a simulator point cloud from the (synthetic) checkpoint's density field")
    ap.add_argument("--out", default="model/chair.ply")
    ap.add_argument("--sub_res", type=int, default=60)
    ap.add_argument("--sub_coeff", type=float, default=0.1)
    ap.add_argument("--density_threshold", type=float, default=0.05)
    a = ap.parse_args()
    o = scene.default_opt(sub_res=a.sub_res, sub_coeff=a.sub_coeff, density_threshold=a.density_threshold)
    net = NeRFNetwork(encoding="hashgrid", bound=o["bound"], cuda_ray=True).to("cuda:0").load_checkpoint_dict(scene.make_checkpoint(bound=o["bound"]))
    s = AdaptiveUniformSampling(o, net)
    p, v = s.sample(generator=torch.Generator(device="cuda:0").manual_seed(0))
    write_ply(a.out, p, v)
    print(s.last, "->", os.path.abspath(a.out))
