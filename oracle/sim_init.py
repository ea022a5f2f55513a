"""ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED.

Init-time half of the simulator: a restatement of
simulator/solver.py:13-538 with the Warp kernels of simulator/cpu_utils.py and
simulator/cuda_utils.py:3-81,235-279 replaced by numpy formulations.  The
tensor bookkeeping is kept in torch-CPU *on purpose*: several reference lines
round through float32 because of torch's type-promotion rules (e.g.
``kdx = res.max() * dx / (kres-1)`` is a 0-dim float32 tensor, solver.py:184;
``(IP_grid + 0.5) * dx`` is float32, :177), and running the same expressions
reproduces those roundings without restating them by hand.

Unpinned third-party pieces (absent from /root/reference):
  * kornia.utils.grid.create_meshgrid3d (+ the [1,2] channel swap, solver.py:162-169):
    restated as grid[i,j,k] = (i,j,k), the only layout consistent with
    IP_mask[gx,gy,gz] indexing (:141-148,175-177).
  * torch.linalg.inv / .inverse(): numpy.linalg.inv here.
"""
import numpy as np
import torch

from . import (calc_elastic, collect_rhs_IP, matvec3, stepforward as _step, update_F)

torchfloat = torch.float64


# ---- simulator/func_utils.py:73-112 --------------------------------------------------------
def idx(x, y):
    if x > y:
        x, y = y, x
    return 4 + y if x == 0 else 5 + x + y


def P(p):  # [...,3] -> [...,10]
    x, y, z = p[..., 0], p[..., 1], p[..., 2]
    return np.stack([np.ones_like(x), x, y, z, x * x, x * y, x * z, y * y, y * z, z * z], axis=-1)


def Pj(p, j):
    a = np.zeros(p.shape[:-1] + (10,))
    a[..., j + 1] = 1.0
    for i in range(3):
        a[..., idx(i, j)] = p[..., i]
    a[..., idx(j, j)] += p[..., j]
    return a


def Pjk(p, j, k):
    a = np.zeros(p.shape[:-1] + (10,))
    a[..., idx(j, k)] = 1.0
    if j == k:
        a[..., idx(j, k)] += 1.0
    return a


# ---- simulator/func_utils.py:43-70 ---------------------------------------------------------
def weights(r, p, q):
    """p [n,1,3], q [n,8,3] -> w [n,8], dw [n,8,3], ddw [n,8,3,3]."""
    diff = p - q
    d = np.linalg.norm(diff, axis=-1) / r
    inside = d < 1
    om = 1.0 - d ** 2
    w = np.where(inside, om ** 3, 0.0)
    dw = np.where(inside[..., None], -6.0 * (om ** 2)[..., None] * diff / (r ** 2), 0.0)
    e = diff / (r ** 2)
    ddw = -6.0 * (om ** 2)[..., None, None] * np.eye(3) / (r ** 2) + 24.0 * om[..., None, None] * e[..., :, None] * e[..., None, :]
    ddw = np.where(inside[..., None, None], ddw, 0.0)
    return w, dw, ddw


def init_GMLS(r, pos, topo, kernel_pos):
    """calc_G / calc_Gp / calc_weight (simulator/cpu_utils.py:3-152) + the batched inverse (solver.py:357).

    Returns Nx [n,8,10], dNx [n,8,3,10], ddNx [n,8,3,3,10] (fp64).
    """
    pos = np.asarray(pos, np.float64)
    n = pos.shape[0]
    q = np.asarray(kernel_pos, np.float64)[np.asarray(topo)]  # [n,8,3]
    p = pos[:, None, :]
    w, dw, ddw = weights(float(r), p, q)
    act = (w > 0.0)  # `if weight <= 0.0: continue`

    Pq = P(q)  # [n,8,10]
    prim = Pq[..., :, None] * Pq[..., None, :]
    for j in range(3):
        a = Pj(q, j)
        prim = prim + a[..., :, None] * a[..., None, :]
        for k in range(3):
            b = Pjk(q, j, k)
            prim = prim + b[..., :, None] * b[..., None, :]
    prim = prim * act[..., None, None]
    G = np.einsum("ni,niab->nab", w, prim)
    dG = np.einsum("nix,niab->nxab", dw, prim)
    ddG = np.einsum("nixy,niab->nxyab", ddw, prim)
    Gi = np.linalg.inv(G)

    Pv = P(pos)  # [n,10]
    Gp = np.einsum("nab,nb->na", Gi, Pv)
    dPv = np.stack([Pj(pos, x) for x in range(3)], axis=1)  # [n,3,10]
    GidG = np.einsum("nab,nxbc->nxac", Gi, dG)  # Gi dG[x]
    dGp = np.einsum("nab,nxb->nxa", Gi, dPv) - np.einsum("nxab,nb->nxa", GidG, Gp)
    ddGp = np.zeros((n, 3, 3, 10))
    for x in range(3):
        for y in range(3):
            ddPv = Pjk(pos, x, y)
            t = np.einsum("nab,nb->na", Gi, ddPv)
            t -= np.einsum("nab,nb->na", GidG[:, x], np.einsum("nab,nb->na", Gi, dPv[:, y]))
            t -= np.einsum("nab,nb->na", GidG[:, y], np.einsum("nab,nb->na", Gi, dPv[:, x]))
            t -= np.einsum("nab,nb->na", np.einsum("nab,nbc->nac", Gi, ddG[:, x, y]), Gp)
            t += np.einsum("nab,nb->na", GidG[:, y], np.einsum("nab,nb->na", GidG[:, x], Gp))
            t += np.einsum("nab,nb->na", GidG[:, x], np.einsum("nab,nb->na", GidG[:, y], Gp))
            ddGp[:, x, y] = t

    # probe vectors b_c(q): slot 0 = P(q); 1+x = Pj(q,x); idx(x,y) += Pjk(q,x,y) over all 9 ordered pairs
    Bq = np.zeros((n, 8, 10, 10))
    Bq[:, :, 0] = Pq
    for x in range(3):
        Bq[:, :, 1 + x] = Pj(q, x)
        for y in range(3):
            Bq[:, :, idx(x, y)] += Pjk(q, x, y)
    g0 = np.einsum("nicb,nb->nic", Bq, Gp)
    g1 = np.einsum("nicb,njb->nijc", Bq, dGp)
    g2 = np.einsum("nicb,njkb->nijkc", Bq, ddGp)
    Nx = g0 * w[..., None]
    dNx = g0[:, :, None, :] * dw[..., None] + g1 * w[..., None, None]
    ddNx = (g0[:, :, None, None, :] * ddw[..., None] + g1[:, :, None, :, :] * dw[:, :, :, None, None] + g1[:, :, :, None, :] * dw[:, :, None, :, None]
            + g2 * w[..., None, None, None])
    Nx *= act[..., None]
    dNx *= act[..., None, None]
    ddNx *= act[..., None, None, None]
    return Nx, dNx, ddNx


def build_IP_global(dx, dt, topo, mu, lam, rho, Nx, dNx, ddNx, dim):
    """build_IP_global (simulator/cuda_utils.py:22-55): dense (10 n_k)^2 scalar system matrix."""
    mat = np.zeros((dim, dim))
    n = topo.shape[0]
    rows = (np.asarray(topo)[:, :, None] * 10 + np.arange(10)[None, None, :]).reshape(n, 80)
    for v in range(n):
        nv = Nx[v].reshape(80)
        c0 = rho[v] * dx ** 3 / dt ** 2
        c1 = dx ** 3 * (rho[v] * dx ** 2 / 12.0 / dt ** 2 + mu[v] + lam[v])
        c2 = dx ** 5 * (mu[v] + lam[v]) / 12.0
        blk = c0 * np.outer(nv, nv)
        for p in range(3):
            d = dNx[v, :, p, :].reshape(80)
            blk += c1 * np.outer(d, d)
            for q in range(3):
                dd = ddNx[v, :, p, q, :].reshape(80)
                blk += c2 * np.outer(dd, dd)
        mat[np.ix_(rows[v], rows[v])] += blk
    return mat


def build_pin_global(stiff, vidx, topo, Nx, mat):
    """build_pin_global (simulator/cuda_utils.py:58-81)."""
    for vv in vidx:
        rows = (np.asarray(topo)[vv][:, None] * 10 + np.arange(10)[None, :]).reshape(80)
        nv = Nx[vv].reshape(80)
        mat[np.ix_(rows, rows)] += stiff * np.outer(nv, nv)
    return mat


class OracleSimulator:
    """Restatement of simulator/solver.py::Simulator (init + step), CPU only."""

    def __init__(self, dt=1e-2, iters=20, bbox=None, kres=7, dx=1, gravity=None, stiff=1e5, base=None):
        bbox = torch.tensor([1.0, 1.0, 1.0], dtype=torchfloat) if bbox is None else bbox.clone()
        base = torch.tensor([-0.5, -0.5, -0.5], dtype=torchfloat) if base is None else base.clone()
        gravity = torch.tensor([0.0, -9.8, 0.0], dtype=torchfloat) if gravity is None else gravity.clone()
        bbox *= 1.02  # solver.py:24 (in the caller's dtype: main_gui.py passes float32 tensors)
        base *= 1.01  # :25
        bbox, gravity, base = bbox.to(torchfloat), gravity.to(torchfloat), base.to(torchfloat)
        self.dt, self.iters, self.dx, self.kres, self.stiff = dt, iters, dx, kres, stiff
        self.res = (bbox // dx).to(torch.int32)
        self.base, self.gravity = base, gravity

    def InitializeFromArrays(self, pos, mass, mu, lam, pin):
        """solver.py:115-137 with the PLY already parsed."""
        self.pos = torch.from_numpy(np.asarray(pos, np.float64))
        assert self.pos.shape[0] > 0
        self.mass = torch.from_numpy(np.asarray(mass, np.float64))
        self.mu = torch.from_numpy(np.asarray(mu, np.float64))
        self.lam = torch.from_numpy(np.asarray(lam, np.float64))
        self.is_pin = torch.from_numpy(np.asarray(pin).astype(bool))
        self.initialize()

    def initialize(self):  # solver.py:139-331
        res, kres = self.res, self.kres
        self.grid_idx = ((self.pos - self.base) // self.dx).to(torch.int32).long()
        r0, r1, r2 = int(res[0]), int(res[1]), int(res[2])
        self.IP_mask = torch.zeros((r0, r1, r2), dtype=torch.bool)
        gi = self.grid_idx
        self.IP_mask[gi[:, 0], gi[:, 1], gi[:, 2]] = True
        n_IP = int(self.IP_mask.sum())
        self.IP_idx = -torch.ones((r0, r1, r2), dtype=torch.int32)
        self.IP_idx[self.IP_mask] = torch.arange(0, n_IP, 1, dtype=torch.int32)
        self.pts_IP = self.IP_idx[gi[:, 0], gi[:, 1], gi[:, 2]]
        ii, jj, kk = torch.meshgrid(torch.arange(r0, dtype=torch.int32), torch.arange(r1, dtype=torch.int32),
                                    torch.arange(r2, dtype=torch.int32), indexing="ij")
        IP_pos = torch.stack([ii, jj, kk], dim=-1)
        self.IP_grid = IP_pos[self.IP_mask, :]
        self.IP_pos = (self.IP_grid + 0.5) * self.dx + self.base
        self.kernel_mask = torch.zeros((kres, kres, kres), dtype=torch.bool)
        self.kdx = ((res.max()) * self.dx) / (kres - 1)
        IP2K = ((self.IP_pos - self.base) // self.kdx).to(torch.int32).long()
        for S in range(8):
            x, y, z = S >> 2 & 1, S >> 1 & 1, S & 1
            self.kernel_mask[IP2K[:, 0] + x, IP2K[:, 1] + y, IP2K[:, 2] + z] |= True
        n_k = int(self.kernel_mask.sum())
        self.kernel_idx = torch.zeros((kres, kres, kres), dtype=torch.int32)
        self.kernel_idx[self.kernel_mask] = torch.arange(0, n_k, 1, dtype=torch.int32)
        pts2K = ((self.pos - self.base) // self.kdx).to(torch.int32).long()
        self.IP_kernel = torch.zeros((n_IP, 8), dtype=torch.int32)
        self.pts_kernel = torch.zeros((self.pos.size(0), 8), dtype=torch.int32)
        for S in range(8):
            x, y, z = S >> 2 & 1, S >> 1 & 1, S & 1
            self.IP_kernel[:, S] = self.kernel_idx[IP2K[:, 0] + x, IP2K[:, 1] + y, IP2K[:, 2] + z]
            self.pts_kernel[:, S] = self.kernel_idx[pts2K[:, 0] + x, pts2K[:, 1] + y, pts2K[:, 2] + z]
        a = torch.arange(kres, dtype=torch.int32)
        ki, kj, kk_ = torch.meshgrid(a, a, a, indexing="ij")
        kernel_pos = torch.stack([ki, kj, kk_], dim=-1)
        self.kernel_grid = kernel_pos[self.kernel_mask, :]
        self.kernel_pos = self.kernel_grid * self.kdx + self.base
        self.n_k, self.n_IP = n_k, n_IP

        kdx = float(self.kdx)
        kp = self.kernel_pos.numpy()
        self.pts_Nx, self.pts_dNx, self.pts_ddNx = init_GMLS(kdx, self.pos.numpy(), self.pts_kernel.numpy(), kp)
        self.IP_Nx, self.IP_dNx, self.IP_ddNx = init_GMLS(kdx, self.IP_pos.numpy(), self.IP_kernel.numpy(), kp)

        # collect_IP, solver.py:427-450 (collect_param: cuda_utils.py:3-19)
        pts_IP = self.pts_IP.numpy().astype(np.int64)
        m, mu, lam = self.mass.numpy(), self.mu.numpy(), self.lam.numpy()
        s_mu, s_lam, s_m = np.zeros(n_IP), np.zeros(n_IP), np.zeros(n_IP)
        np.add.at(s_mu, pts_IP, mu * m)
        np.add.at(s_lam, pts_IP, lam * m)
        np.add.at(s_m, pts_IP, m)
        self.IP_mu, self.IP_lam, self.IP_rho = s_mu / s_m, s_lam / s_m, s_m / (self.dx ** 3)

        self.build_global()

        # rest DOFs, solver.py:258-275
        dof = np.zeros((n_k, 10, 3))
        dof[:, 0, :] = kp
        for x in range(3):
            dof[:, 1 + x, x] = 1.0
        self.dof = dof.reshape(n_k * 10, 3).copy()
        self.dof_rest = self.dof.copy()
        self.dof_vel = np.zeros_like(self.dof)
        self.dof_f = np.zeros_like(self.dof)
        # rhs_rest (:314), rhs_gravity (:316-331; collect_gravity: cuda_utils.py:262-279)
        self.rhs_rest = self.build_rhs() + matvec3(self.Mmat, self.dof)
        g = self.gravity.numpy()
        rg = np.zeros((n_k * 10, 3))
        rows = (self.IP_kernel.numpy().astype(np.int64)[:, :, None] * 10 + np.arange(10)[None, None, :])
        mIP = self.IP_rho * self.dx * self.dx * self.dx
        np.add.at(rg, rows.reshape(-1), (mIP[:, None, None] * self.IP_Nx).reshape(-1)[:, None] * g[None, :])
        self.rhs_gravity = rg

    def build_global(self):  # solver.py:453-538
        n_k, dx, dt = self.n_k, self.dx, self.dt
        dim = n_k * 10
        topo = self.IP_kernel.numpy()
        mat = build_IP_global(dx, dt, topo, self.IP_mu, self.IP_lam, self.IP_rho, self.IP_Nx, self.IP_dNx, self.IP_ddNx, dim)
        vid = np.nonzero(self.is_pin.numpy())[0]
        assert self.pts_kernel.min() >= 0 and self.pts_kernel.max() < n_k
        mat = build_pin_global(self.stiff, vid, self.pts_kernel.numpy(), self.pts_Nx, mat)
        self.A = mat
        # active kernels = positive diagonal (:499-504); +1e-3 I; inverse; scatter back (:505-511)
        active = np.array([i for i in range(n_k) if mat[i * 10, i * 10] > 0.0], dtype=np.int64)
        self.active = active
        lst = (active[:, None] * 10 + np.arange(10)[None, :]).reshape(-1)
        sub = mat[np.ix_(lst, lst)].copy()
        sub[np.arange(sub.shape[0]), np.arange(sub.shape[0])] += 1e-3
        inv = np.linalg.inv(sub)
        self.Ainv = np.zeros((dim, dim))
        self.Ainv[np.ix_(lst, lst)] = inv
        zero = np.zeros(self.n_IP)
        self.Mmat = build_IP_global(dx, dt, topo, zero, zero, self.IP_rho, self.IP_Nx, self.IP_dNx, self.IP_ddNx, dim)

    def full_matrices(self):
        """The reference's literal (30 n_k)^2 forms (solver.py:493-496,532-538) — small cases only."""
        dim = self.n_k * 10
        G = np.zeros((dim * 3, dim * 3))
        M = np.zeros((dim * 3, dim * 3))
        for c in range(3):
            G[c::3, c::3] = self.Ainv
            M[c::3, c::3] = self.Mmat
        return G, M

    def build_rhs(self):  # solver.py:541-571
        RF, VF, _ = calc_elastic(self.IP_kernel.numpy(), self.IP_dNx, self.dof)
        return collect_rhs_IP(self.dx, self.IP_kernel.numpy(), self.IP_mu, self.IP_lam, self.IP_dNx, RF, VF, self.n_k * 10)

    def get_IP_info(self):  # solver.py:402-424
        return update_F(self.IP_kernel.numpy(), self.dof, self.IP_Nx, self.IP_dNx, self.IP_ddNx)

    def update_force(self, vid, f):  # solver.py:578-588
        f = np.asarray(f, np.float64)
        dof_f = np.zeros_like(self.dof)
        m = self.IP_rho[vid] * (self.dx ** 3)
        topo = self.IP_kernel.numpy()
        for i in range(8):
            kid = int(topo[vid, i])
            for j in range(10):
                dof_f[kid * 10 + j] += m * self.IP_Nx[vid, i, j] * f
        self.dof_f = dof_f

    def clear_force(self):  # solver.py:590-593
        self.dof_f = np.zeros_like(self.dof)

    def state(self):
        return dict(iters=self.iters, dt=self.dt, dx=self.dx, IP_kernel=np.ascontiguousarray(self.IP_kernel.numpy(), np.int32),
                    IP_mu=np.ascontiguousarray(self.IP_mu), IP_lam=np.ascontiguousarray(self.IP_lam), IP_dNx=np.ascontiguousarray(self.IP_dNx),
                    Ainv=np.ascontiguousarray(self.Ainv), Mmat=np.ascontiguousarray(self.Mmat), dof_rest=self.dof_rest, rhs_rest=self.rhs_rest,
                    rhs_gravity=self.rhs_gravity, dof_f=np.ascontiguousarray(self.dof_f), dof=self.dof, dof_vel=self.dof_vel)

    def stepforward(self):  # solver.py:595-602
        _step(self.state())

    step = stepforward

    def update_pos(self):  # solver.py:604-617 (update_pos_kernel: cuda_utils.py:191-203)
        topo = self.pts_kernel.numpy().astype(np.int64)
        d = self.dof.reshape(self.n_k, 10, 3)[topo]  # [n,8,10,3]
        return np.einsum("nic,nicr->nr", self.pts_Nx, d)
