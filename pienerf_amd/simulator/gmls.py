"""Init-time Q-GMLS precompute as batched torch tensor algebra on the simulator's device.

Replaces, for initialisation only, the reference's Warp-CPU kernels ``calc_G / calc_Gp / calc_weight``
(simulator/cpu_utils.py:3-152) and Warp-GPU assembly kernels ``build_IP_global / build_pin_global``
(simulator/cuda_utils.py:22-81); like the reference (solver.py:357,508) the two inverses go through torch.linalg.
This is start-up work, not the per-frame hot path; HIP kernels for it are SURVEY §8f rank 1 ("next").

Formulation (deliberately different from the oracle's explicit-matrix one, so the two cross-check):
  prim(q) = P P^T + sum_j Pj Pj^T + sum_jk Pjk Pjk^T = Phi(q)^T Phi(q),
  Phi(q) = [P(q); dP/dx(q); dP/dy(q); dP/dz(q); 2e4; s2 e5; s2 e6; 2e7; s2 e8; 2e9]   (s2 = sqrt 2)
so G = sum_i w_i Phi_i^T Phi_i and products with dG_x / ddG_xy are evaluated matrix-free:
  dGp[x]    = G^-1 (dP_x(p) - dG_x Gp)
  ddGp[x,y] = G^-1 (ddP_xy  - dG_x dGp[y] - dG_y dGp[x] - ddG_xy Gp)
which is cpu_utils.py:70-87 after substituting dGp.
"""
import math

import torch

F64 = torch.float64
_SLOT = {(0, 0): 4, (0, 1): 5, (0, 2): 6, (1, 1): 7, (1, 2): 8, (2, 2): 9}  # idx(), func_utils.py:73-81


def _slot(a, b):
    return _SLOT[(min(a, b), max(a, b))]


def basis(p):
    """P(p) [...,10] (func_utils.py:84-92)."""
    x, y, z = p.unbind(-1)
    return torch.stack([torch.ones_like(x), x, y, z, x * x, x * y, x * z, y * y, y * z, z * z], dim=-1)


def basis_grad(p):
    """dP/dp_j [...,3,10] (Pj, func_utils.py:95-103)."""
    x, y, z = p.unbind(-1)
    o, l = torch.zeros_like(x), torch.ones_like(x)
    return torch.stack([
        torch.stack([o, l, o, o, 2 * x, y, z, o, o, o], dim=-1),
        torch.stack([o, o, l, o, o, x, o, 2 * y, z, o], dim=-1),
        torch.stack([o, o, o, l, o, o, x, o, y, 2 * z], dim=-1)], dim=-2)


def basis_hess(device):
    """d2P/dp_j dp_k [3,3,10], constant (Pjk, func_utils.py:106-112)."""
    h = torch.zeros(3, 3, 10, dtype=F64, device=device)
    for j in range(3):
        for k in range(3):
            h[j, k, _slot(j, k)] = 2.0 if j == k else 1.0
    return h


def init_GMLS(r, pos, topo, kernel_pos, chunk=8192):
    """Shape functions Nx [n,8,10], dNx [n,8,3,10], ddNx [n,8,3,3,10] for points `pos` with neighbour kernels `topo`."""
    outs = ([], [], [])
    for s in range(0, pos.shape[0], chunk):
        res = _init_GMLS_chunk(float(r), pos[s:s + chunk], topo[s:s + chunk], kernel_pos)
        for o, t in zip(outs, res):
            o.append(t)
    return tuple(torch.cat(o, dim=0).contiguous() for o in outs)


def _init_GMLS_chunk(r, pos, topo, kernel_pos):
    dev = pos.device
    n = pos.shape[0]
    q = kernel_pos[topo.long()]                       # [n,8,3]
    diff = pos[:, None, :] - q
    d2 = (diff * diff).sum(-1) / (r * r)
    om = 1.0 - d2
    inside = torch.sqrt((diff * diff).sum(-1)) / r < 1.0  # `d >= 1 -> 0` (func_utils.py:44-49)
    w = torch.where(inside, om ** 3, torch.zeros_like(om))
    active = w > 0.0                                   # `if weight <= 0.0: continue` (cpu_utils.py:28-29)
    e = diff / (r * r)
    dw = torch.where(inside[..., None], -6.0 * (om ** 2)[..., None] * e, torch.zeros_like(e))
    eye = torch.eye(3, dtype=F64, device=dev)
    ddw = -6.0 * (om ** 2)[..., None, None] * eye / (r * r) + 24.0 * om[..., None, None] * e[..., :, None] * e[..., None, :]
    ddw = torch.where(inside[..., None, None], ddw, torch.zeros_like(ddw))
    af = active.to(F64)
    w, dw, ddw = w * af, dw * af[..., None], ddw * af[..., None, None]

    s2 = math.sqrt(2.0)
    const_rows = torch.zeros(6, 10, dtype=F64, device=dev)
    for row, (slot, val) in enumerate(((4, 2.0), (5, s2), (6, s2), (7, 2.0), (8, s2), (9, 2.0))):
        const_rows[row, slot] = val
    Phi = torch.cat([basis(q)[:, :, None, :], basis_grad(q), const_rows.expand(n, 8, 6, 10)], dim=2)  # [n,8,10,10]
    probe_scale = torch.tensor([1, 1, 1, 1, 1, s2, s2, 1, s2, 1], dtype=F64, device=dev)             # probes for slots 5,6,8 are 2e, not s2 e
    Probe = Phi * probe_scale[None, None, :, None]

    def apply(coef, u):  # (sum_i coef_i Phi_i^T Phi_i) u,  coef [n,8], u [n,10]
        t = torch.einsum("nirc,nc->nir", Phi, u)
        return torch.einsum("ni,nirc,nir->nc", coef, Phi, t)

    G = torch.einsum("ni,nirc,nird->ncd", w, Phi, Phi)
    Gi = torch.linalg.inv(G)
    Gp = torch.einsum("ncd,nd->nc", Gi, basis(pos))
    dP = basis_grad(pos)                                # [n,3,10]
    ddP = basis_hess(dev)
    dGp = torch.stack([torch.einsum("ncd,nd->nc", Gi, dP[:, x] - apply(dw[..., x], Gp)) for x in range(3)], dim=1)  # [n,3,10]
    ddGp = torch.empty(n, 3, 3, 10, dtype=F64, device=dev)
    for x in range(3):
        for y in range(3):
            rhs = ddP[x, y][None, :] - apply(dw[..., x], dGp[:, y]) - apply(dw[..., y], dGp[:, x]) - apply(ddw[..., x, y], Gp)
            ddGp[:, x, y] = torch.einsum("ncd,nd->nc", Gi, rhs)

    g0 = torch.einsum("nicb,nb->nic", Probe, Gp)
    g1 = torch.einsum("nicb,njb->nijc", Probe, dGp)
    g2 = torch.einsum("nicb,njkb->nijkc", Probe, ddGp)
    Nx = g0 * w[..., None]
    dNx = g0[:, :, None, :] * dw[..., None] + g1 * w[..., None, None]
    ddNx = (g0[:, :, None, None, :] * ddw[..., None] + g1[:, :, None, :, :] * dw[:, :, :, None, None] + g1[:, :, :, None, :] * dw[:, :, None, :, None]
            + g2 * w[..., None, None, None])
    return Nx, dNx, ddNx


def index_add_ordered(dst, index, src):
    """dst.index_add_(0, index, src) with a summation order that does not depend on the race of fp64 atomics (torch's sort-based
    path), so that two initialisations of the same scene produce bit-identical matrices (the reference's Warp kernels add in race
    order, cuda_utils.py:22-81; an initialisation-time cost only)."""
    prev, warn = torch.are_deterministic_algorithms_enabled(), torch.is_deterministic_algorithms_warn_only_enabled()
    torch.use_deterministic_algorithms(True)
    try:
        dst.index_add_(0, index, src)
    finally:
        torch.use_deterministic_algorithms(prev, warn_only=warn)
    return dst


def assemble_IP_matrix(dim, dx, dt, topo, mu, lam, rho, Nx, dNx, ddNx, chunk=512):
    """System matrix [dim,dim] (dim = 10 n_k) of build_IP_global (cuda_utils.py:22-55): per IP the 80x80 block
    Z^T diag(c) Z with Z = [N; dN_p; ddN_pq] (13 x 80)."""
    dev = Nx.device
    n = Nx.shape[0]
    mat = torch.zeros(dim * dim, dtype=F64, device=dev)
    ar10 = torch.arange(10, device=dev)
    for s in range(0, n, chunk):
        sl = slice(s, min(s + chunk, n))
        m = sl.stop - sl.start
        Z = torch.cat([Nx[sl].reshape(m, 1, 80), dNx[sl].permute(0, 2, 1, 3).reshape(m, 3, 80), ddNx[sl].permute(0, 2, 3, 1, 4).reshape(m, 9, 80)], dim=1)
        c0 = rho[sl] * dx ** 3 / dt ** 2
        c1 = dx ** 3 * (rho[sl] * dx ** 2 / 12.0 / dt ** 2 + mu[sl] + lam[sl])
        c2 = dx ** 5 * (mu[sl] + lam[sl]) / 12.0
        coef = torch.cat([c0[:, None], c1[:, None].expand(m, 3), c2[:, None].expand(m, 9)], dim=1)   # [m,13]
        blocks = torch.einsum("vr,vra,vrb->vab", coef, Z, Z)                                            # [m,80,80]
        rows = (topo[sl].long()[:, :, None] * 10 + ar10[None, None, :]).reshape(m, 80)
        flat = rows[:, :, None] * dim + rows[:, None, :]
        index_add_ordered(mat, flat.reshape(-1), blocks.reshape(-1))
    return mat.view(dim, dim)


def add_pin_penalty(mat, stiff, pin_ids, pts_topo, pts_Nx):
    """build_pin_global (cuda_utils.py:58-81): mat += stiff * N N^T for every pinned point."""
    if pin_ids.numel() == 0:
        return mat
    dev = mat.device
    dim = mat.shape[0]
    nv = pts_Nx[pin_ids].reshape(-1, 80)
    rows = (pts_topo[pin_ids].long()[:, :, None] * 10 + torch.arange(10, device=dev)[None, None, :]).reshape(-1, 80)
    flat = rows[:, :, None] * dim + rows[:, None, :]
    index_add_ordered(mat.view(-1), flat.reshape(-1), (stiff * nv[:, :, None] * nv[:, None, :]).reshape(-1))
    return mat


def inverse_spd(mat):
    """Dense fp64 inverse of the system block (solver.py:508: `.inverse()`, an LU).  The block is symmetric positive definite (mass + stiffness +
    pin penalty + 1e-3 I), so it is factored as L L^T and inverted from the factor (half the work of the LU, a symmetric result); LU if the
    factorisation reports a non-positive pivot.  On the device when its solver stack is available, else host LAPACK (init only).  The explicit
    inverse is kept for the substep itself: applying the factor instead would be two triangular solves — a chain of 10 n_k dependent steps per
    local/global iteration against one 7 us matrix-vector product."""
    def _inv(m):
        L, info = torch.linalg.cholesky_ex(m)
        if int(info) == 0:
            return torch.cholesky_inverse(L)
        return torch.linalg.inv(m)
    try:
        return _inv(mat)
    except RuntimeError:
        return _inv(mat.cpu()).to(mat.device)
