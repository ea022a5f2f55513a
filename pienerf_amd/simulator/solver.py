"""``Simulator`` with the reference's public surface (simulator/solver.py:12-617, /root/reference).

Kept: constructor arguments, ``InitializeFromPly``, ``get_IP_info`` (fp32, permuted layouts of solver.py:422-424),
``stepforward`` (+ alias ``step``), ``update_force`` / ``clear_force``, ``OutputToPly``, attributes ``dx``, ``IP_pos``,
``dof`` ... as torch tensors on the GPU.  Changed on purpose:
  * the per-substep work is HIP (libpienerf_hip.so: pn_sim_stepforward / pn_sim_update_F / pn_sim_update_force);
  * ``global_matrix`` / ``mass_matrix_invt2`` are stored in their kron(A, I3) factor form ``Ainv`` / ``Mmat``
    ([10 n_k]^2 instead of [30 n_k]^2, solver.py:493-496,532-538) — same products, 9x fewer bytes;
  * importing this module does not call ``torch.set_default_device("cuda")`` (func_utils.py:6).
"""
import ctypes as C
import os

import numpy as np
import torch

from .. import scene
from .._lib import check, lib, ptr, stream_ptr
from . import gmls

torchfloat = torch.float64
npfloat = np.float64


class _null_ctx:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


CELL_CHUNK_IPS = 32   # points per chunk of the substep's cell form = pn_sim_cells_chunk_ips() (csrc/pn_sim.hip: PN_CELL_IPS), checked in _prepare_cells


class Simulator:
    def __init__(self, dt=1e-2, iters=20, bbox=torch.tensor([1.0, 1.0, 1.0], dtype=torchfloat), kres=7, dx=1,
                 gravity=torch.tensor([0.0, -9.8, 0.0], dtype=torchfloat), stiff=1e5, base=torch.tensor([-0.5, -0.5, -0.5], dtype=torchfloat),
                 device="cuda", persistent=None, svd=None):
        self.device = torch.device(device)
        # svd: which decomposition stands in for wp.svd3 (cuda_utils.py:107) in calc_elastic.  "jacobi" (default): the converged, warm-started threshold
        # Jacobi; "mcadams" / "mcadams:N": the published algorithm wp.svd3 implements (McAdams et al., TR1690) with N fixed sweeps (default 8, the setting
        # of double-precision builds; 4 is the paper's single-precision setting) — csrc/pn_sim.hip: svd3_mcadams.  None: environment PN_SIM_SVD.
        svd = (svd if svd is not None else os.environ.get("PN_SIM_SVD", "jacobi")).strip().lower()
        name, _, n = svd.partition(":")
        if name not in ("jacobi", "mcadams") or (n and not n.isdigit()) or (name == "jacobi" and n):
            raise ValueError(f"Simulator: svd must be 'jacobi', 'mcadams' or 'mcadams:<sweeps>', got {svd!r}")
        self.svd_sweeps = 0 if name == "jacobi" else int(n or 8)
        if self.svd_sweeps == 0 and name == "mcadams" or self.svd_sweeps > 64:
            raise ValueError("Simulator: mcadams sweeps must be in 1..64")
        # persistent: run the local/global iterations of a substep as ONE cooperative kernel (pn_sim_stepforward_coop) instead of four launches per
        # iteration.  None: environment PN_SIM_COOP (1 / 0), default off — the persistent kernel wants every CU for itself, which suits a GPU that
        # only simulates (the owner rank of a frame-parallel job, a latency-bound single frame) and not one that renders three frames beside it
        if persistent is None:
            persistent = os.environ.get("PN_SIM_COOP", "") == "1"
        self.persistent = bool(persistent)
        if self.persistent and self.svd_sweeps:
            raise ValueError("Simulator: the persistent substep has the default decomposition only; svd='mcadams' runs on the cell and CSR forms")
        self._coop = None
        # cell_form: calc_elastic + collect_rhs_IP of a local/global iteration as one launch per kernel-grid cell chunk (pn_sim_stepforward_cells, 21
        # launches per substep instead of 31); PN_SIM_FORM=csr keeps the round-1-4 launch form (three launches per iteration over per-kernel CSR lists)
        self.cell_form = os.environ.get("PN_SIM_FORM", "cells") != "csr"
        bbox = bbox.clone() * 1.02   # solver.py:24-25 multiply in the caller's dtype (main_gui.py passes float32), then widen
        base = base.clone() * 1.01
        self.dt, self.iters, self.dx, self.kres, self.stiff = dt, iters, dx, kres, stiff
        bbox = bbox.to(dtype=torchfloat)
        self.res = (bbox // dx).to(dtype=torch.int32).to(self.device)
        self.base = base.to(dtype=torchfloat).to(self.device)
        self.gravity = gravity.to(dtype=torchfloat).to(self.device)
        self.dof = None
        self._work = None
        # stream on which update_force / clear_force are enqueued (None: the caller's current stream).  A harness that runs the substeps
        # on a stream of their own (harness.py: overlap_sim, capture_pipelined) sets it to that stream, so that a force change is ordered
        # BETWEEN two substeps instead of racing with one
        self.force_stream = None
        self.force_hooks = None   # (before, after): a pipeline whose substeps do not run on force_stream orders the change between two of them (frames.FramePipeline)

    # ------------------------------------------------------------------ IO (solver.py:109-137)
    def InitializeFromPly(self, path):
        c = scene.cloud_from_ply(path)
        self.InitializeFromArrays(c["pos"], c["mass"], c["mu"], c["lam"], c["pin"])

    def InitializeFromArrays(self, pos, mass, mu, lam, pin):
        dev = self.device
        self.pos = torch.from_numpy(np.asarray(pos, npfloat)).to(dev)
        assert self.pos.shape[0] > 0
        self.mass = torch.from_numpy(np.asarray(mass, npfloat)).to(dev)
        self.mu = torch.from_numpy(np.asarray(mu, npfloat)).to(dev)
        self.lam = torch.from_numpy(np.asarray(lam, npfloat)).to(dev)
        self.is_pin = torch.from_numpy(np.asarray(pin).astype(bool)).to(dev)
        if not bool((self.mass > 0).all()):  # collect_IP divides by the summed mass of every occupied cell (solver.py:450)
            raise ValueError("Simulator: every point needs mass > 0 (a PLY written by OutputToPly carries positions only)")
        self.initialize()

    def OutputToPly(self, path):
        """solver.py:109-113: the deformed point positions as a vertex element with double x, y, z only."""
        p = self.update_pos().cpu().numpy().astype(np.float64)
        scene.write_ply(path, dict(pos=p), props=("x", "y", "z"))

    # ------------------------------------------------------------------ init (solver.py:139-331)
    def initialize(self):
        self.precompute()
        self._work = torch.empty(int(lib().pn_sim_work_doubles(self.n_k, self.n_IP)), dtype=torchfloat, device=self.device)
        self._prepared = False  # pn_sim_prepare runs with the first substep (the CSR lists it reads are built further down)
        if self.cell_form:
            self._prepare_cells()
        self.rhs_rest = (self.build_rhs() + self._matvec(self.Mmat, self.dof)).contiguous()   # solver.py:314

    def precompute(self):
        """Everything of initialize() that is tensor bookkeeping / torch.linalg (device-agnostic); the HIP-backed rest
        state (rhs_rest) is finished by initialize()."""
        dev, res, kres = self.device, self.res, self.kres
        r0, r1, r2 = (int(v) for v in res.cpu())
        self.grid_idx = ((self.pos - self.base) // self.dx).to(dtype=torch.int32).long()
        gi = self.grid_idx
        self.IP_mask = torch.zeros((r0, r1, r2), dtype=torch.bool, device=dev)
        self.IP_mask[gi[:, 0], gi[:, 1], gi[:, 2]] = True
        n_IP = int(self.IP_mask.sum())
        self.IP_idx = -torch.ones((r0, r1, r2), dtype=torch.int32, device=dev)
        self.IP_idx[self.IP_mask] = torch.arange(0, n_IP, 1, dtype=torch.int32, device=dev)
        self.pts_IP = self.IP_idx[gi[:, 0], gi[:, 1], gi[:, 2]]
        # kornia.create_meshgrid3d + channel swap (solver.py:162-169) == grid[i,j,k] = (i,j,k)
        ax = [torch.arange(r, dtype=torch.int32, device=dev) for r in (r0, r1, r2)]
        cell_ijk = torch.stack(torch.meshgrid(*ax, indexing="ij"), dim=-1)
        self.IP_grid = cell_ijk[self.IP_mask, :]
        self.IP_pos = (self.IP_grid + 0.5) * self.dx + self.base          # float32 product, then float64 sum (:177)
        self.kernel_mask = torch.zeros((kres, kres, kres), dtype=torch.bool, device=dev)
        self.kdx = ((res.max()) * self.dx) / (kres - 1)                  # 0-dim float32 tensor (:184)
        IP2K = ((self.IP_pos - self.base) // self.kdx).to(dtype=torch.int32).long()
        corners = [(S >> 2 & 1, S >> 1 & 1, S & 1) for S in range(8)]
        for x, y, z in corners:
            self.kernel_mask[IP2K[:, 0] + x, IP2K[:, 1] + y, IP2K[:, 2] + z] |= True
        n_k = int(self.kernel_mask.sum())
        self.kernel_idx = torch.zeros((kres, kres, kres), dtype=torch.int32, device=dev)
        self.kernel_idx[self.kernel_mask] = torch.arange(0, n_k, 1, dtype=torch.int32, device=dev)
        pts2K = ((self.pos - self.base) // self.kdx).to(dtype=torch.int32).long()
        self.IP_kernel = torch.stack([self.kernel_idx[IP2K[:, 0] + x, IP2K[:, 1] + y, IP2K[:, 2] + z] for x, y, z in corners], dim=1).contiguous()
        self.pts_kernel = torch.stack([self.kernel_idx[pts2K[:, 0] + x, pts2K[:, 1] + y, pts2K[:, 2] + z] for x, y, z in corners], dim=1).contiguous()
        ka = torch.arange(kres, dtype=torch.int32, device=dev)
        self.kernel_grid = torch.stack(torch.meshgrid(ka, ka, ka, indexing="ij"), dim=-1)[self.kernel_mask, :]
        self.kernel_pos = self.kernel_grid * self.kdx + self.base       # float32 product, then float64 sum (:248)
        self.n_k, self.n_IP = n_k, n_IP

        kdx = float(self.kdx)
        self.pts_Nx, self.pts_dNx, self.pts_ddNx = gmls.init_GMLS(kdx, self.pos, self.pts_kernel, self.kernel_pos)
        self.IP_Nx, self.IP_dNx, self.IP_ddNx = gmls.init_GMLS(kdx, self.IP_pos, self.IP_kernel, self.kernel_pos)
        self.IP_mu, self.IP_lam, self.IP_rho = self.collect_IP()
        self.build_global()

        # rest state: translation = kernel position, affine = identity, quadratic = 0 (solver.py:258-275)
        dof = torch.zeros((n_k, 10, 3), dtype=torchfloat, device=dev)
        dof[:, 0, :] = self.kernel_pos
        for x in range(3):
            dof[:, 1 + x, x] = 1.0
        self.dof = dof.reshape(-1).contiguous()
        self.dof_tilde = self.dof.clone()
        self.dof_rest = self.dof.clone()
        self.dof_vel = torch.zeros_like(self.dof)
        self.dof_f = torch.zeros_like(self.dof)

        # per-kernel CSR of (IP, neighbour slot) pairs (count_IP_kernel / allocate_IP_kernel, solver.py:277-313), in ascending order
        keys = self.IP_kernel.reshape(-1).long()
        order = torch.sort(keys, stable=True).indices
        self.buffer = order.to(torch.int32).contiguous()                 # entry = vid*8 + dir
        self.kernel_cnt = torch.bincount(keys, minlength=n_k).to(torch.int32).contiguous()
        self.kernel_bg = (torch.cumsum(self.kernel_cnt, dim=0, dtype=torch.int32) - self.kernel_cnt).contiguous()
        self.tot = int(self.kernel_cnt.sum())
        # dNx rows in CSR order (6.9 MB on the chair): the step driver's collect_rhs then reads each kernel's entries as one
        # contiguous stream instead of chasing `buffer` (pn_sim_stepforward, dNx_csr)
        self.dNx_csr = self.IP_dNx.reshape(n_IP * 8, 30)[order].contiguous()
        self.csr_pos = torch.empty_like(self.buffer)
        self.csr_pos[order] = torch.arange(order.numel(), dtype=torch.int32, device=dev)   # inverse of `buffer`

        self._IP2K = IP2K
        self._cells = self._cells_work = None
        if self.cell_form:   # (torch bookkeeping only; the work area and the check against the library's chunk size: _prepare_cells)
            self._build_cells()

        m = (self.IP_rho * self.dx * self.dx * self.dx)                                      # collect_gravity, cuda_utils.py:262-279
        rg = torch.zeros((n_k * 10, 3), dtype=torchfloat, device=dev)
        rows = (self.IP_kernel.long()[:, :, None] * 10 + torch.arange(10, device=dev)[None, None, :]).reshape(-1)
        gmls.index_add_ordered(rg, rows, (m[:, None, None] * self.IP_Nx).reshape(-1)[:, None] * self.gravity[None, :])
        self.rhs_gravity = rg.reshape(-1).contiguous()

    def _build_cells(self):
        """Layout of the substep's CELL form (include/pienerf_hip.h: pn_sim_stepforward_cells): the integration points of one kernel-grid cell share
        their 8 neighbour kernels (IP_kernel rows are equal, solver.py:186-205), so they are sorted by cell and every cell is cut into chunks of at most
        pn_sim_cells_chunk_ips() points — one workgroup each, computing calc_elastic and the points' contributions to collect_rhs_IP in one launch."""
        dev, n_IP, n_k, kres = self.device, self.n_IP, self.n_k, self.kres
        B = CELL_CHUNK_IPS
        cell = (self._IP2K[:, 0] * kres + self._IP2K[:, 1]) * kres + self._IP2K[:, 2]
        order = torch.sort(cell, stable=True).indices                                  # points by cell, ascending point index inside a cell
        cs = cell[order]
        first = torch.ones(n_IP, dtype=torch.bool, device=dev)
        first[1:] = cs[1:] != cs[:-1]
        start = torch.nonzero(first).reshape(-1)                                       # where each cell begins in the sorted order
        cnt = torch.diff(torch.cat([start, torch.tensor([n_IP], device=dev)]))
        nch = (cnt + B - 1) // B                                                       # chunks per cell
        n_chunks = int(nch.sum())
        ch_cell = torch.repeat_interleave(torch.arange(len(cnt), device=dev), nch)     # chunk -> cell
        ch_first = torch.cumsum(nch, 0) - nch                                          # first chunk of each cell
        ch_j = torch.arange(n_chunks, device=dev) - ch_first[ch_cell]                  # chunk's index inside its cell
        ch_begin = start[ch_cell] + ch_j * B                                           # first point (sorted order) of the chunk
        ch_count = torch.minimum(cnt[ch_cell] - ch_j * B, torch.tensor(B, device=dev))
        local = torch.arange(B, device=dev)[None, :]
        valid = local < ch_count[:, None]                                              # [n_chunks, B]
        src = order[torch.clamp(ch_begin[:, None] + local, max=n_IP - 1)]              # original point index per chunk position
        topo = self.IP_kernel.long()
        assert bool((topo[src[valid]] == topo[src[:, :1].expand(-1, B)[valid]]).all()), "points of one kernel-grid cell must share their 8 kernels"
        tab = torch.zeros((n_chunks, 12), dtype=torch.int32, device=dev)
        tab[:, 0] = ch_count.to(torch.int32)
        tab[:, 1:9] = topo[src[:, 0]].to(torch.int32)
        g = self.IP_dNx.reshape(n_IP, 8, 30)[src] * valid[:, :, None, None].to(torchfloat)   # [n_chunks, B, 8, 30]
        # lane l of wave w = point w * 8 + l // 8, slot l % 8: [chunk][wave][15][64][2]
        g = g.reshape(n_chunks, B // 8, 8, 8, 15, 2).permute(0, 1, 4, 2, 3, 5).reshape(n_chunks, B // 8, 15, 64, 2)
        self._cells = dict(n_chunks=n_chunks, B=B, tab=tab.contiguous(), dNx=g.contiguous(),
                           mu=(self.IP_mu[src] * valid).reshape(-1).contiguous(), lam=(self.IP_lam[src] * valid).reshape(-1).contiguous(), src=src, valid=valid)
        # per kernel: the (chunk, slot) pairs that refer to it, ascending
        keys = tab[:, 1:9].reshape(-1).long()
        kp = torch.sort(keys, stable=True).indices
        kcnt = torch.bincount(keys, minlength=n_k)
        self._cells["kp_list"] = kp.to(torch.int32).contiguous()
        pos = torch.empty_like(kp)
        pos[kp] = torch.arange(kp.numel(), device=dev)
        self._cells["kp_pos"] = pos.to(torch.int32).contiguous()                       # where (chunk, slot) stores its partial sum: its rank in its kernel's run
        self._cells["kp_bg"] = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), torch.cumsum(kcnt, 0)]).to(torch.int32).contiguous()
        self._cells_work = None   # belongs to the layout: _prepare_cells() makes a new one (initialize(), or the first substep after a rebuild)

    def _prepare_cells(self):
        """The cell form's work area: identity rotations for the warm-started SVD, arrival counters (never inside a stream capture)."""
        if int(lib().pn_sim_cells_chunk_ips()) != CELL_CHUNK_IPS:
            raise RuntimeError(f"libpienerf_hip.so cuts cells into chunks of {lib().pn_sim_cells_chunk_ips()} points, solver.py into {CELL_CHUNK_IPS}")
        n_chunks = self._cells["n_chunks"]
        self._cells_work = torch.empty(int(lib().pn_sim_cells_work_doubles(self.n_k, n_chunks)), dtype=torchfloat, device=self.device)
        check(lib().pn_sim_cells_prepare(self.n_k, n_chunks, ptr(self._cells_work), stream_ptr()), "sim_cells_prepare")

    def reset_warm_start(self):
        """Forget the SVD warm start (the V of every integration point's previous local/global iteration, kept in the cell form's work area).  Results
        do not depend on it beyond the decomposition's stopping rule (off-diagonals <= 1e-11 of the diagonal: 1e-10 relative on the displacements), but
        BIT-equal replays of a trajectory do: whoever restores dof / dof_vel to replay (harness.capture, tests) calls this as well."""
        if self._cells_work is not None:
            check(lib().pn_sim_cells_prepare(self.n_k, self._cells["n_chunks"], ptr(self._cells_work), stream_ptr()), "sim_cells_prepare")
        if self._work is not None and self._prepared:
            check(lib().pn_sim_prepare(self.n_k, self.n_IP, ptr(self.kernel_bg), ptr(self.kernel_cnt), ptr(self._work), stream_ptr()), "sim_prepare")

    def collect_IP(self):  # solver.py:427-450
        n_IP, idx = self.n_IP, self.pts_IP.long()
        z = torch.zeros(n_IP, dtype=torchfloat, device=self.device)
        s_mu = gmls.index_add_ordered(z.clone(), idx, self.mu * self.mass)
        s_lam = gmls.index_add_ordered(z.clone(), idx, self.lam * self.mass)
        s_m = gmls.index_add_ordered(z.clone(), idx, self.mass)
        return (s_mu / s_m).contiguous(), (s_lam / s_m).contiguous(), (s_m / (self.dx ** 3)).contiguous()

    def build_global(self):  # solver.py:453-538
        dim = self.n_k * 10
        mat = gmls.assemble_IP_matrix(dim, self.dx, self.dt, self.IP_kernel, self.IP_mu, self.IP_lam, self.IP_rho, self.IP_Nx, self.IP_dNx, self.IP_ddNx)
        assert self.pts_kernel.min() >= 0 and self.pts_kernel.max() < self.n_k
        vid = torch.nonzero(self.is_pin).reshape(-1)
        mat = gmls.add_pin_penalty(mat, self.stiff, vid, self.pts_kernel, self.pts_Nx)
        diag = mat.diagonal()[0::10]
        self.active_kernels = torch.nonzero(diag > 0.0).reshape(-1)                      # `global_matrix[i*30, i*30] > 0` (:499-504)
        lst = (self.active_kernels[:, None] * 10 + torch.arange(10, device=self.device)[None, :]).reshape(-1)
        sub = mat[lst][:, lst].clone()
        sub.diagonal().add_(1e-3)                                                         # :507
        inv = gmls.inverse_spd(sub)
        self.Ainv = torch.zeros((dim, dim), dtype=torchfloat, device=self.device)
        self.Ainv[lst[:, None], lst[None, :]] = inv
        self.Ainv = self.Ainv.contiguous()
        zero = torch.zeros_like(self.IP_mu)
        self.Mmat = gmls.assemble_IP_matrix(dim, self.dx, self.dt, self.IP_kernel, zero, zero, self.IP_rho, self.IP_Nx, self.IP_dNx, self.IP_ddNx).contiguous()

    # the reference's (30 n_k)^2 forms, materialised on demand (tests / interop only)
    @property
    def global_matrix(self):
        return torch.kron(self.Ainv, torch.eye(3, dtype=torchfloat, device=self.device))

    @property
    def mass_matrix_invt2(self):
        return torch.kron(self.Mmat, torch.eye(3, dtype=torchfloat, device=self.device))

    # ------------------------------------------------------------------ per-frame (HIP)
    def _matvec(self, A, x):
        y = torch.empty_like(x)
        check(lib().pn_sim_matvec3(A.shape[0], ptr(A), ptr(x), ptr(y), stream_ptr()), "sim_matvec3")
        return y

    def build_rhs(self):  # solver.py:541-571
        n = self.n_IP
        RF = torch.empty((n, 3, 3), dtype=torchfloat, device=self.device)
        VF = torch.empty_like(RF)
        check(lib().pn_sim_set_svd(self.svd_sweeps), "sim_set_svd")
        check(lib().pn_sim_calc_elastic(n, ptr(self.IP_kernel), ptr(self.IP_dNx), ptr(self.dof), ptr(RF), ptr(VF), None, stream_ptr()), "calc_elastic")
        rhs = torch.empty_like(self.dof)
        check(lib().pn_sim_collect_rhs(self.n_k, float(self.dx), ptr(self.kernel_bg), ptr(self.kernel_cnt), ptr(self.buffer), ptr(self.IP_mu),
                                       ptr(self.IP_lam), ptr(self.IP_dNx), ptr(RF), ptr(VF), ptr(rhs), stream_ptr()), "collect_rhs")
        return rhs

    def get_IP_info(self, dof=None, out=None):  # solver.py:402-424
        """(IP_pos, IP_F, IP_dF) of the current state.  `dof` (a [30 n_k] fp64 snapshot of self.dof) and `out` (three preallocated
        fp32 tensors) are extensions for the pipelined harness: frames in flight each own a snapshot and a set of IP buffers."""
        n, dev = self.n_IP, self.device
        if out is None:
            pos = torch.empty((n, 3), dtype=torch.float32, device=dev)
            F = torch.empty((n, 9), dtype=torch.float32, device=dev)
            dF = torch.empty((n, 27), dtype=torch.float32, device=dev)
        else:
            pos, F, dF = out
        check(lib().pn_sim_update_F(n, ptr(self.IP_kernel), ptr(self.dof if dof is None else dof), ptr(self.IP_Nx), ptr(self.IP_dNx), ptr(self.IP_ddNx),
                                    ptr(pos), ptr(F), ptr(dF), stream_ptr()), "update_F")
        return pos, F, dF

    def _prepare_persistent(self):
        """Lays out the persistent kernel's pieces (one host read-back of the CSR counts); falls back to the launch form when the scene does not fit."""
        # one workgroup per CU, minus a few CUs left to whatever else runs on the device meanwhile (a collective's kernels, a copy kernel): a
        # persistent workgroup takes a CU's whole register file, and with no CU to spare the launch would wait for those kernels to end
        n_wg = min(max(int(lib().pn_device_cu_count()) - int(os.environ.get("PN_SIM_COOP_RESERVE", "8")), 8), 256)
        nbytes = int(lib().pn_sim_coop_bytes(self.n_k, self.n_IP, n_wg))
        self._coop = None
        if nbytes == 0:
            self.persistent = False
            return
        buf = torch.zeros(nbytes, dtype=torch.uint8, device=self.device)
        plan = (C.c_int32 * 3)()
        rc = lib().pn_sim_coop_prepare(self.n_k, self.n_IP, n_wg, ptr(self.kernel_bg), ptr(self.kernel_cnt), ptr(buf), plan, stream_ptr())
        if rc != 0:  # PN_ERR_ARG: lists too long for register-resident pieces
            self.persistent = False
            return
        self._coop = (buf, n_wg, plan)

    def enable_persistent(self):
        """Switch to the persistent substep (a harness calls this for a GPU that only simulates: the owner rank of a frame-parallel job with a
        dedicated simulator).  PN_SIM_COOP=0 vetoes it; scenes that do not fit keep the launch form.  Returns whether it is on."""
        if os.environ.get("PN_SIM_COOP", "") == "0":
            return False
        self.persistent = True
        if getattr(self, "_prepared", False) and self._coop is None:
            self._prepare_persistent()
        return self.persistent and (self._coop is not None or not getattr(self, "_prepared", False))

    def persistent_timed_out(self):
        """True if a persistent substep gave up waiting at a device-wide barrier (its workgroups could not all become resident): results invalid."""
        if self._coop is None:
            return False
        flag = C.c_int32(0)
        check(lib().pn_sim_coop_status(ptr(self._coop[0]), C.byref(flag)), "sim_coop_status")
        return flag.value != 0

    def stepforward(self):  # solver.py:595-602
        check(lib().pn_sim_set_svd(self.svd_sweeps), "sim_set_svd")   # process-global in the library: every enqueue names its own choice
        if not self._prepared:
            check(lib().pn_sim_prepare(self.n_k, self.n_IP, ptr(self.kernel_bg), ptr(self.kernel_cnt), ptr(self._work), stream_ptr()), "sim_prepare")
            self._prepared = True
            if self.persistent:
                self._prepare_persistent()
        if self.persistent and self._coop is not None and 1 <= self.iters <= 32:
            buf, n_wg, plan = self._coop
            check(lib().pn_sim_stepforward_coop(self.n_k, self.n_IP, int(self.iters), float(self.dt), float(self.dx), ptr(self.IP_kernel), ptr(self.IP_mu),
                                                ptr(self.IP_lam), ptr(self.IP_dNx), ptr(self.dNx_csr), ptr(self.csr_pos), ptr(self.Ainv), ptr(self.Mmat),
                                                ptr(self.dof_rest), ptr(self.rhs_rest), ptr(self.rhs_gravity), ptr(self.dof_f), ptr(self.dof), ptr(self.dof_vel),
                                                ptr(self._work), ptr(buf), n_wg, plan, stream_ptr()), "stepforward_coop")
            return
        if self.cell_form and self._cells is not None and int(self.iters) >= 1:
            if self._cells_work is None:   # precompute() ran again since initialize(): the layout is new, so is its work area
                self._prepare_cells()
            c = self._cells
            check(lib().pn_sim_stepforward_cells(self.n_k, c["n_chunks"], int(self.iters), float(self.dt), float(self.dx), ptr(c["tab"]), ptr(c["dNx"]),
                                                 ptr(c["mu"]), ptr(c["lam"]), ptr(c["kp_bg"]), ptr(c["kp_pos"]), ptr(self.Ainv), ptr(self.Mmat),
                                                 ptr(self.dof_rest), ptr(self.rhs_rest), ptr(self.rhs_gravity), ptr(self.dof_f), ptr(self.dof), ptr(self.dof_vel),
                                                 ptr(self._cells_work), stream_ptr()), "stepforward_cells")
            return
        check(lib().pn_sim_stepforward(self.n_k, self.n_IP, int(self.iters), float(self.dt), float(self.dx), ptr(self.IP_kernel), ptr(self.kernel_bg),
                                       ptr(self.kernel_cnt), ptr(self.buffer), ptr(self.IP_mu), ptr(self.IP_lam), ptr(self.IP_dNx), ptr(self.dNx_csr), ptr(self.csr_pos), ptr(self.Ainv),
                                       ptr(self.Mmat), ptr(self.dof_rest), ptr(self.rhs_rest), ptr(self.rhs_gravity), ptr(self.dof_f), ptr(self.dof),
                                       ptr(self.dof_vel), ptr(self._work), 1, stream_ptr()), "stepforward")

    step = stepforward  # BASELINE.json's name for the same entry point

    def _force_launch(self, vid, f3):
        st = self.force_stream
        if st is not None:  # ordered between two substeps of the simulator's own stream, after whatever the caller has enqueued so far
            st.wait_stream(torch.cuda.current_stream(self.device))
            if self.force_hooks:
                self.force_hooks[0]()
        with torch.cuda.stream(st) if st is not None else _null_ctx():
            check(lib().pn_sim_update_force(self.n_k, int(vid), f3.ctypes.data if f3 is not None else None, float(self.dx), ptr(self.IP_kernel),
                                            ptr(self.IP_rho), ptr(self.IP_Nx), ptr(self.dof_f), stream_ptr()), "update_force")
        if st is not None:
            # ... and before whatever the caller enqueues next on ITS stream: a substep launched there (sim.stepforward(), a whole-step graph) must
            # not read dof_f while the kernel above is still writing it
            ev = torch.cuda.Event()
            ev.record(st)
            torch.cuda.current_stream(self.device).wait_event(ev)
            if self.force_hooks:
                self.force_hooks[1]()

    def update_force(self, vid, f):  # solver.py:578-588
        """dof_f = the pick force `f` on IP `vid`, written whole by one launch on `force_stream` (or the current stream): it acts from the
        next substep enqueued after this call."""
        f3 = np.ascontiguousarray(f.detach().cpu().numpy() if torch.is_tensor(f) else f, dtype=np.float64)
        assert 0 <= int(vid) < self.n_IP and f3.shape == (3,)
        self._force_launch(vid, f3)

    def clear_force(self):  # solver.py:590-593
        self._force_launch(-1, None)

    def update_pos(self):  # solver.py:604-617 (update_pos_kernel) — only used by OutputToPly
        d = self.dof.view(self.n_k, 10, 3)[self.pts_kernel.long()]
        self.pos = torch.einsum("nic,nicr->nr", self.pts_Nx, d)
        return self.pos
