#!/usr/bin/env python
"""The global step of a local/global iteration in its two forms (simulator/solver.py:505-511 pre-inverts the system matrix; SURVEY §8f rank 1 asks whether a
Cholesky factor with two triangular solves per iteration would do): x = A^-1 b as ONE dense product with the explicit inverse (what pn_sim.hip's k_matvec3
does: [10 n_k]^2 fp64, 3 right-hand sides) against x = L^-T (L^-1 b) through rocBLAS trsm, at n_k = 139 (chair) and 343 (sim_dx 0.035).
    python tools/time_trisolve.py"""
import torch

dev = "cuda:0"
torch.manual_seed(0)
for n_k in (139, 343):
    n = 10 * n_k
    M = torch.randn(n, n, dtype=torch.float64, device=dev)
    A = M @ M.T + n * torch.eye(n, dtype=torch.float64, device=dev)
    L = torch.linalg.cholesky(A)
    Ainv = torch.cholesky_inverse(L)
    b = torch.randn(n, 3, dtype=torch.float64, device=dev)
    x1 = Ainv @ b
    x2 = torch.cholesky_solve(b, L)
    y = torch.linalg.solve_triangular(L, b, upper=False)
    x3 = torch.linalg.solve_triangular(L.T.contiguous(), y, upper=True)
    err2 = float((x1 - x2).abs().max() / x1.abs().max())
    err3 = float((x1 - x3).abs().max() / x1.abs().max())

    def timed(fn, reps=200):
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3

    LT = L.T.contiguous()
    t_mm = timed(lambda: torch.matmul(Ainv, b))
    t_cs = timed(lambda: torch.cholesky_solve(b, L))
    t_ts = timed(lambda: torch.linalg.solve_triangular(LT, torch.linalg.solve_triangular(L, b, upper=False), upper=True))
    print(f"n_k {n_k} (n = {n}): explicit inverse, one product (torch.matmul) {t_mm:.1f} us | cholesky_solve {t_cs:.1f} us | two solve_triangular {t_ts:.1f} us "
          f"| bytes: inverse {n * n * 8 / 1e6:.1f} MB, factor {n * (n + 1) * 4 / 1e6:.1f} MB | agreement {err2:.1e} / {err3:.1e}")
