"""``get_rays`` and ``get_pnts_in_grids`` with the reference's signatures (nerf/utils.py:54-138, 355-386),
backed by HIP kernels.  The rest of the reference's nerf/utils.py (metrics, meshing, seeding) is off-path."""
import numpy as np
import torch

from .._lib import check, lib, ptr, require_gpu, stream_ptr


def get_rays(poses, intrinsics, H, W, N=-1, error_map=None, patch_size=1):
    """nerf/utils.py:54-138, inference form (N = -1): poses [1,4,4] cam2world, intrinsics (fx, fy, cx, cy).

    Returns {'rays_o': [1,H*W,3], 'rays_d': [1,H*W,3]} fp32 on the pose's device."""
    if N > 0 or error_map is not None or patch_size != 1:
        raise RuntimeError("get_rays: only the full-image inference form (N=-1) is on the simulate-and-render path")
    if poses.shape[0] != 1:
        raise RuntimeError("get_rays: one pose per call (the GUI / render harness passes [1,4,4])")
    require_gpu(poses)
    fx, fy, cx, cy = (float(v) for v in intrinsics)
    pose = poses[0].detach().to(torch.float32).contiguous()
    rays_o = torch.empty(1, H * W, 3, dtype=torch.float32, device=poses.device)
    rays_d = torch.empty(1, H * W, 3, dtype=torch.float32, device=poses.device)
    check(lib().pn_get_rays(ptr(pose), fx, fy, cx, cy, int(H), int(W), ptr(rays_o), ptr(rays_d), stream_ptr()), "get_rays")
    return {"rays_o": rays_o, "rays_d": rays_d}


def get_pnts_in_grids(n_vtx, n_grid, pnts, bbmin, bbmax, hgs, resolution):
    """nerf/utils.py:355-386: counting sort of the deformed IPs into `hgs` cells -> (pig_cnt, pig_bgn, pig_idx) int32.

    Slots inside a cell are in ascending point id (the reference's order is an atomic race)."""
    require_gpu(pnts, bbmin, resolution)
    n_vtx, n_grid = int(n_vtx), int(n_grid)
    pnts = pnts.to(torch.float32).contiguous()
    bbmin = bbmin.to(torch.float32).contiguous()
    resolution = resolution.to(torch.int32).contiguous()
    dev = pnts.device
    pig_idx = torch.zeros((n_vtx,), dtype=torch.int32, device=dev)
    pig_cnt = torch.zeros((n_grid,), dtype=torch.int32, device=dev)
    pig_bgn = torch.zeros((n_grid,), dtype=torch.int32, device=dev)
    err = torch.zeros(1, dtype=torch.int32, device=dev)
    check(lib().pn_pnts_in_grids(n_vtx, n_grid, ptr(pnts), ptr(bbmin), float(hgs), ptr(resolution), ptr(pig_cnt), ptr(pig_bgn), ptr(pig_idx),
                                 ptr(err), stream_ptr()), "get_pnts_in_grids")
    return pig_cnt, pig_bgn, pig_idx
