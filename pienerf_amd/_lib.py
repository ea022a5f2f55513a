"""ctypes binding of libpienerf_hip.so (include/pienerf_hip.h).

There is NO CPU fallback: if the library is missing or a symbol is absent this module raises at
first use, and every op raises RuntimeError when handed a non-GPU tensor.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PN_LIB_PATH") or os.path.join(_HERE, "lib", "libpienerf_hip.so")  # PN_LIB_PATH: tuning builds (tools/build_variant.py)

P = C.c_void_p
u32, i32, f32, f64, u64 = C.c_uint32, C.c_int, C.c_float, C.c_double, C.c_uint64


class RenderOpts(C.Structure):
    """pn_render_opts (include/pienerf_hip.h)."""
    _fields_ = [("max_iter_num", i32), ("hash_grid_size", f32), ("num_seek_IP", i32), ("IP_dx", f32), ("cut", i32), ("cut_bounds", f32 * 6),
                ("bound", f32), ("min_near", f32), ("dt_gamma", f32), ("max_steps", u32), ("T_thresh", f32), ("cascade", u32), ("grid_size", u32),
                ("density_scale", f32), ("bg_color", f32), ("fp16", i32), ("reuse_tables", i32), ("ray_batch", i32), ("throughput", i32), ("throughput_trips", i32), ("ray_tile_w", i32),
                ("fused_from", i32), ("fused_whole", i32), ("fused_fold", i32), ("fused_grid", i32)]


# name -> (restype, argtypes); every function declared in include/pienerf_hip.h
SIGNATURES = {
    "pn_version": (C.c_char_p, []),
    "pn_last_error": (C.c_char_p, []),
    "pn_stream_create_cu_mask": (i32, [u32, u32, u32, i32, C.POINTER(C.c_void_p)]),
    "pn_stream_destroy": (i32, [P]),
    "pn_device_cu_count": (i32, []),
    "pn_march_set_tail_rounds": (i32, [i32]),
    "pn_march_set_skip_dda": (i32, [i32]),
    "pn_copier_create": (i32, [C.POINTER(P)]),
    "pn_copier_destroy": (None, [P]),
    "pn_copier_submit": (i32, [P, P, P, C.c_uint64, P, C.POINTER(C.c_uint64)]),
    "pn_copier_wait": (i32, [P, C.c_uint64]),
    "pn_near_far_from_aabb": (i32, [P, P, P, u32, f32, P, P, P]),
    "pn_sph_from_ray": (i32, [P, P, f32, u32, P, P]),
    "pn_march_rays_quadratic_bending": (i32, [P, P, P, i32, i32, P, P, P, P, i32, P, P, f32, P, i32, f32, i32, P, u32, u32, P, P, P, P, f32, f32, u32,
                                              u32, u32, P, P, P, P, P, P, P, P, P]),
    "pn_composite_rays": (i32, [u32, u32, f32, P, P, P, P, P, P, P, P, P]),
    "pn_compact_rays": (i32, [P, u32, P, P, P, P]),
    "pn_compact_scratch_ints": (u32, [u32]),
    "pn_pnts_in_grids": (i32, [i32, i32, P, P, f32, P, P, P, P, P, P]),
    "pn_get_rays": (i32, [P, f32, f32, f32, f32, i32, i32, P, P, P]),
    "pn_grid_encode_forward": (i32, [P, P, P, P, u32, u32, u32, u32, f32, u32, P, u32, i32, u32, i32, P]),
    "pn_sh_encode_forward": (i32, [P, P, u32, u32, u32, P, P]),
    "pn_net_create": (i32, [C.POINTER(P), P, P, u32, u32, f32, u32, f32, P, P, P, P, P, P]),
    "pn_net_destroy": (None, [P]),
    "pn_net_update": (i32, [P, P, P, P, P, P, P, P]),
    "pn_net_enable_half": (i32, [P, P]),
    "pn_nerf_forward_half": (i32, [P, P, P, u32, f32, P, P, P]),
    "pn_nerf_density_half": (i32, [P, P, u32, P, P, P]),
    "pn_nerf_sigma": (i32, [P, P, u32, f32, P, i32, P]),
    "pn_host_float_to_half": (i32, [P, P, u32]),
    "pn_grid_encode_forward_half": (i32, [P, P, P, P, u32, u32, u32, u32, f32, u32, u32, i32, u32, i32, P]),
    "pn_march_rays_train": (i32, [P, P, P, f32, f32, u32, u32, u32, u32, u32, P, P, P, P, P, P, P, P, P]),
    "pn_composite_rays_train_forward": (i32, [P, P, P, P, u32, u32, f32, P, P, P, P]),
    "pn_composite_rays_train_backward": (i32, [P, P, P, P, P, P, P, P, u32, u32, f32, P, P, P]),
    "pn_grid_encode_backward": (i32, [P, P, P, P, P, u32, u32, u32, u32, f32, u32, P, P, u32, i32, u32, P]),
    "pn_grid_encode_backward_half": (i32, [P, P, P, P, u32, u32, u32, u32, f32, u32, u32, i32, u32, P]),
    "pn_grad_total_variation": (i32, [P, P, P, P, f32, u32, u32, u32, u32, f32, u32, u32, i32, P]),
    "pn_sh_encode_backward": (i32, [P, P, u32, u32, u32, P, P, P]),
    "pn_march_rays": (i32, [u32, u32, P, P, P, P, f32, f32, u32, u32, u32, P, P, P, P, P, P, P, P]),
    "pn_packbits": (i32, [P, u32, f32, P, P]),
    "pn_morton3D": (i32, [P, u32, P, P]),
    "pn_morton3D_invert": (i32, [P, u32, P, P]),
    "pn_nerf_forward": (i32, [P, P, P, u32, f32, P, P, P]),
    "pn_nerf_density": (i32, [P, P, u32, P, P, P]),
    "pn_frame_create": (i32, [C.POINTER(P), u32, u32, u32]),
    "pn_frame_destroy": (None, [P]),
    "pn_render_deformed": (i32, [P, P, C.POINTER(RenderOpts), P, P, u32, P, P, P, P, i32, P, P, P, P, P, P, P]),
    "pn_render_deformed_async": (i32, [P, P, C.POINTER(RenderOpts), P, P, u32, P, P, P, P, i32, P, P, P, P, P, i32, P]),
    "pn_render_status": (i32, [P, P, i32, P]),
    "pn_frame_reset_unfinished": (i32, [P, P]),
    "pn_render_continue": (i32, [P, P, C.POINTER(RenderOpts), P, P, u32, P, P, P, P, P, P, i32, i32, P]),
    "pn_render_static": (i32, [P, P, C.POINTER(RenderOpts), P, P, u32, P, P, P, P, P, P, P, i32, P]),
    "pn_mark_untrained_grid": (i32, [P, u32, f32, f32, f32, f32, u32, u32, f32, P, P, P]),
    "pn_density_cells_full": (i32, [u32, u32, f32, P, P, P]),
    "pn_density_cells_partial": (i32, [u32, u32, f32, u32, P, P, P, P, P, P, P, P, P]),
    "pn_density_partial_scratch_ints": (u64, [u32]),
    "pn_density_scatter": (i32, [u32, P, P, P, P]),
    "pn_density_grid_update": (i32, [u32, P, P, f32, f32, P, P, P, P]),
    "pn_frame_march_counters": (i32, [P, i32, P, P]),
    "pn_frame_trip_times": (i32, [P, P, P, i32, P, P]),
    "pn_frame_fused_clocks": (i32, [P, P, P, i32, P]),
    "pn_net_form": (i32, [P]),
    "pn_net_form_epoch": (i32, [P]),
    "pn_frame_trip_records": (i32, [P, P, P, i32, P]),
    "pn_sim_update_F": (i32, [i32, P, P, P, P, P, P, P, P, P]),
    "pn_sim_calc_elastic": (i32, [i32, P, P, P, P, P, P, P]),
    "pn_sim_collect_rhs": (i32, [i32, f64, P, P, P, P, P, P, P, P, P, P]),
    "pn_sim_matvec3": (i32, [i32, P, P, P, P]),
    "pn_sim_stepforward": (i32, [i32, i32, i32, f64, f64, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, i32, P]),
    "pn_sim_prepare": (i32, [i32, i32, P, P, P, P]),
    "pn_sim_set_svd": (i32, [i32]),
    "pn_sim_get_svd": (i32, []),
    "pn_sim_cells_chunk_ips": (i32, []),
    "pn_sim_cells_work_doubles": (u64, [i32, i32]),
    "pn_sim_cells_prepare": (i32, [i32, i32, P, P]),
    "pn_sim_stepforward_cells": (i32, [i32, i32, i32, f64, f64, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P]),
    "pn_sim_work_doubles": (u64, [i32, i32]),
    "pn_sim_coop_bytes": (u64, [i32, i32, i32]),
    "pn_sim_coop_prepare": (i32, [i32, i32, i32, P, P, P, C.POINTER(i32), P]),
    "pn_sim_coop_status": (i32, [P, C.POINTER(i32)]),
    "pn_sim_coop_clocks": (i32, [P, C.POINTER(u64)]),
    "pn_sim_stepforward_coop": (i32, [i32, i32, i32, f64, f64, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, i32, C.POINTER(i32), P]),
    "pn_sim_update_force": (i32, [i32, i32, P, f64, P, P, P, P, P]),
}

_lib = None


def lib():
    """The loaded library with argtypes set.  Raises if it was not built (python -m pienerf_amd.build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: build it with `python -m pienerf_amd.build` (hipcc, gfx950). There is no CPU fallback.")
        # One HIP runtime per process: torch ships its own libamdhip64.so.7 and device memory / streams come from torch, so
        # torch must be loaded first — the library's NEEDED libamdhip64.so.7 then binds to that already-loaded copy instead of
        # pulling in /opt/rocm's second runtime (which would see "no ROCm-capable device").
        import torch  # noqa: F401
        h = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(h, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = h
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().pn_last_error().decode(errors="replace")
        raise RuntimeError(f"libpienerf_hip: {what} failed with code {rc}: {msg}")


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("pienerf_amd ops run on the GPU only (HIP kernels); got a CPU tensor and there is no CPU fallback")


def ptr(t):
    """Device pointer of a contiguous torch tensor (or None)."""
    if t is None:
        return None
    assert t.is_contiguous(), "tensor must be contiguous"
    return C.c_void_p(t.data_ptr())


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
