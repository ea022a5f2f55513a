// Index / constant helpers shared by the encoder kernels (pn_nerf_forward.hip: forward + fused network; pn_encoder_grad.hip: dy_dx,
// backward, total variation).  gfx950 only.
#pragma once
#include "pn_common.h"

// get_grid_index for D = 3 (gridencoder.cu:65-84); `dense` = 0 -> fast_hash, else number of strided dims.
struct LevelIdx { uint32_t dense, hs, mask, nomod, stride1; };
__device__ __forceinline__ LevelIdx level_idx(const PnGridLevels& lv, uint32_t level, int align_corners) {
    return LevelIdx{lv.dense[level], lv.hashmap_size[level], lv.mask[level], lv.nomod[level],
                    align_corners ? lv.resolution[level] : lv.resolution[level] + 1};
}
__device__ __forceinline__ uint32_t grid_index3(const LevelIdx& L, uint32_t g0, uint32_t g1, uint32_t g2) {
    if (L.dense == 0) {
        const uint32_t index = g0 ^ (g1 * 2654435761u) ^ (g2 * 805459861u);
        return L.mask ? (index & L.mask) : (index % L.hs);
    }
    const uint32_t index = g0 + (L.dense > 1 ? g1 * L.stride1 : 0u) + (L.dense > 2 ? g2 * L.stride1 * L.stride1 : 0u);
    return L.nomod ? index : (index % L.hs);
}


// Real SH basis in the reference's sign convention (shencoder.cu:50-68); constants are the closed forms of its comments.
#define SH_C0 0.28209479177387814f   /* 1/(2 sqrt(pi)) */
#define SH_C1 0.48860251190291992f   /* sqrt(3)/(2 sqrt(pi)) */
#define SH_C2A 1.0925484305920792f   /* sqrt(15)/(2 sqrt(pi)) */
#define SH_C2B 0.94617469575755997f  /* 3 sqrt(5)/(4 sqrt(pi)) */
#define SH_C2C 0.31539156525251999f  /* sqrt(5)/(4 sqrt(pi)) */
#define SH_C2D 0.54627421529603959f  /* sqrt(15)/(4 sqrt(pi)) */
#define SH_C3A 0.59004358992664352f  /* sqrt(70)/(8 sqrt(pi)) */
#define SH_C3B 2.8906114426405538f   /* sqrt(105)/(2 sqrt(pi)) */
#define SH_C3C 0.45704579946446572f  /* sqrt(42)/(8 sqrt(pi)) */
#define SH_C3D 0.3731763325901154f   /* sqrt(7)/(4 sqrt(pi)) */
#define SH_C3E 1.4453057213202769f   /* sqrt(105)/(4 sqrt(pi)) */


// training-side launchers (pn_encoder_grad.hip) the forward entry points chain into when dy_dx is requested
int pn_grid_dy_dx_launch(const float* inputs, const float* embeddings, const PnGridLevels& lv, uint32_t B, uint32_t C, int align_corners, uint32_t interp,
                         float* dy_dx, hipStream_t st);
int pn_sh_dy_dx_launch(const float* inputs, float* dy_dx, uint32_t B, uint32_t C, hipStream_t st);
// input dimensions 2, 4, 5 of the stand-alone grid op (pn_grid_nd.hip; gridencoder.cu:386-399,430-444): fp32, forward (+ dy_dx) and backward
int pn_grid_nd_forward_launch(const float* inputs, const float* embeddings, const int* offsets_host, float* outputs, uint32_t B, uint32_t D, uint32_t C,
                              uint32_t L, float S, uint32_t H, float* dy_dx, uint32_t gridtype, int align_corners, uint32_t interp, int out_bl_major,
                              hipStream_t st);
int pn_grid_nd_grad_tv_launch(const float* inputs, const float* embeddings, float* grad, const int* offsets_host, float weight, uint32_t B, uint32_t D, uint32_t C,
                              uint32_t L, float S, uint32_t H, uint32_t gridtype, int align_corners, hipStream_t st);
int pn_grid_nd_backward_launch(const float* grad, const float* inputs, const int* offsets_host, float* grad_embeddings, uint32_t B, uint32_t D, uint32_t C,
                               uint32_t L, float S, uint32_t H, const float* dy_dx, float* grad_inputs, uint32_t gridtype, int align_corners, uint32_t interp,
                               hipStream_t st);
