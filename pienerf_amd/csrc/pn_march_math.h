// Shared pieces of the device-side ray march with the inverse-GMLS ("quadratic bending") warp, for gfx950: scalar helpers,
// the literal flat-index 3x3 routines and the kernel parameter block (included by pn_march_tables.h; the march itself is pn_march_window.h).
//
// Semantics follow kernel_march_rays_quadratic_bending and its helpers
// (/root/reference raymarching/src/raymarching.cu:930-1434) including the output-changing quirks listed in
// SURVEY.md §8a R7q.  This translation unit is compiled with -ffp-contract=off: every float operation rounds once, in
// the order written, so the integer decisions (search cell, accepted-IP count, mip level, voxel index) match the CPU
// oracle bit for bit.  Mixed-precision subexpressions of the reference are kept (double `0.5 * ... * H`, `bbmax - 1e-6`,
// the double Newton residual).
#pragma once
#include "pn_common.h"

namespace pnm {

__device__ __forceinline__ float clampf(float x, float lo, float hi) { return fminf(hi, fmaxf(lo, x)); }
__device__ __forceinline__ float signf(float x) { return copysignf(1.0f, x); }

__device__ __forceinline__ int mip_from_pos(float x, float y, float z, float max_cascade) {
    const float mx = fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z)));
    int e;
    frexpf(mx, &e);
    return (int)fminf(max_cascade - 1, fmaxf(0.0f, (float)e));
}
__device__ __forceinline__ int mip_from_dt(float dt, float H, float max_cascade) {
    const float mx = (dt * H) * 0.5f;  // == (float)((double)(dt * H) * 0.5): halving is exact
    int e;
    frexpf(mx, &e);
    return (int)fminf(max_cascade - 1, fmaxf(0.0f, (float)e));
}
__device__ __forceinline__ uint32_t expand_bits(uint32_t v) {
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
__device__ __forceinline__ uint32_t morton3D(uint32_t x, uint32_t y, uint32_t z) {
    return expand_bits(x) | (expand_bits(y) << 1) | (expand_bits(z) << 2);
}

// 26 neighbour offsets in the reference's visiting order (raymarching.cu:1006-1016), packed 2 bits per component (+1).
__device__ __constant__ const signed char NBR26[26][3] = {
    {-1, 0, 0}, {0, -1, 0}, {0, 0, -1}, {1, 0, 0},  {0, 1, 0},  {0, 0, 1},  {-1, -1, 0}, {-1, 0, -1}, {0, -1, -1},
    {1, 1, 0},  {1, 0, 1},  {0, 1, 1},  {-1, 1, 0}, {-1, 0, 1}, {0, -1, 1}, {1, -1, 0},  {1, 0, -1},  {0, 1, -1},
    {-1, -1, 1}, {-1, 1, -1}, {1, -1, -1}, {1, 1, -1}, {1, -1, 1}, {-1, 1, 1}, {-1, -1, -1}, {1, 1, 1}};

// raymarching.cu:940-984 literal flat-index helpers
__device__ __forceinline__ void dot31(const float* __restrict__ T, const float* V, float* M) {
#pragma unroll
    for (int m = 0; m < 9; m++) M[m] = T[m] * V[0] + T[9 + m] * V[1] + T[18 + m] * V[2];
}
__device__ __forceinline__ void mul31(const float* M, const float* V, float* R) {
    R[0] = M[0] * V[0] + M[3] * V[1] + M[6] * V[2];
    R[1] = M[1] * V[0] + M[4] * V[1] + M[7] * V[2];
    R[2] = M[2] * V[0] + M[5] * V[1] + M[8] * V[2];
}
__device__ __forceinline__ void inv3x3(const float* A, float* Ai) {
    const float det = A[0] * (A[4] * A[8] - A[5] * A[7]) - A[1] * (A[3] * A[8] - A[5] * A[6]) + A[2] * (A[3] * A[7] - A[4] * A[6]);
    if (det == 0) return;  // Ai stays 0 (the reference never acts on the failure code, :1285-1287)
    const float id = 1.0f / det;
    Ai[0] = id * (A[4] * A[8] - A[5] * A[7]);
    Ai[1] = id * (A[2] * A[7] - A[1] * A[8]);
    Ai[2] = id * (A[1] * A[5] - A[2] * A[4]);
    Ai[3] = id * (A[5] * A[6] - A[3] * A[8]);
    Ai[4] = id * (A[0] * A[8] - A[2] * A[6]);
    Ai[5] = id * (A[2] * A[3] - A[0] * A[5]);
    Ai[6] = id * (A[3] * A[7] - A[4] * A[6]);
    Ai[7] = id * (A[1] * A[6] - A[0] * A[7]);
    Ai[8] = id * (A[0] * A[4] - A[1] * A[3]);
}

struct MarchParams {
    const int *pig_cnt, *pig_bgn, *pig_idx;
    int n_vtx, n_grid;
    const float *p_ori, *p_def, *F_IP, *dF_IP;
    int max_iter_num;
    const float *bbmin, *bbmax;
    float hgs;
    const int* resolution;
    int num_seek_IP;
    float IP_dx;
    int cut;
    const float* cut_bounds;
    const float *rays_t, *rays_o, *rays_d;
    float bound, dt_gamma;
    uint32_t max_steps, C, H;
    const uint8_t* grid;
    const float* fars;
    int* err_flag;
    unsigned long long* stats;  // optional [4]: marching iterations, candidates scanned, IP warps, samples emitted (bench instrumentation)
};

}  // namespace pnm
