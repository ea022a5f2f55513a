// Hash-grid encoder for input dimensions D = 2, 4, 5 (fp32): the drop-in's stand-alone op covers what the extension dispatches over at
// gridencoder/src/gridencoder.cu:386-399 (forward + dy_dx), :430-444 (backward).  The render path is D = 3 (pn_nerf_forward.hip, pn_net_tile.h) and does
// not come here.  gfx950 only.
//
// One thread per (sample, level), blockIdx.y = level — a launch sweeps one level's table at a time, which stays in the XCD L2s.  Index arithmetic and
// summation order are kernel_grid<float, D, C>'s (corner idx ascending, channels inside), `inputs * scale + offset` rounded once (nvcc's default
// contraction, as the D = 3 kernels do), so results equal the reference kernel's bit for bit on the contracting build.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/pienerf_hip.h"
#include "pn_common.h"

namespace {

struct NdLevels {
    uint32_t offset[PN_MAX_LEVELS], hs[PN_MAX_LEVELS], res[PN_MAX_LEVELS];
    float scale[PN_MAX_LEVELS];
    uint32_t L;
};

// get_grid_index<D, C> with ch = 0 (gridencoder.cu:66-84): strided while the stride fits the table, else (hash grid type) the coherent prime hash
template <uint32_t D>
__device__ __forceinline__ uint32_t index_nd(uint32_t gridtype, bool align, uint32_t hs, uint32_t res, const uint32_t (&p)[D]) {
    constexpr uint32_t primes[7] = {1u, 2654435761u, 805459861u, 3674653429u, 2097192037u, 1434869437u, 2165219737u};
    uint32_t stride = 1, index = 0;
#pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        if (stride <= hs) {
            index += p[d] * stride;
            stride *= align ? res : (res + 1);
        }
    }
    if (gridtype == 0 && stride > hs) {
        index = 0;
#pragma unroll
        for (uint32_t d = 0; d < D; d++) index ^= p[d] * primes[d];
    }
    return index % hs;
}

template <uint32_t D>
struct CellNd {
    float pos[D], deriv[D];
    uint32_t pg[D];
};

template <uint32_t D>
__device__ __forceinline__ bool locate_nd(const float* __restrict__ in, float scale, bool align, uint32_t interp, CellNd<D>& c) {
    bool inside = true;
#pragma unroll
    for (uint32_t d = 0; d < D; d++) inside = inside && !(in[d] < 0 || in[d] > 1);   // gridencoder.cu:113-118
    if (!inside) return false;
#pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        float p = fmaf(in[d], scale, align ? 0.0f : 0.5f);
        c.pg[d] = (uint32_t)floorf(p);
        p -= (float)c.pg[d];
        if (interp == 1) { c.deriv[d] = 6 * p * (1.0f - p); p = p * p * (3.0f - 2.0f * p); }
        else c.deriv[d] = 1.0f;
        c.pos[d] = p;
    }
    return true;
}

// outputs: [L, B, C] (rows == 0, the reference kernel's layout) or [B, L * C] (rows == 1, what grid.py:57 permutes to); dy_dx (may be null) [B, L, D, C]
template <uint32_t D, uint32_t C>
__global__ void __launch_bounds__(256) k_grid_nd(const float* __restrict__ inputs, const float* __restrict__ emb, NdLevels lv, uint32_t B, uint32_t gridtype,
                                                 int align, uint32_t interp, int rows, float* __restrict__ outputs, float* __restrict__ dy_dx) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const uint32_t level = blockIdx.y, L = lv.L, hs = lv.hs[level], res = lv.res[level];
    const float scale = lv.scale[level];
    const float* __restrict__ table = emb + (size_t)lv.offset[level] * C;
    float* out = outputs + (rows ? ((size_t)b * L + level) : ((size_t)level * B + b)) * C;
    float* dd = dy_dx ? dy_dx + ((size_t)b * L + level) * D * C : nullptr;
    CellNd<D> c;
    if (!locate_nd<D>(inputs + (size_t)b * D, scale, align != 0, interp, c)) {
#pragma unroll
        for (uint32_t ch = 0; ch < C; ch++) out[ch] = 0;
        if (dd) for (uint32_t i = 0; i < D * C; i++) dd[i] = 0;
        return;
    }
    float r[C];
#pragma unroll
    for (uint32_t ch = 0; ch < C; ch++) r[ch] = 0;
#pragma unroll
    for (uint32_t idx = 0; idx < (1u << D); idx++) {
        float w = 1;
        uint32_t pl[D];
#pragma unroll
        for (uint32_t d = 0; d < D; d++) {
            if ((idx & (1u << d)) == 0) { w *= 1 - c.pos[d]; pl[d] = c.pg[d]; }
            else { w *= c.pos[d]; pl[d] = c.pg[d] + 1; }
        }
        const uint32_t index = index_nd<D>(gridtype, align != 0, hs, res, pl) * C;
#pragma unroll
        for (uint32_t ch = 0; ch < C; ch++) r[ch] += w * table[index + ch];
    }
#pragma unroll
    for (uint32_t ch = 0; ch < C; ch++) out[ch] = r[ch];
    if (!dd) return;
#pragma unroll
    for (uint32_t gd = 0; gd < D; gd++) {   // gridencoder.cu:204-243
        float g[C];
#pragma unroll
        for (uint32_t ch = 0; ch < C; ch++) g[ch] = 0;
#pragma unroll
        for (uint32_t idx = 0; idx < (1u << (D - 1)); idx++) {
            float w = scale;
            uint32_t pl[D];
#pragma unroll
            for (uint32_t nd = 0; nd < D - 1; nd++) {
                const uint32_t d = (nd >= gd) ? (nd + 1) : nd;
                if ((idx & (1u << nd)) == 0) { w *= 1 - c.pos[d]; pl[d] = c.pg[d]; }
                else { w *= c.pos[d]; pl[d] = c.pg[d] + 1; }
            }
            pl[gd] = c.pg[gd];
            const uint32_t il = index_nd<D>(gridtype, align != 0, hs, res, pl) * C;
            pl[gd] = c.pg[gd] + 1;
            const uint32_t ir = index_nd<D>(gridtype, align != 0, hs, res, pl) * C;
#pragma unroll
            for (uint32_t ch = 0; ch < C; ch++) g[ch] += w * (table[ir + ch] - table[il + ch]) * c.deriv[gd];
        }
#pragma unroll
        for (uint32_t ch = 0; ch < C; ch++) dd[gd * C + ch] = g[ch];
    }
}

// kernel_grid_backward<float, D, C, N_C> (gridencoder.cu:248-340): grad [L, B, C]; hardware fp32 atomics into the level's table
template <uint32_t D, uint32_t C>
__global__ void __launch_bounds__(256) k_grid_nd_backward(const float* __restrict__ grad, const float* __restrict__ inputs, NdLevels lv, uint32_t B,
                                                          uint32_t gridtype, int align, uint32_t interp, float* __restrict__ grad_emb) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const uint32_t level = blockIdx.y, hs = lv.hs[level], res = lv.res[level];
    CellNd<D> c;
    if (!locate_nd<D>(inputs + (size_t)b * D, lv.scale[level], align != 0, interp, c)) return;
    float* __restrict__ gt = grad_emb + (size_t)lv.offset[level] * C;
    float g[C];
#pragma unroll
    for (uint32_t ch = 0; ch < C; ch++) g[ch] = grad[((size_t)level * B + b) * C + ch];
#pragma unroll
    for (uint32_t idx = 0; idx < (1u << D); idx++) {
        float w = 1;
        uint32_t pl[D];
#pragma unroll
        for (uint32_t d = 0; d < D; d++) {
            if ((idx & (1u << d)) == 0) { w *= 1 - c.pos[d]; pl[d] = c.pg[d]; }
            else { w *= c.pos[d]; pl[d] = c.pg[d] + 1; }
        }
        const uint32_t index = index_nd<D>(gridtype, align != 0, hs, res, pl) * C;
#pragma unroll
        for (uint32_t ch = 0; ch < C; ch++) unsafeAtomicAdd(gt + index + ch, w * g[ch]);
    }
}

// kernel_input_backward (gridencoder.cu:343-369)
__global__ void __launch_bounds__(256) k_grid_nd_input_backward(const float* __restrict__ grad, const float* __restrict__ dy_dx, float* __restrict__ grad_inputs,
                                                                uint32_t B, uint32_t L, uint32_t D, uint32_t C) {
    const uint32_t t = threadIdx.x + blockIdx.x * blockDim.x;
    if (t >= B * D) return;
    const uint32_t b = t / D, d = t - b * D;
    float r = 0;
    for (uint32_t l = 0; l < L; l++)
        for (uint32_t ch = 0; ch < C; ch++) r += grad[((size_t)l * B + b) * C + ch] * dy_dx[(((size_t)b * L + l) * D + d) * C + ch];
    grad_inputs[t] = r;
}

// kernel_grad_tv<float, D, C> (gridencoder.cu:506-611)
template <uint32_t D, uint32_t C>
__global__ void __launch_bounds__(256) k_grid_nd_grad_tv(const float* __restrict__ inputs, const float* __restrict__ emb, float* __restrict__ grad, NdLevels lv,
                                                         float weight, uint32_t B, uint32_t gridtype, int align) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const uint32_t level = blockIdx.y, hs = lv.hs[level], res = lv.res[level];
    const float scale = lv.scale[level];
    const float* __restrict__ in = inputs + (size_t)b * D;
    bool inside = true;
#pragma unroll
    for (uint32_t d = 0; d < D; d++) inside = inside && !(in[d] < 0 || in[d] > 1);
    if (!inside) return;
    const float* __restrict__ table = emb + (size_t)lv.offset[level] * C;
    float* __restrict__ gt = grad + (size_t)lv.offset[level] * C;
    uint32_t pg[D];
#pragma unroll
    for (uint32_t d = 0; d < D; d++) pg[d] = (uint32_t)floorf(fmaf(in[d], scale, align ? 0.0f : 0.5f));
    float results[C], idelta[C];
#pragma unroll
    for (uint32_t ch = 0; ch < C; ch++) { results[ch] = 0; idelta[ch] = 0; }
    const uint32_t index = index_nd<D>(gridtype, align != 0, hs, res, pg) * C;
    const float w = weight / (2 * D);
#pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        const uint32_t cur = pg[d];
        if (cur < res) {
            pg[d] = cur + 1;
            const uint32_t ir = index_nd<D>(gridtype, align != 0, hs, res, pg) * C;
#pragma unroll
            for (uint32_t ch = 0; ch < C; ch++) { const float gv = table[index + ch] - table[ir + ch]; results[ch] += gv; idelta[ch] += gv * gv; }
        }
        if (cur > 0) {
            pg[d] = cur - 1;
            const uint32_t il = index_nd<D>(gridtype, align != 0, hs, res, pg) * C;
#pragma unroll
            for (uint32_t ch = 0; ch < C; ch++) { const float gv = table[index + ch] - table[il + ch]; results[ch] += gv; idelta[ch] += gv * gv; }
        }
        pg[d] = cur;
    }
#pragma unroll
    for (uint32_t ch = 0; ch < C; ch++) unsafeAtomicAdd(gt + index + ch, w * results[ch] * (1.0f / sqrtf(idelta[ch] + 1e-9f)));
}

int fill_levels(NdLevels* lv, const int* offsets_host, uint32_t L, float S, uint32_t H) {
    if (L == 0 || L > PN_MAX_LEVELS) return PN_ERR_ARG;
    lv->L = L;
    for (uint32_t l = 0; l < L; l++) {
        const float scale = exp2f(l * S) * H - 1.0f;          // gridencoder.cu:133-134
        lv->scale[l] = scale;
        lv->res[l] = (uint32_t)ceilf(scale) + 1;
        lv->offset[l] = (uint32_t)offsets_host[l];
        lv->hs[l] = (uint32_t)(offsets_host[l + 1] - offsets_host[l]);
        if (lv->hs[l] == 0) return PN_ERR_ARG;
    }
    return PN_OK;
}

}  // namespace

#define PN_ND_DISPATCH(KERNEL, ...)                                                                   \
    do {                                                                                              \
        switch (D * 16 + C) {                                                                         \
            case 2 * 16 + 1: KERNEL<2, 1><<<grid, 256, 0, st>>>(__VA_ARGS__); break;                  \
            case 2 * 16 + 2: KERNEL<2, 2><<<grid, 256, 0, st>>>(__VA_ARGS__); break;                  \
            case 2 * 16 + 4: KERNEL<2, 4><<<grid, 256, 0, st>>>(__VA_ARGS__); break;                  \
            case 2 * 16 + 8: KERNEL<2, 8><<<grid, 256, 0, st>>>(__VA_ARGS__); break;                  \
            case 4 * 16 + 1: KERNEL<4, 1><<<grid, 256, 0, st>>>(__VA_ARGS__); break;                  \
            case 4 * 16 + 2: KERNEL<4, 2><<<grid, 256, 0, st>>>(__VA_ARGS__); break;                  \
            case 4 * 16 + 4: KERNEL<4, 4><<<grid, 256, 0, st>>>(__VA_ARGS__); break;                  \
            case 4 * 16 + 8: KERNEL<4, 8><<<grid, 256, 0, st>>>(__VA_ARGS__); break;                  \
            case 5 * 16 + 1: KERNEL<5, 1><<<grid, 256, 0, st>>>(__VA_ARGS__); break;                  \
            case 5 * 16 + 2: KERNEL<5, 2><<<grid, 256, 0, st>>>(__VA_ARGS__); break;                  \
            case 5 * 16 + 4: KERNEL<5, 4><<<grid, 256, 0, st>>>(__VA_ARGS__); break;                  \
            case 5 * 16 + 8: KERNEL<5, 8><<<grid, 256, 0, st>>>(__VA_ARGS__); break;                  \
            default: return PN_ERR_ARG;                                                               \
        }                                                                                             \
    } while (0)

// called by pn_grid_encode_forward / pn_grid_encode_backward (their D != 3 branch); same argument meaning
int pn_grid_nd_forward_launch(const float* inputs, const float* embeddings, const int* offsets_host, float* outputs, uint32_t B, uint32_t D, uint32_t C,
                              uint32_t L, float S, uint32_t H, float* dy_dx, uint32_t gridtype, int align_corners, uint32_t interp, int out_bl_major,
                              hipStream_t st) {
    PN_REQUIRE(inputs && embeddings && offsets_host && outputs);
    PN_REQUIRE((D == 2 || D == 4 || D == 5) && gridtype <= 1 && interp <= 1);   // gridencoder.cu:393-398
    NdLevels lv;
    if (fill_levels(&lv, offsets_host, L, S, H)) { PN_REQUIRE(L >= 1 && L <= PN_MAX_LEVELS); }
    const dim3 grid(pn_div_up(B, 256), L, 1);
    PN_ND_DISPATCH(k_grid_nd, inputs, embeddings, lv, B, gridtype, align_corners, interp, out_bl_major, outputs, dy_dx);
    PN_LAUNCH_CHECK();
    return PN_OK;
}

int pn_grid_nd_backward_launch(const float* grad, const float* inputs, const int* offsets_host, float* grad_embeddings, uint32_t B, uint32_t D, uint32_t C,
                               uint32_t L, float S, uint32_t H, const float* dy_dx, float* grad_inputs, uint32_t gridtype, int align_corners, uint32_t interp,
                               hipStream_t st) {
    PN_REQUIRE(grad && inputs && offsets_host && grad_embeddings);
    PN_REQUIRE((D == 2 || D == 4 || D == 5) && gridtype <= 1 && interp <= 1);   // gridencoder.cu:437-442
    PN_REQUIRE((dy_dx == nullptr) == (grad_inputs == nullptr));
    NdLevels lv;
    if (fill_levels(&lv, offsets_host, L, S, H)) { PN_REQUIRE(L >= 1 && L <= PN_MAX_LEVELS); }
    const dim3 grid(pn_div_up(B, 256), L, 1);
    PN_ND_DISPATCH(k_grid_nd_backward, grad, inputs, lv, B, gridtype, align_corners, interp, grad_embeddings);
    if (dy_dx) k_grid_nd_input_backward<<<pn_div_up((uint64_t)B * D, 256), 256, 0, st>>>(grad, dy_dx, grad_inputs, B, L, D, C);
    PN_LAUNCH_CHECK();
    return PN_OK;
}

int pn_grid_nd_grad_tv_launch(const float* inputs, const float* embeddings, float* grad, const int* offsets_host, float weight, uint32_t B, uint32_t D, uint32_t C,
                              uint32_t L, float S, uint32_t H, uint32_t gridtype, int align_corners, hipStream_t st) {
    PN_REQUIRE(inputs && embeddings && grad && offsets_host);
    PN_REQUIRE((D == 2 || D == 4 || D == 5) && gridtype <= 1);   // gridencoder.cu:629-634
    NdLevels lv;
    if (fill_levels(&lv, offsets_host, L, S, H)) { PN_REQUIRE(L >= 1 && L <= PN_MAX_LEVELS); }
    const dim3 grid(pn_div_up(B, 256), L, 1);
    PN_ND_DISPATCH(k_grid_nd_grad_tv, inputs, embeddings, grad, lv, weight, B, gridtype, align_corners);
    PN_LAUNCH_CHECK();
    return PN_OK;
}
