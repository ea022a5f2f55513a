// Training-side encoder kernels for gfx950 (off the simulate-and-render hot path): the hash grid's dy_dx,
// backward and total-variation gradient, the SH encoder's dy_dx and backward.
//
// Reference: gridencoder/src/gridencoder.cu:199-243 (dy_dx branch of kernel_grid), :248-340 (kernel_grid_backward), :343-369
// (kernel_input_backward), :506-611 (kernel_grad_tv); shencoder/src/shencoder.cu:125-355 (dy_dx branch of kernel_sh), :358-383
// (kernel_sh_backward) — paths relative to /root/reference.
//
// Mapping: one lane per (sample, level) with blockIdx.y = level, like the forward kernel, so that a level's table slice (and its
// gradient slice) stays in the XCD L2s while that level is processed; scatter-adds are hardware fp32 atomics
// (global_atomic_add_f32 at L2, no CAS loop).  Like the reference's atomicAdd the summation order is not fixed, so gradients are
// compared with the oracle to a tolerance, not bit for bit.
#include "pn_encoders.h"
#include "pn_sh_bands.h"

namespace {

struct Cell { float pos[3]; float deriv[3]; uint32_t pg[3]; };

__device__ __forceinline__ bool locate(const float* __restrict__ in3, float scale, int align_corners, uint32_t interp, Cell& c) {
    const float in0 = in3[0], in1 = in3[1], in2 = in3[2];
    if (in0 < 0 || in0 > 1 || in1 < 0 || in1 > 1 || in2 < 0 || in2 > 1) return false;
    const float v[3] = {in0, in1, in2};
#pragma unroll
    for (int d = 0; d < 3; d++) {
        float p = fmaf(v[d], scale, align_corners ? 0.0f : 0.5f);  // the forward kernel's explicit single rounding
        c.pg[d] = (uint32_t)floorf(p);
        p -= (float)c.pg[d];
        if (interp == 1) { c.deriv[d] = 6 * p * (1 - p); p = p * p * (3.0f - 2.0f * p); }
        else c.deriv[d] = 1.0f;
        c.pos[d] = p;
    }
    return true;
}

// dy_dx [B, L, 3, C] (gridencoder.cu:199-243)
template <uint32_t C>
__global__ void __launch_bounds__(256) k_grid_dy_dx(const float* __restrict__ inputs, const float* __restrict__ emb, PnGridLevels lv, uint32_t B,
                                                    int align_corners, uint32_t interp, float* __restrict__ dy_dx) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const uint32_t level = blockIdx.y;
    float* out = dy_dx + ((size_t)b * lv.L + level) * 3 * C;
    Cell c;
    const float scale = lv.scale[level];
    if (!locate(inputs + (size_t)b * 3, scale, align_corners, interp, c)) {
#pragma unroll
        for (uint32_t i = 0; i < 3 * C; i++) out[i] = 0;  // gridencoder.cu:115-125
        return;
    }
    const float* __restrict__ table = emb + (size_t)lv.offset[level] * C;
    const LevelIdx LI = level_idx(lv, level, align_corners);
#pragma unroll
    for (uint32_t gd = 0; gd < 3; gd++) {
        float rg[C];
#pragma unroll
        for (uint32_t ch = 0; ch < C; ch++) rg[ch] = 0;
#pragma unroll
        for (uint32_t idx = 0; idx < 4; idx++) {
            float w = scale;
            uint32_t pl[3];
#pragma unroll
            for (uint32_t nd = 0; nd < 2; nd++) {
                const uint32_t d = (nd >= gd) ? (nd + 1) : nd;
                if ((idx & (1u << nd)) == 0) { w *= 1 - c.pos[d]; pl[d] = c.pg[d]; }
                else { w *= c.pos[d]; pl[d] = c.pg[d] + 1; }
            }
            pl[gd] = c.pg[gd];
            const uint32_t il = grid_index3(LI, pl[0], pl[1], pl[2]) * C;
            pl[gd] = c.pg[gd] + 1;
            const uint32_t ir = grid_index3(LI, pl[0], pl[1], pl[2]) * C;
#pragma unroll
            for (uint32_t ch = 0; ch < C; ch++) rg[ch] += w * (table[ir + ch] - table[il + ch]) * c.deriv[gd];
        }
#pragma unroll
        for (uint32_t ch = 0; ch < C; ch++) out[gd * C + ch] = rg[ch];
    }
}

// grad_embeddings += w * grad (gridencoder.cu:248-340); grad [L, B, C].
// Device-scope fp32 atomics execute at the memory side on gfx950 (the per-XCD L2s are not coherent), ~10 ns each when they pile up
// on one address, and that is exactly what the coarse levels do: samples arrive in ray order (pn_march_rays_train keeps them so), so
// neighbouring lanes sit in the same cell of a 16..100-cell-wide level and hit the same 8 corners.  Each wave therefore folds runs of
// equal target rows before touching memory: a lane starts a run when its row differs from the previous lane's, run ids come from a
// ballot + prefix popcount, a 6-step shuffle tree adds a lane's partial to the lane `off` below it while both carry the same run id
// (ids are monotonic, so equal ids = one contiguous run), and only run heads issue the atomic.  Fine hashed levels (no sharing)
// degenerate to one atomic per lane, as before; coarse levels drop to one per cell crossing.
template <uint32_t C>
__global__ void __launch_bounds__(256) k_grid_backward(const float* __restrict__ grad, const float* __restrict__ inputs, PnGridLevels lv, uint32_t B,
                                                       int align_corners, uint32_t interp, float* __restrict__ grad_emb) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t level = blockIdx.y;
    const uint32_t lane = threadIdx.x & 63;
    Cell c;
    const bool valid = b < B && locate(inputs + (size_t)b * 3, lv.scale[level], align_corners, interp, c);
    float* __restrict__ gt = grad_emb + (size_t)lv.offset[level] * C;
    const LevelIdx LI = level_idx(lv, level, align_corners);
    float g[C];
#pragma unroll
    for (uint32_t ch = 0; ch < C; ch++) g[ch] = valid ? grad[((size_t)level * B + b) * C + ch] : 0.0f;
#pragma unroll
    for (uint32_t idx = 0; idx < 8; idx++) {
        float w = 1;
        uint32_t pl[3];
#pragma unroll
        for (int d = 0; d < 3; d++) {
            if ((idx & (1u << d)) == 0) { w *= 1 - c.pos[d]; pl[d] = c.pg[d]; }
            else { w *= c.pos[d]; pl[d] = c.pg[d] + 1; }
        }
        const uint32_t row = valid ? grid_index3(LI, pl[0], pl[1], pl[2]) : 0xFFFFFFFFu;
        float v[C];
#pragma unroll
        for (uint32_t ch = 0; ch < C; ch++) v[ch] = valid ? w * g[ch] : 0.0f;
        const uint32_t prev = __shfl_up(row, 1, 64);
        const bool head = lane == 0 || prev != row;
        const unsigned long long heads = __ballot(head);
        const uint32_t run = __popcll(heads & ((2ull << lane) - 1ull));  // number of heads at or below this lane: monotonic run id
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t run2 = __shfl_down(run, off, 64);
            const bool take = lane + off < 64 && run2 == run;
#pragma unroll
            for (uint32_t ch = 0; ch < C; ch++) {
                const float o = __shfl_down(v[ch], off, 64);
                if (take) v[ch] += o;
            }
        }
        if (head && row != 0xFFFFFFFFu) {
#pragma unroll
            for (uint32_t ch = 0; ch < C; ch++) unsafeAtomicAdd(gt + (size_t)row * C + ch, v[ch]);
        }
    }
}

// kernel_grid_backward<at::Half> (gridencoder.cu:248-341; the table is half under autocast, grid.py:43-44): grad [L,B,C] half, every contribution
// rounded to half — (__half)(w * grad), :327 — and accumulated into the half table two channels at a time (:324-331: atomicAdd on a __half2; here
// global_atomic_pk_add_f16).  The order in which contributions meet is the atomics' in the reference, so its result is only defined up to the rounding of
// a sum of halves; the runs of equal rows inside a wave are folded first as above, with half additions (each partial sum rounded to half, as an atomic
// would leave it), and the run heads issue the atomics.  C even (the reference uses this path for N_C % 2 == 0; with one channel autocast keeps fp32).
typedef _Float16 pn_gh2 __attribute__((ext_vector_type(2)));
template <uint32_t C>
__global__ void __launch_bounds__(256) k_grid_backward_h(const _Float16* __restrict__ grad, const float* __restrict__ inputs, PnGridLevels lv, uint32_t B,
                                                         int align_corners, uint32_t interp, _Float16* __restrict__ grad_emb) {
    static_assert(C % 2 == 0, "channel pairs");
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t level = blockIdx.y;
    const uint32_t lane = threadIdx.x & 63;
    Cell c;
    const bool valid = b < B && locate(inputs + (size_t)b * 3, lv.scale[level], align_corners, interp, c);
    _Float16* __restrict__ gt = grad_emb + (size_t)lv.offset[level] * C;
    const LevelIdx LI = level_idx(lv, level, align_corners);
    float g[C];
#pragma unroll
    for (uint32_t ch = 0; ch < C; ch++) g[ch] = valid ? (float)grad[((size_t)level * B + b) * C + ch] : 0.0f;
#pragma unroll
    for (uint32_t idx = 0; idx < 8; idx++) {
        float w = 1;
        uint32_t pl[3];
#pragma unroll
        for (int d = 0; d < 3; d++) {
            if ((idx & (1u << d)) == 0) { w *= 1 - c.pos[d]; pl[d] = c.pg[d]; }
            else { w *= c.pos[d]; pl[d] = c.pg[d] + 1; }
        }
        const uint32_t row = valid ? grid_index3(LI, pl[0], pl[1], pl[2]) : 0xFFFFFFFFu;
        _Float16 v[C];
#pragma unroll
        for (uint32_t ch = 0; ch < C; ch++) v[ch] = valid ? (_Float16)(w * g[ch]) : (_Float16)0.0f;
        const uint32_t prev = __shfl_up(row, 1, 64);
        const bool head = lane == 0 || prev != row;
        const unsigned long long heads = __ballot(head);
        const uint32_t run = __popcll(heads & ((2ull << lane) - 1ull));
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t run2 = __shfl_down(run, off, 64);
            const bool take = lane + off < 64 && run2 == run;
#pragma unroll
            for (uint32_t ch = 0; ch < C; ch += 2) {
                const pn_gh2 mine = {v[ch], v[ch + 1]};
                const pn_gh2 o = __builtin_bit_cast(pn_gh2, __shfl_down(__builtin_bit_cast(int, mine), off, 64));
                if (take) { v[ch] = v[ch] + o.x; v[ch + 1] = v[ch + 1] + o.y; }
            }
        }
        if (head && row != 0xFFFFFFFFu) {
#pragma unroll
            for (uint32_t ch = 0; ch < C; ch += 2) {
                const pn_gh2 pk = {v[ch], v[ch + 1]};
                __builtin_amdgcn_global_atomic_fadd_v2f16(reinterpret_cast<pn_gh2*>(gt + (size_t)row * C + ch), pk);
            }
        }
    }
}

// grad_inputs[b, d] = sum_{l,c} grad[l,b,c] * dy_dx[b,l,d,c] (gridencoder.cu:343-369)
__global__ void __launch_bounds__(256) k_grid_input_backward(const float* __restrict__ grad, const float* __restrict__ dy_dx, float* __restrict__ grad_inputs,
                                                             uint32_t B, uint32_t L, uint32_t C) {
    const uint32_t t = threadIdx.x + blockIdx.x * blockDim.x;
    if (t >= B * 3) return;
    const uint32_t b = t / 3, d = t - b * 3;
    float result = 0;
    for (uint32_t l = 0; l < L; l++)
        for (uint32_t ch = 0; ch < C; ch++) result += grad[((size_t)l * B + b) * C + ch] * dy_dx[(((size_t)b * L + l) * 3 + d) * C + ch];
    grad_inputs[t] = result;
}

// kernel_grad_tv (gridencoder.cu:506-611)
template <uint32_t C>
__global__ void __launch_bounds__(256) k_grad_tv(const float* __restrict__ inputs, const float* __restrict__ emb, float* __restrict__ grad, PnGridLevels lv,
                                                 float weight, uint32_t B, int align_corners) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const uint32_t level = blockIdx.y;
    Cell c;
    if (!locate(inputs + (size_t)b * 3, lv.scale[level], align_corners, 0, c)) return;
    const float* __restrict__ table = emb + (size_t)lv.offset[level] * C;
    float* __restrict__ gt = grad + (size_t)lv.offset[level] * C;
    const LevelIdx LI = level_idx(lv, level, align_corners);
    const uint32_t resolution = lv.resolution[level];
    float results[C], idelta[C], here[C];
    uint32_t pg[3] = {c.pg[0], c.pg[1], c.pg[2]};
    const uint32_t index = grid_index3(LI, pg[0], pg[1], pg[2]) * C;
#pragma unroll
    for (uint32_t ch = 0; ch < C; ch++) { results[ch] = 0; idelta[ch] = 0; here[ch] = table[index + ch]; }
    const float w = weight / (2 * 3);
#pragma unroll
    for (int d = 0; d < 3; d++) {
        const uint32_t cur = pg[d];
        if (cur < resolution) {
            pg[d] = cur + 1;
            const uint32_t ir = grid_index3(LI, pg[0], pg[1], pg[2]) * C;
#pragma unroll
            for (uint32_t ch = 0; ch < C; ch++) { const float gv = here[ch] - table[ir + ch]; results[ch] += gv; idelta[ch] += gv * gv; }
        }
        if (cur > 0) {
            pg[d] = cur - 1;
            const uint32_t il = grid_index3(LI, pg[0], pg[1], pg[2]) * C;
#pragma unroll
            for (uint32_t ch = 0; ch < C; ch++) { const float gv = here[ch] - table[il + ch]; results[ch] += gv; idelta[ch] += gv * gv; }
        }
        pg[d] = cur;
    }
#pragma unroll
    for (uint32_t ch = 0; ch < C; ch++) unsafeAtomicAdd(gt + index + ch, w * results[ch] * (1.0f / sqrtf(idelta[ch] + 1e-9f)));
}

// d/dx, d/dy, d/dz of the 16 polynomials of sh16 (pn_nerf_forward.hip); shencoder.cu:125-355 tabulates the same derivatives
__device__ __forceinline__ void sh16_grad(float x, float y, float z, float* gx, float* gy, float* gz) {
    const float xy = x * y, xz = x * z, yz = y * z, x2 = x * x, y2 = y * y, z2 = z * z;
#pragma unroll
    for (int i = 0; i < 16; i++) gx[i] = gy[i] = gz[i] = 0.0f;
    gy[1] = -SH_C1; gz[2] = SH_C1; gx[3] = -SH_C1;
    gx[4] = SH_C2A * y; gy[4] = SH_C2A * x;
    gy[5] = -SH_C2A * z; gz[5] = -SH_C2A * y;
    gz[6] = 2.0f * SH_C2B * z;
    gx[7] = -SH_C2A * z; gz[7] = -SH_C2A * x;
    gx[8] = 2.0f * SH_C2D * x; gy[8] = -2.0f * SH_C2D * y;
    gx[9] = -6.0f * SH_C3A * xy; gy[9] = SH_C3A * (-3.0f * x2 + 3.0f * y2);
    gx[10] = SH_C3B * yz; gy[10] = SH_C3B * xz; gz[10] = SH_C3B * xy;
    gy[11] = SH_C3C * (1.0f - 5.0f * z2); gz[11] = -10.0f * SH_C3C * yz;
    gz[12] = SH_C3D * (15.0f * z2 - 3.0f);
    gx[13] = SH_C3C * (1.0f - 5.0f * z2); gz[13] = -10.0f * SH_C3C * xz;
    gx[14] = 2.0f * SH_C3E * xz; gy[14] = -2.0f * SH_C3E * yz; gz[14] = SH_C3E * (x2 - y2);
    gx[15] = SH_C3A * (-3.0f * x2 + 3.0f * y2); gy[15] = 6.0f * SH_C3A * xy;
}

__global__ void __launch_bounds__(256) k_sh_dy_dx(const float* __restrict__ inputs, float* __restrict__ dy_dx, uint32_t B, uint32_t C) {
    const uint32_t b = threadIdx.x + blockIdx.x * blockDim.x;
    if (b >= B) return;
    float g[3][16];
    const float x = inputs[(size_t)b * 3], y = inputs[(size_t)b * 3 + 1], z = inputs[(size_t)b * 3 + 2];
    sh16_grad(x, y, z, g[0], g[1], g[2]);
    const uint32_t C2 = C * C;
    for (uint32_t d = 0; d < 3; d++)
        for (uint32_t i = 0; i < (C2 < 16u ? C2 : 16u); i++) dy_dx[((size_t)b * 3 + d) * C2 + i] = g[d][i];
    if (C > 4) {  // degree 5-8 (shencoder.cu:125-355): bands 4.. by recurrence, straight into the three rows
        float* row = dy_dx + (size_t)b * 3 * C2;
        pnsh::high_bands(x, y, z, (int)C, nullptr, row, row + C2, row + 2 * C2);
    }
}

__global__ void __launch_bounds__(256) k_sh_backward(const float* __restrict__ grad, uint32_t B, uint32_t C, const float* __restrict__ dy_dx,
                                                     float* __restrict__ grad_inputs) {
    const uint32_t t = threadIdx.x + blockIdx.x * blockDim.x;
    const uint32_t b = t / 3;
    if (b >= B) return;
    const uint32_t d = t - b * 3, C2 = C * C;
    float acc = grad_inputs[t];  // `+=` like the reference (shencoder.cu:378); the wrapper passes zeros
    for (uint32_t ch = 0; ch < C2; ch++) acc += grad[(size_t)b * C2 + ch] * dy_dx[((size_t)b * 3 + d) * C2 + ch];
    grad_inputs[t] = acc;
}

}  // namespace

int pn_grid_dy_dx_launch(const float* inputs, const float* embeddings, const PnGridLevels& lv, uint32_t B, uint32_t C, int align_corners, uint32_t interp,
                         float* dy_dx, hipStream_t st) {
    dim3 grid(pn_div_up(B, 256), lv.L, 1);
    switch (C) {
        case 1: k_grid_dy_dx<1><<<grid, 256, 0, st>>>(inputs, embeddings, lv, B, align_corners, interp, dy_dx); break;
        case 2: k_grid_dy_dx<2><<<grid, 256, 0, st>>>(inputs, embeddings, lv, B, align_corners, interp, dy_dx); break;
        case 4: k_grid_dy_dx<4><<<grid, 256, 0, st>>>(inputs, embeddings, lv, B, align_corners, interp, dy_dx); break;
        default: k_grid_dy_dx<8><<<grid, 256, 0, st>>>(inputs, embeddings, lv, B, align_corners, interp, dy_dx); break;
    }
    PN_LAUNCH_CHECK();
    return PN_OK;
}

int pn_sh_dy_dx_launch(const float* inputs, float* dy_dx, uint32_t B, uint32_t C, hipStream_t st) {
    k_sh_dy_dx<<<pn_div_up(B, 256), 256, 0, st>>>(inputs, dy_dx, B, C);
    PN_LAUNCH_CHECK();
    return PN_OK;
}

extern "C" int pn_grid_encode_backward(const float* grad, const float* inputs, const float* embeddings, const int* offsets_host, float* grad_embeddings,
                                       uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, const float* dy_dx, float* grad_inputs,
                                       uint32_t gridtype, int align_corners, uint32_t interp, void* stream) {
    (void)embeddings;  // the fp32 path never reads the table in backward (the reference passes it for its dtype only)
    if (B == 0) return PN_OK;
    PN_REQUIRE(grad && inputs && offsets_host && grad_embeddings);
    if (D != 3)  // gridencoder.cu:437-442: D = 2, 4, 5
        return pn_grid_nd_backward_launch(grad, inputs, offsets_host, grad_embeddings, B, D, C, L, S, H, dy_dx, grad_inputs, gridtype, align_corners, interp,
                                          (hipStream_t)stream);
    PN_REQUIRE(D == 3 && (C == 1 || C == 2 || C == 4 || C == 8) && gridtype <= 1 && interp <= 1);
    PN_REQUIRE((dy_dx == nullptr) == (grad_inputs == nullptr));
    PnGridLevels lv;
    if (pn_fill_grid_levels(&lv, offsets_host, L, C, S, H, gridtype, align_corners)) { PN_REQUIRE(L >= 1 && L <= PN_MAX_LEVELS); }
    dim3 grid(pn_div_up(B, 256), L, 1);
    hipStream_t st = (hipStream_t)stream;
    switch (C) {
        case 1: k_grid_backward<1><<<grid, 256, 0, st>>>(grad, inputs, lv, B, align_corners, interp, grad_embeddings); break;
        case 2: k_grid_backward<2><<<grid, 256, 0, st>>>(grad, inputs, lv, B, align_corners, interp, grad_embeddings); break;
        case 4: k_grid_backward<4><<<grid, 256, 0, st>>>(grad, inputs, lv, B, align_corners, interp, grad_embeddings); break;
        default: k_grid_backward<8><<<grid, 256, 0, st>>>(grad, inputs, lv, B, align_corners, interp, grad_embeddings); break;
    }
    if (dy_dx) k_grid_input_backward<<<pn_div_up((uint64_t)B * 3, 256), 256, 0, st>>>(grad, dy_dx, grad_inputs, B, L, C);
    PN_LAUNCH_CHECK();
    return PN_OK;
}

extern "C" int pn_grid_encode_backward_half(const uint16_t* grad, const float* inputs, const int* offsets_host, uint16_t* grad_embeddings, uint32_t B,
                                            uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, uint32_t gridtype, int align_corners, uint32_t interp,
                                            void* stream) {
    if (B == 0) return PN_OK;
    PN_REQUIRE(grad && inputs && offsets_host && grad_embeddings);
    PN_REQUIRE(D == 3 && (C == 2 || C == 4 || C == 8) && gridtype <= 1 && interp <= 1);
    PnGridLevels lv;
    if (pn_fill_grid_levels(&lv, offsets_host, L, C, S, H, gridtype, align_corners)) { PN_REQUIRE(L >= 1 && L <= PN_MAX_LEVELS); }
    dim3 grid(pn_div_up(B, 256), L, 1);
    hipStream_t st = (hipStream_t)stream;
    const _Float16* g = reinterpret_cast<const _Float16*>(grad);
    _Float16* ge = reinterpret_cast<_Float16*>(grad_embeddings);
    switch (C) {
        case 2: k_grid_backward_h<2><<<grid, 256, 0, st>>>(g, inputs, lv, B, align_corners, interp, ge); break;
        case 4: k_grid_backward_h<4><<<grid, 256, 0, st>>>(g, inputs, lv, B, align_corners, interp, ge); break;
        default: k_grid_backward_h<8><<<grid, 256, 0, st>>>(g, inputs, lv, B, align_corners, interp, ge); break;
    }
    PN_LAUNCH_CHECK();
    return PN_OK;
}

extern "C" int pn_grad_total_variation(const float* inputs, const float* embeddings, float* grad, const int* offsets_host, float weight, uint32_t B,
                                       uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, uint32_t gridtype, int align_corners, void* stream) {
    if (B == 0) return PN_OK;
    PN_REQUIRE(inputs && embeddings && grad && offsets_host);
    if (D != 3)  // gridencoder.cu:629-634: D = 2, 4, 5
        return pn_grid_nd_grad_tv_launch(inputs, embeddings, grad, offsets_host, weight, B, D, C, L, S, H, gridtype, align_corners, (hipStream_t)stream);
    PN_REQUIRE(D == 3 && (C == 1 || C == 2 || C == 4 || C == 8) && gridtype <= 1);
    PnGridLevels lv;
    if (pn_fill_grid_levels(&lv, offsets_host, L, C, S, H, gridtype, align_corners)) { PN_REQUIRE(L >= 1 && L <= PN_MAX_LEVELS); }
    dim3 grid(pn_div_up(B, 256), L, 1);
    hipStream_t st = (hipStream_t)stream;
    switch (C) {
        case 1: k_grad_tv<1><<<grid, 256, 0, st>>>(inputs, embeddings, grad, lv, weight, B, align_corners); break;
        case 2: k_grad_tv<2><<<grid, 256, 0, st>>>(inputs, embeddings, grad, lv, weight, B, align_corners); break;
        case 4: k_grad_tv<4><<<grid, 256, 0, st>>>(inputs, embeddings, grad, lv, weight, B, align_corners); break;
        default: k_grad_tv<8><<<grid, 256, 0, st>>>(inputs, embeddings, grad, lv, weight, B, align_corners); break;
    }
    PN_LAUNCH_CHECK();
    return PN_OK;
}

extern "C" int pn_sh_encode_backward(const float* grad, const float* inputs, uint32_t B, uint32_t D, uint32_t C, const float* dy_dx, float* grad_inputs,
                                     void* stream) {
    (void)inputs;
    if (B == 0) return PN_OK;
    PN_REQUIRE(grad && dy_dx && grad_inputs && D == 3 && C >= 1 && C <= 8);
    k_sh_backward<<<pn_div_up((uint64_t)B * 3, 256), 256, 0, (hipStream_t)stream>>>(grad, B, C, dy_dx, grad_inputs);
    PN_LAUNCH_CHECK();
    return PN_OK;
}
