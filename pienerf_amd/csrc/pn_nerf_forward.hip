// Hash-grid / SH encoders and the fused NeRFNetwork.forward kernel for gfx950.
//
// Reference: gridencoder/src/gridencoder.cu:50-245 (kernel_grid), shencoder/src/shencoder.cu:27-123 (kernel_sh),
// nerf/network.py:98-127 (forward), nerf/activation.py (trunc_exp) — paths relative to /root/reference.
//
// Fused kernel layout (one wave = 32 samples, two lanes per sample):
//   lane l: sample s = l & 31, half h = l >> 5.  Half h gathers hash levels [8h, 8h+8) -> 16 features per lane.
//   Every dense layer is D = W·X^T on v_mfma_f32_32x32x16_bf16 at fp32 accuracy (three-way bf16 split, see k_nerf_forward):
//     A operand (lane l) = 8 K-elements of W[out = tile*32 + (l&31)][.] — pre-permuted and pre-split on the host, streamed from LDS
//     B operand (lane l) = 8 of this lane's own activation registers, split in registers
//     D layout: lane l holds out-rows (r&3) + 8*(r>>2) + 4h, r = 0..15, of column s.
//   The K index of (chunk kc, lane half h, element e) is whatever feature that lane holds in register 8 kc + e, so the D layout of
//   one layer is directly the B layout of the next: activations never leave registers; the only LDS traffic is the 61 KB weight
//   image shared by all waves of a workgroup.
#include <math.h>
#include <stdlib.h>

#include "pn_common.h"
#include "pn_encoders.h"

#include "pn_net_tile.h"
#include "pn_sh_bands.h"

// ------------------------------------------------------------------------------------------------ level table
int pn_fill_grid_levels(PnGridLevels* g, const int* offsets_host, uint32_t L, uint32_t C, float S, uint32_t H, uint32_t gridtype,
                        int align_corners) {
    if (L == 0 || L > PN_MAX_LEVELS) return PN_ERR_ARG;
    g->L = L;
    g->C = C;
    for (uint32_t l = 0; l < L; l++) {
        const float scale = exp2f(l * S) * H - 1.0f;            // gridencoder.cu:133
        const uint32_t res = (uint32_t)ceilf(scale) + 1;        // :134
        const uint32_t hs = (uint32_t)(offsets_host[l + 1] - offsets_host[l]);
        // replay get_grid_index's stride loop (gridencoder.cu:65-84) to learn whether this level is hashed
        uint32_t stride = 1, dims = 0;
        for (uint32_t d = 0; d < 3 && stride <= hs; d++) { stride *= align_corners ? res : (res + 1); dims++; }
        const bool hashed = (gridtype == 0 && stride > hs);
        g->offset[l] = (uint32_t)offsets_host[l];
        g->hashmap_size[l] = hs;
        g->resolution[l] = res;
        g->scale[l] = scale;
        g->dense[l] = hashed ? 0u : dims;  // number of dims that enter the direct index (3 = fully dense)
        g->mask[l] = (hs & (hs - 1)) == 0 ? hs - 1 : 0u;
        // fully dense, untiled: max index = (res+1)^3 - 1 < stride <= hs, so `% hs` is the identity
        g->nomod[l] = (!hashed && dims == 3 && !align_corners && gridtype == 0) ? 1u : 0u;
    }
    return PN_OK;
}

// ------------------------------------------------------------------------------------------------ op-level grid encoder

// One thread per (sample, level); blockIdx.y = level keeps one level's table hot in the XCD L2s (gridencoder.cu:103,388).
// T = float: kernel_grid<float,3,C>.  T = _Float16: kernel_grid<at::Half,3,C> — the table and the outputs are half, positions and weights
// stay float, and `results[ch] += w * grid[index + ch]` rounds the float product to half and adds half + half (c10::Half operators).
template <uint32_t C, typename T>
__device__ __forceinline__ void grid_encode_one(float in0, float in1, float in2, const T* __restrict__ emb, const PnGridLevels& lv, uint32_t level, int align_corners,
                                                uint32_t interp, T (&res)[C]) {
#pragma unroll
    for (uint32_t c = 0; c < C; c++) res[c] = (T)0.0f;
    if (in0 < 0 || in0 > 1 || in1 < 0 || in1 > 1 || in2 < 0 || in2 > 1) return;  // gridencoder.cu:113-133
    const T* __restrict__ table = emb + (size_t)lv.offset[level] * C;
    const LevelIdx LI = level_idx(lv, level, align_corners);
    const float scale = lv.scale[level];
    float pos[3] = {in0, in1, in2};
    uint32_t pg[3];
#pragma unroll
    for (int d = 0; d < 3; d++) {
        pos[d] = fmaf(pos[d], scale, align_corners ? 0.0f : 0.5f);  // explicit single rounding, as in the oracle
        pg[d] = (uint32_t)floorf(pos[d]);
        pos[d] -= (float)pg[d];
        if (interp == 1) pos[d] = pos[d] * pos[d] * (3.0f - 2.0f * pos[d]);
    }
#pragma unroll
    for (uint32_t idx = 0; idx < 8; idx++) {
        float w = 1;
        uint32_t pl[3];
#pragma unroll
        for (int d = 0; d < 3; d++) {
            if ((idx & (1u << d)) == 0) { w *= 1 - pos[d]; pl[d] = pg[d]; }
            else { w *= pos[d]; pl[d] = pg[d] + 1; }
        }
        const uint32_t index = grid_index3(LI, pl[0], pl[1], pl[2]) * C;
        if (sizeof(T) == 4 && C == 2) {
            const float2 v = *reinterpret_cast<const float2*>(table + index);
            res[0] += (T)(w * v.x);
            res[1] += (T)(w * v.y);
        } else {
#pragma unroll
            for (uint32_t c = 0; c < C; c++) res[c] = res[c] + rounded_product<T>(w, table[index + c]);
        }
    }
}

// [L,B,C] output (the reference kernel's own layout, gridencoder.cu:105): blockIdx.y = level, a thread per sample — the launch sweeps one level's
// table at a time (it stays in the XCD L2s), neighbouring lanes write neighbouring rows.
template <uint32_t C, typename T>
__global__ void __launch_bounds__(256) k_grid_encode(const float* __restrict__ inputs, const T* __restrict__ emb, PnGridLevels lv, uint32_t B,
                                                     int align_corners, uint32_t interp, T* __restrict__ outputs) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const uint32_t level = blockIdx.y;
    T res[C];
    grid_encode_one<C, T>(inputs[b * 3], inputs[b * 3 + 1], inputs[b * 3 + 2], emb, lv, level, align_corners, interp, res);
    T* out = outputs + ((size_t)level * B + b) * C;
#pragma unroll
    for (uint32_t c = 0; c < C; c++) out[c] = res[c];
}

// [B,L*C] output (what gridencoder/grid.py:57 obtains with an extra permute pass).  A thread per sample that wrote its C values of every level
// straight to its row put neighbouring lanes' stores L*C*4 bytes apart (0.309 ms per 1.02 M samples against 0.172 for [L,B,C]); a thread per
// (sample, level) with the level varying fastest writes coalesced but gathers from all L tables at once (0.306 ms: the tables no longer take turns
// in L2).  Here a workgroup keeps its 256 samples, walks the levels like the [L,B,C] launch does — the workgroups of a launch move through the
// levels roughly together — collects the rows in LDS (row stride padded by one bank) and writes them out whole.
template <uint32_t C, typename T>
__global__ void __launch_bounds__(256) k_grid_encode_rows(const float* __restrict__ inputs, const T* __restrict__ emb, PnGridLevels lv, uint32_t B,
                                                          int align_corners, uint32_t interp, T* __restrict__ outputs) {
    extern __shared__ __attribute__((aligned(16))) unsigned char rows_raw[];
    T* rows = reinterpret_cast<T*>(rows_raw);
    const uint32_t row = lv.L * C, stride = row + (sizeof(T) == 4 ? 1 : 2);
    const uint32_t b0 = blockIdx.x * 256u, b = b0 + threadIdx.x;
    float in0 = -1.f, in1 = -1.f, in2 = -1.f;  // past the end: encoded as out of range, never written
    if (b < B) { in0 = inputs[b * 3]; in1 = inputs[b * 3 + 1]; in2 = inputs[b * 3 + 2]; }
    for (uint32_t level = 0; level < lv.L; level++) {
        T res[C];
        grid_encode_one<C, T>(in0, in1, in2, emb, lv, level, align_corners, interp, res);
#pragma unroll
        for (uint32_t c = 0; c < C; c++) rows[threadIdx.x * stride + level * C + c] = res[c];
    }
    __syncthreads();
    const uint32_t n_rows = min(256u, B - b0);
    for (uint32_t i = threadIdx.x; i < n_rows * row; i += 256) outputs[(size_t)b0 * row + i] = rows[(i / row) * stride + (i % row)];
}

template <typename T>
static int grid_encode_launch(const float* inputs, const T* embeddings, const int* offsets_host, T* outputs, uint32_t B, uint32_t D, uint32_t C,
                              uint32_t L, float S, uint32_t H, uint32_t gridtype, int align_corners, uint32_t interp, int out_bl_major,
                              PnGridLevels* lv_out, hipStream_t st) {
    PN_REQUIRE(inputs && embeddings && offsets_host && outputs);
    PN_REQUIRE(D == 3);                                   // D = 2, 4, 5 (gridencoder.cu:386-395): fp32 in pn_grid_nd.hip; the half table form is D = 3 only
    PN_REQUIRE(C == 1 || C == 2 || C == 4 || C == 8);     // gridencoder.cu:376-382
    PN_REQUIRE(gridtype <= 1 && interp <= 1);
    PnGridLevels lv;
    if (pn_fill_grid_levels(&lv, offsets_host, L, C, S, H, gridtype, align_corners)) { PN_REQUIRE(L >= 1 && L <= PN_MAX_LEVELS); }
    if (out_bl_major) {
        const size_t lds = (size_t)256 * (L * C + (sizeof(T) == 4 ? 1 : 2)) * sizeof(T);
        PN_REQUIRE(lds <= 150 * 1024);
        const dim3 grid(pn_div_up(B, 256), 1, 1);
#define PN_ROWS_LAUNCH(C_)                                                                                                                        \
    do {                                                                                                                                          \
        if (lds > 64 * 1024) /* opted into per call: rows this long (C = 8 with 16 levels) are not on any hot path */                             \
            PN_HIP_CHECK(hipFuncSetAttribute((const void*)k_grid_encode_rows<C_, T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));     \
        k_grid_encode_rows<C_, T><<<grid, 256, lds, st>>>(inputs, embeddings, lv, B, align_corners, interp, outputs);                            \
    } while (0)
        switch (C) {
            case 1: PN_ROWS_LAUNCH(1); break;
            case 2: PN_ROWS_LAUNCH(2); break;
            case 4: PN_ROWS_LAUNCH(4); break;
            default: PN_ROWS_LAUNCH(8); break;
        }
#undef PN_ROWS_LAUNCH
    } else {
        const dim3 grid(pn_div_up(B, 256), L, 1);
        switch (C) {
            case 1: k_grid_encode<1, T><<<grid, 256, 0, st>>>(inputs, embeddings, lv, B, align_corners, interp, outputs); break;
            case 2: k_grid_encode<2, T><<<grid, 256, 0, st>>>(inputs, embeddings, lv, B, align_corners, interp, outputs); break;
            case 4: k_grid_encode<4, T><<<grid, 256, 0, st>>>(inputs, embeddings, lv, B, align_corners, interp, outputs); break;
            default: k_grid_encode<8, T><<<grid, 256, 0, st>>>(inputs, embeddings, lv, B, align_corners, interp, outputs); break;
        }
    }
    PN_LAUNCH_CHECK();
    if (lv_out) *lv_out = lv;
    return PN_OK;
}

extern "C" int pn_grid_encode_forward(const float* inputs, const float* embeddings, const int* offsets_host, float* outputs, uint32_t B, uint32_t D,
                                      uint32_t C, uint32_t L, float S, uint32_t H, float* dy_dx, uint32_t gridtype, int align_corners,
                                      uint32_t interp, int out_bl_major, void* stream) {
    if (B == 0) return PN_OK;  // empty tensors have null data pointers
    if (D != 3)                // gridencoder.cu:393-398: D = 2, 4, 5 (anything else: "GridEncoding: D must be 2, 3, 4, 5" -> PN_ERR_ARG)
        return pn_grid_nd_forward_launch(inputs, embeddings, offsets_host, outputs, B, D, C, L, S, H, dy_dx, gridtype, align_corners, interp, out_bl_major,
                                         (hipStream_t)stream);
    PnGridLevels lv;
    const int rc = grid_encode_launch<float>(inputs, embeddings, offsets_host, outputs, B, D, C, L, S, H, gridtype, align_corners, interp, out_bl_major,
                                             &lv, (hipStream_t)stream);
    if (rc) return rc;
    if (dy_dx) return pn_grid_dy_dx_launch(inputs, embeddings, lv, B, C, align_corners, interp, dy_dx, (hipStream_t)stream);  // training side (pn_encoder_grad.hip)
    return PN_OK;
}

extern "C" int pn_grid_encode_forward_half(const float* inputs, const uint16_t* embeddings, const int* offsets_host, uint16_t* outputs, uint32_t B,
                                           uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, uint32_t gridtype, int align_corners,
                                           uint32_t interp, int out_bl_major, void* stream) {
    if (B == 0) return PN_OK;
    return grid_encode_launch<_Float16>(inputs, reinterpret_cast<const _Float16*>(embeddings), offsets_host, reinterpret_cast<_Float16*>(outputs), B, D,
                                        C, L, S, H, gridtype, align_corners, interp, out_bl_major, nullptr, (hipStream_t)stream);
}


__global__ void __launch_bounds__(256) k_sh_encode(const float* __restrict__ inputs, float* __restrict__ outputs, uint32_t B, uint32_t C) {
    const uint32_t b = threadIdx.x + blockIdx.x * blockDim.x;
    if (b >= B) return;
    float o[16];
    sh16(inputs[b * 3], inputs[b * 3 + 1], inputs[b * 3 + 2], o);
    const uint32_t C2 = C * C;
    for (uint32_t i = 0; i < (C2 < 16u ? C2 : 16u); i++) outputs[(size_t)b * C2 + i] = o[i];
}

// degree 5-8 (shencoder.cu:69-123): bands 0-3 as above, bands 4.. by recurrence (pn_sh_bands.h), written straight to the output row
__global__ void __launch_bounds__(256) k_sh_encode_high(const float* __restrict__ inputs, float* __restrict__ outputs, uint32_t B, uint32_t C) {
    const uint32_t b = threadIdx.x + blockIdx.x * blockDim.x;
    if (b >= B) return;
    const float x = inputs[b * 3], y = inputs[b * 3 + 1], z = inputs[b * 3 + 2];
    float o[16];
    sh16(x, y, z, o);
    float* row = outputs + (size_t)b * C * C;
    for (uint32_t i = 0; i < 16; i++) row[i] = o[i];
    pnsh::high_bands(x, y, z, (int)C, row, nullptr, nullptr, nullptr);
}

extern "C" int pn_sh_encode_forward(const float* inputs, float* outputs, uint32_t B, uint32_t D, uint32_t C, float* dy_dx, void* stream) {
    if (B == 0) return PN_OK;  // empty tensors have null data pointers
    PN_REQUIRE(inputs && outputs && D == 3 && C >= 1 && C <= 8);  // shencoder.cu:42-123 (the wrapper asserts degree <= 8, sphere_harmonics.py:70)
    if (C > 4) k_sh_encode_high<<<pn_div_up(B, 256), 256, 0, (hipStream_t)stream>>>(inputs, outputs, B, C);
    else k_sh_encode<<<pn_div_up(B, 256), 256, 0, (hipStream_t)stream>>>(inputs, outputs, B, C);
    PN_LAUNCH_CHECK();
    if (dy_dx) return pn_sh_dy_dx_launch(inputs, dy_dx, B, C, (hipStream_t)stream);  // training side (pn_encoder_grad.hip)
    return PN_OK;
}

// ------------------------------------------------------------------------------------------------ fused network
// colour-net input slot -> index into cat([SH16, geo15]) (31 = zero pad); low half / high half of the lane pair
static const int PN_MAPL[16] = {16, 17, 18, 23, 24, 25, 26, 0, 1, 2, 3, 4, 5, 6, 7, 8};
static const int PN_MAPU[16] = {19, 20, 21, 22, 27, 28, 29, 30, 9, 10, 11, 12, 13, 14, 15, 31};

// Host: wp[m][lane] = the weight that multiplies activation register m (numbered through the four MFMA layers: 2 x 16, 32, 2 x 16,
// 2 x 32) of lane half lane >> 5, for output row lane & 31 of that layer's tile (see header comment).
static void pack_weights(const float* W0, const float* W1, const float* W2, const float* W3, float* wp) {
    auto krow = [](int q, int h) { const int t = q >> 4, r = q & 15; return t * 32 + (r & 3) + 8 * (r >> 2) + 4 * h; };
    int m = 0;
    for (int t = 0; t < 2; t++)  // layer 0: 32 -> 64
        for (int kk = 0; kk < 16; kk++, m++)
            for (int l = 0; l < 64; l++) wp[m * 64 + l] = W0[(t * 32 + (l & 31)) * 32 + kk + 16 * (l >> 5)];
    for (int q = 0; q < 32; q++, m++)  // layer 1: 64 -> 16 (rows 16..31 zero)
        for (int l = 0; l < 64; l++) wp[m * 64 + l] = ((l & 31) < 16) ? W1[(l & 31) * 64 + krow(q, l >> 5)] : 0.0f;
    for (int t = 0; t < 2; t++)  // layer 2: 31 (+1 pad) -> 64
        for (int kk = 0; kk < 16; kk++, m++)
            for (int l = 0; l < 64; l++) {
                const int ci = (l >> 5) ? PN_MAPU[kk] : PN_MAPL[kk];
                wp[m * 64 + l] = (ci < 31) ? W2[(t * 32 + (l & 31)) * 31 + ci] : 0.0f;
            }
    for (int t = 0; t < 2; t++)  // layer 3: 64 -> 64
        for (int q = 0; q < 32; q++, m++)
            for (int l = 0; l < 64; l++) wp[m * 64 + l] = W3[(t * 32 + (l & 31)) * 64 + krow(q, l >> 5)];
}

// fp32 -> fp16, round to nearest even (what `tensor.to(torch.half)` and v_cvt_f16_f32 do), returned as the 16 payload bits
static uint16_t pn_f2h_bits(float f) {
    uint32_t x;
    memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7fffffffu;
    if (x >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | ((x > 0x7f800000u) ? 0x200u : 0u));  // inf / nan
    if (x >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);                                        // rounds to >= 65520 -> inf
    if (x < 0x33000001u) return (uint16_t)sign;                                                     // <= 2^-25 -> 0 (ties to even)
    int e = (int)(x >> 23) - 127;
    uint32_t m = (x & 0x7fffffu) | 0x800000u;  // 24-bit significand
    int shift = (e < -14) ? (13 + (-14 - e)) : 13;  // bits dropped (subnormal halves drop more)
    uint32_t q = m >> shift, rem = m & ((1u << shift) - 1u), halfway = 1u << (shift - 1);
    if (rem > halfway || (rem == halfway && (q & 1u))) q++;
    if (e < -14) return (uint16_t)(sign | q);  // subnormal (q may carry into the smallest normal: the encoding is continuous)
    return (uint16_t)(sign | (((uint32_t)(e + 15) << 10) + (q - 0x400u)));  // mantissa carry rolls into the exponent
}
static float pn_h2f(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 31u, m = h & 0x3ffu;
    uint32_t x;
    if (e == 0) {
        if (m == 0) x = sign;
        else { float v = (float)m * 5.9604644775390625e-8f; memcpy(&x, &v, 4); x |= sign; }  // m * 2^-24
    } else if (e == 31) x = sign | 0x7f800000u | (m << 13);
    else x = sign | ((e + 112u) << 23) | (m << 13);
    float f;
    memcpy(&f, &x, 4);
    return f;
}

// Both LDS weight images from the five row-major [out,in] matrices.
//   simg (PN_NET_SPLIT_BYTES): operand group (layer, out tile t, K chunk kc) takes the 8 consecutive activation registers m0 + 8 kc + e
//     (e = 0..7) of pack_weights' stream as the 8 K-elements of its lane, each weight cut into three bf16 pieces w = hi + mid + lo
//     (truncation splits: exact, 8 + 8 + 8 significant bits); then the 64 -> 3 layer's fp32 weights for the vector ALU
//     (wlast[h][q][o] = W4[o][krow(q, h)], q = tile*16 + register index of the D layout; an MFMA tile would be 29/32 row padding).
//   himg (PN_NET_HALF_BYTES): the same groups with every weight rounded once to fp16 (autocast's `weight.to(half)`), one piece.
//   ximg (PN_NET_X_BYTES): the same groups with every weight as two fp16 pieces w = hi + lo (round to nearest even both), then the 64 -> 3 layer's fp32 weights.
//     `xs` = the five power-of-two layer scales of net_choose_form (layer l's inputs are carried as xs[l] * value): W0 xs1/xs0, W1 xs2/xs1, W2's geometry
//     columns xs3/xs2 and its SH columns xs3 (the SH basis enters unscaled), W3 xs4/xs3, W4 / xs4 — exact, ReLU commutes with positive factors.
static void build_weight_images(const float* W0, const float* W1, const float* W2, const float* W3, const float* W4, unsigned char* simg,
                                unsigned char* himg, unsigned char* ximg, const double* xs) {
    float* host = new float[160 * 64];
    float* hostx = new float[160 * 64];
    pack_weights(W0, W1, W2, W3, host);
    {
        float* X = new float[64 * 32 + 16 * 64 + 64 * 31 + 64 * 64];
        float *X0 = X, *X1 = X0 + 64 * 32, *X2 = X1 + 16 * 64, *X3 = X2 + 64 * 31;
        for (int i = 0; i < 64 * 32; i++) X0[i] = (float)((double)W0[i] * (xs[1] / xs[0]));
        for (int i = 0; i < 16 * 64; i++) X1[i] = (float)((double)W1[i] * (xs[2] / xs[1]));
        for (int i = 0; i < 64; i++)
            for (int j = 0; j < 31; j++) X2[i * 31 + j] = (float)((double)W2[i * 31 + j] * (j < 16 ? xs[3] : xs[3] / xs[2]));
        for (int i = 0; i < 64 * 64; i++) X3[i] = (float)((double)W3[i] * (xs[4] / xs[3]));
        pack_weights(X0, X1, X2, X3, hostx);
        delete[] X;
    }
    uint16_t* s16 = reinterpret_cast<uint16_t*>(simg);
    uint16_t* h16 = reinterpret_cast<uint16_t*>(himg);
    uint16_t* x16 = reinterpret_cast<uint16_t*>(ximg);
    int G = 0;
    auto emit = [&](int m0) {
        for (int l = 0; l < 64; l++)
            for (int e2 = 0; e2 < 8; e2++) {
                float v = host[(m0 + e2) * 64 + l];
                h16[((size_t)G * 64 + l) * 8 + e2] = pn_f2h_bits(v);
                {
                    const float vx = hostx[(m0 + e2) * 64 + l];
                    const uint16_t xh = pn_f2h_bits(vx);
                    const float r = vx - pn_h2f(xh);   // exact: the remainder of a round-to-nearest fp16 has at most 13 significant bits
                    x16[((size_t)(G * 2 + 0) * 64 + l) * 8 + e2] = xh;
                    x16[((size_t)(G * 2 + 1) * 64 + l) * 8 + e2] = pn_f2h_bits(r);
                }
                for (int p = 0; p < 3; p++) {
                    uint32_t u;
                    memcpy(&u, &v, 4);
                    u &= 0xffff0000u;
                    float h;
                    memcpy(&h, &u, 4);
                    s16[((size_t)(G * 3 + p) * 64 + l) * 8 + e2] = (uint16_t)(u >> 16);
                    v -= h;
                }
            }
        G++;
    };
    for (int t = 0; t < 2; t++) for (int kc = 0; kc < 2; kc++) emit(0 + t * 16 + 8 * kc);    // layer 0: groups 0..3
    for (int kc = 0; kc < 4; kc++) emit(32 + 8 * kc);                                         // layer 1: groups 4..7
    for (int t = 0; t < 2; t++) for (int kc = 0; kc < 2; kc++) emit(64 + t * 16 + 8 * kc);   // layer 2: groups 8..11
    for (int t = 0; t < 2; t++) for (int kc = 0; kc < 4; kc++) emit(96 + t * 32 + 8 * kc);   // layer 3: groups 12..19
    float* wlast = reinterpret_cast<float*>(simg + PN_NET_SPLIT_W_BYTES);
    float* wlast_h = reinterpret_cast<float*>(himg + PN_NET_HALF_W_BYTES);
    float* wlast_x = reinterpret_cast<float*>(ximg + PN_NET_X_W_BYTES);
    for (int h = 0; h < 2; h++)
        for (int q = 0; q < 32; q++)
            for (int o = 0; o < 3; o++) {
                const int t = q >> 4, r = q & 15;
                const float w = W4[o * 64 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * h];
                wlast[(h * 32 + q) * 3 + o] = w;
                wlast_x[(h * 32 + q) * 3 + o] = (float)((double)w / xs[4]);
                wlast_h[(h * 32 + q) * 3 + o] = pn_h2f(pn_f2h_bits(w));
            }
    delete[] host;
    delete[] hostx;
}

static int net_fail(pn_net* n, hipError_t e, const char* what) {
    snprintf(pn_err_buf, sizeof(pn_err_buf), "%s: %s", what, hipGetErrorString(e));
    pn_net_destroy(n);
    return PN_ERR_HIP;
}

// largest |entry| of the hash tables (bit pattern of a non-negative float orders like the float)
__global__ void __launch_bounds__(256) k_table_absmax(const float* __restrict__ emb, uint32_t n, unsigned* __restrict__ out) {
    unsigned m = 0;
    for (uint32_t i = threadIdx.x + blockIdx.x * blockDim.x; i < n; i += gridDim.x * blockDim.x) {
        const float v = fabsf(emb[i]);
        m = max(m, v == v ? __float_as_uint(v) : 0x7f800000u);   // NaN counts as infinite
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}

// Whether the fp16 hi/lo form runs for these weights and tables (pn_common.h: pn_net::x_ok), and the power-of-two scale every layer's inputs are carried
// at.  Interval bound, layer by layer: |out_i| <= sum_j |W_ij| max|in_j|; ReLU cannot raise it, the SH basis is below 4 in magnitude up to degree 4 on unit
// directions.  Layer l's inputs travel as xs[l] * value with xs[l] the power of two that puts their bound into [2^13, 2^14]: far inside fp16's range, and a
// value's lo piece stays a normal fp16 number down to 2^-17 of the bound (hi + lo then carries 22 bits).  The scales live in the weight image
// (build_weight_images), the kernel multiplies the features by xs[0] and the 16 outputs of the density net by 1 / xs[2]; nothing else.
// xs[0] features, xs[1] hidden (density), xs[2] sigma logit + geometry features, xs[3] / xs[4] the colour net's hidden layers.
static void net_choose_form(pn_net* n, const float* W0, const float* W1, const float* W2, const float* W3, float table_max, double* xs) {
    auto row_bound = [](const float* W, int rows, int cols, int split, double in_a, double in_b) {   // columns < split see in_a, the others in_b
        double worst = 0.0;
        for (int i = 0; i < rows; i++) {
            double s2 = 0.0;
            for (int j = 0; j < cols; j++) s2 += fabs((double)W[i * cols + j]) * (j < split ? in_a : in_b);
            worst = std::max(worst, s2);
        }
        return worst;
    };
    n->x_ok = 0;
    n->x_scale = n->x_rscale = 1.0f;
    for (int l = 0; l < 5; l++) xs[l] = 1.0;
    const char* form = getenv("PN_NET_FORM");   // "bf16": always the three-way bf16 split (A/B runs, tests)
    if (form && strcmp(form, "bf16") == 0) return;
    double bb[5];
    bb[0] = (double)table_max;
    bb[1] = row_bound(W0, 64, 32, 32, bb[0], 0.0);                            // density layer 0 -> hidden
    bb[2] = row_bound(W1, 16, 64, 64, bb[1], 0.0);                            // density layer 1 -> sigma logit + geometry features
    bb[3] = row_bound(W2, 64, 31, 16, 4.0, bb[2]);                            // colour layer 0 on [SH16 | geo15]
    bb[4] = row_bound(W3, 64, 64, 64, bb[3], 0.0);                            // colour layer 1 (its outputs go to the fp32 vector layer)
    double sc[5];
    for (int l = 0; l < 5; l++) {
        if (!(bb[l] > 0.0) || !std::isfinite(bb[l])) return;                  // zero / NaN / infinite somewhere: the bf16 form takes anything
        const int k = (int)floor(log2(16384.0 / bb[l]));                      // bound * 2^k in (2^13, 2^14]
        if (k < -100 || k > 100) return;                                      // (fp32 products of two scales must stay finite)
        sc[l] = ldexp(1.0, k);
    }
    for (int l = 0; l < 5; l++) xs[l] = sc[l];
    n->x_ok = 1;
    n->x_scale = (float)xs[0];
    n->x_rscale = (float)(1.0 / xs[2]);
}

// stages the three images in pinned memory and uploads them behind whatever `stream` holds; no allocation; one stream synchronisation (the tables' abs-max)
static int net_upload_weights(pn_net* n, const float* W0, const float* W1, const float* W2, const float* W3, const float* W4, hipStream_t st) {
    // the tables' largest entry (one small reduction + a 4-byte read-back: this function waits for the host's packing anyway) -> the form of the fp32 network
    unsigned* d_max = reinterpret_cast<unsigned*>(reinterpret_cast<unsigned char*>(n->wx) + PN_NET_X_BYTES);
    PN_HIP_CHECK(hipMemsetAsync(d_max, 0, 4, st));
    k_table_absmax<<<512, 256, 0, st>>>(n->embeddings, n->n_entries * 2u, d_max);
    unsigned bits = 0;
    PN_HIP_CHECK(hipMemcpyAsync(&bits, d_max, 4, hipMemcpyDeviceToHost, st));
    PN_HIP_CHECK(hipStreamSynchronize(st));
    float table_max;
    memcpy(&table_max, &bits, 4);
    double xs[5];
    const int form_before = n->x_ok;
    net_choose_form(n, W0, W1, W2, W3, table_max, xs);
    if (n->x_scales && n->x_ok != form_before) n->form_epoch++;   // (x_scales is null during pn_net_create: nothing captured yet)
    PN_HIP_CHECK(hipEventSynchronize(n->stage_done));  // the previous upload has finished reading the staging buffer (normally long ago)
    unsigned char* simg = reinterpret_cast<unsigned char*>(n->stage);
    unsigned char* himg = simg + PN_NET_SPLIT_BYTES;
    unsigned char* ximg = himg + PN_NET_HALF_BYTES;
    build_weight_images(W0, W1, W2, W3, W4, simg, himg, ximg, xs);
    PN_HIP_CHECK(hipMemcpyAsync(n->wsplit, simg, PN_NET_SPLIT_BYTES, hipMemcpyHostToDevice, st));
    PN_HIP_CHECK(hipMemcpyAsync(n->whalf, himg, PN_NET_HALF_BYTES, hipMemcpyHostToDevice, st));
    const float tail[4] = {0.0f, 0.0f, n->x_scale, n->x_rscale};   // the scales travel with the image they are folded into (pn_common.h: x_scales)
    memcpy(ximg + PN_NET_X_BYTES, tail, 16);
    PN_HIP_CHECK(hipMemcpyAsync(n->wx, ximg, PN_NET_X_BYTES + 16, hipMemcpyHostToDevice, st));
    n->x_scales = reinterpret_cast<const float*>(reinterpret_cast<unsigned char*>(n->wx) + PN_NET_X_BYTES) + 2;
    PN_HIP_CHECK(hipEventRecord(n->stage_done, st));
    return PN_OK;
}

__global__ void __launch_bounds__(256) k_table_to_half(const float2* __restrict__ emb, uint32_t n, uint32_t* __restrict__ out) {
    for (uint32_t i = threadIdx.x + blockIdx.x * blockDim.x; i < n; i += gridDim.x * blockDim.x) {
        const float2 v = emb[i];
        const _Float16 a = (_Float16)v.x, b = (_Float16)v.y;  // v_cvt_f16_f32: round to nearest even = tensor.to(torch.half)
        out[i] = (uint32_t)__builtin_bit_cast(uint16_t, a) | ((uint32_t)__builtin_bit_cast(uint16_t, b) << 16);
    }
}

extern "C" int pn_net_create(pn_net** out, const float* embeddings, const int* offsets_host, uint32_t L, uint32_t C, float per_level_scale_log2,
                             uint32_t base_resolution, float bound, const float* W0, const float* W1, const float* W2, const float* W3,
                             const float* W4, void* stream) {
    PN_REQUIRE(out && embeddings && offsets_host && W0 && W1 && W2 && W3 && W4);
    PN_REQUIRE(L == 16 && C == 2);  // the architecture of nerf/network.py:14-95 / nerf/encoding.py:40-70
    pn_net* n = new pn_net();
    memset(n, 0, sizeof(*n));
    if (pn_fill_grid_levels(&n->levels, offsets_host, L, C, per_level_scale_log2, base_resolution, 0, 0)) { delete n; return PN_ERR_ARG; }
    PnFusedLevel fl[16];
    PnByteLevel bl[16];
    for (uint32_t l = 0; l < L; l++) {
        const PnGridLevels& g = n->levels;
        const bool dense = g.dense[l] == 3 && g.nomod[l];
        const bool hashed = g.dense[l] == 0 && g.mask[l] != 0;
        if (!dense && !hashed) { delete n; PN_REQUIRE(!"level is neither fully dense nor hashed into a power-of-two table"); }
        const uint32_t s1 = g.resolution[l] + 1;
        fl[l] = PnFusedLevel{g.scale[l], g.offset[l], dense ? s1 : 2654435761u, dense ? s1 * s1 : 805459861u, g.mask[l], dense ? 1u : 0u, dense ? 0xffffffffu : 0u, dense ? 0xffffffffu : g.mask[l]};
        if (dense && (uint64_t)s1 * s1 * 8u >= (1u << 24)) { delete n; PN_REQUIRE(!"dense level with a z stride of 2^24 bytes or more"); }
        bl[l] = dense ? PnByteLevel{g.scale[l], g.offset[l] * 8u, 0u, 0u, s1 * 8u, s1 * s1 * 8u, 0xffffffffu, 0u}
                      : PnByteLevel{g.scale[l], g.offset[l] * 8u, 2654435761u * 8u, 805459861u * 8u, 0u, 0u, g.mask[l] * 8u, g.mask[l] * 8u};
    }
    if ((uint64_t)offsets_host[L] * 8u >= (1ull << 32)) { delete n; PN_REQUIRE(!"hash tables of 4 GiB or more (the fp32 kernels address them by 32-bit byte offsets)"); }
    n->embeddings = embeddings;
    n->bound = bound;
    n->n_entries = (uint32_t)offsets_host[L];
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMalloc((void**)&n->wsplit, PN_NET_SPLIT_BYTES);
    if (e == hipSuccess) e = hipMalloc((void**)&n->whalf, PN_NET_HALF_BYTES);
    if (e == hipSuccess) e = hipMalloc((void**)&n->wx, PN_NET_X_BYTES + 16);   // (+ the tail: the tables' abs-max word of net_upload_weights, the two scales)
    if (e == hipSuccess) e = hipMalloc((void**)&n->fused_levels, sizeof(fl));
    if (e == hipSuccess) e = hipMalloc((void**)&n->byte_levels, sizeof(bl));
    if (e == hipSuccess) e = hipHostMalloc((void**)&n->stage, PN_NET_SPLIT_BYTES + PN_NET_HALF_BYTES + PN_NET_X_BYTES + 16);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&n->stage_done, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventRecord(n->stage_done, st);
    if (e == hipSuccess) e = hipMemcpyAsync(n->fused_levels, fl, sizeof(fl), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(n->byte_levels, bl, sizeof(bl), hipMemcpyHostToDevice, st);
    if (e != hipSuccess) return net_fail(n, e, "pn_net_create");
    const int rc = net_upload_weights(n, W0, W1, W2, W3, W4, st);
    if (rc) { pn_net_destroy(n); return rc; }
    e = hipStreamSynchronize(st);  // `fl` lives on this stack frame
    if (e != hipSuccess) return net_fail(n, e, "pn_net_create");
    *out = n;
    return PN_OK;
}

extern "C" int pn_net_update(pn_net* n, const float* embeddings, const float* W0, const float* W1, const float* W2, const float* W3,
                             const float* W4, void* stream) {
    PN_REQUIRE(n && embeddings && W0 && W1 && W2 && W3 && W4);
    hipStream_t st = (hipStream_t)stream;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    PN_HIP_CHECK(hipStreamIsCapturing(st, &cs));
    if (cs != hipStreamCaptureStatusNone) {  // the packing runs on the host and reads host weights: it cannot be part of a captured graph
        snprintf(pn_err_buf, sizeof(pn_err_buf), "pn_net_update: called while the stream is being captured into a HIP graph; refresh the network "
                                                 "weights before capture (a replay would re-upload the weights packed at capture time)");
        return PN_ERR_ARG;
    }
    n->embeddings = embeddings;
    const int rc = net_upload_weights(n, W0, W1, W2, W3, W4, st);
    if (rc) return rc;
    if (n->emb_half) {  // keep the fp16 copy of the tables in step with the fp32 master
        k_table_to_half<<<1024, 256, 0, st>>>(reinterpret_cast<const float2*>(n->embeddings), n->n_entries, reinterpret_cast<uint32_t*>(n->emb_half));
        PN_LAUNCH_CHECK();
    }
    return PN_OK;
}

extern "C" int pn_net_form(const pn_net* n) { return (n && n->x_ok) ? 2 : 0; }
extern "C" int pn_net_form_epoch(const pn_net* n) { return n ? n->form_epoch : 0; }

extern "C" int pn_net_enable_half(pn_net* n, void* stream) {
    PN_REQUIRE(n);
    if (n->emb_half) return PN_OK;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    PN_HIP_CHECK(hipStreamIsCapturing((hipStream_t)stream, &cs));
    if (cs != hipStreamCaptureStatusNone) {
        snprintf(pn_err_buf, sizeof(pn_err_buf), "pn_net_enable_half: the fp16 tables must be created before stream capture (render one frame under autocast first)");
        return PN_ERR_ARG;
    }
    PN_HIP_CHECK(hipMalloc(&n->emb_half, (size_t)n->n_entries * 4));
    k_table_to_half<<<1024, 256, 0, (hipStream_t)stream>>>(reinterpret_cast<const float2*>(n->embeddings), n->n_entries,
                                                          reinterpret_cast<uint32_t*>(n->emb_half));
    PN_LAUNCH_CHECK();
    return PN_OK;
}

extern "C" void pn_net_destroy(pn_net* n) {
    if (!n) return;
    if (n->wsplit) (void)hipFree(n->wsplit);
    if (n->whalf) (void)hipFree(n->whalf);
    if (n->wx) (void)hipFree(n->wx);
    if (n->emb_half) (void)hipFree(n->emb_half);
    if (n->fused_levels) (void)hipFree(n->fused_levels);
    if (n->byte_levels) (void)hipFree(n->byte_levels);
    if (n->stage) (void)hipHostFree(n->stage);
    if (n->stage_done) (void)hipEventDestroy(n->stage_done);
    delete n;
}


// Workgroup = 8 waves sharing ONE 61 KB LDS weight image, one workgroup per CU = 2 waves per SIMD (rounds 1-2: two 4-wave workgroups per CU with an
// image each — the same waves with 122 KB of LDS, which kept the other render lanes' march workgroups (48 KB) off the CU; round 3: in-loop launches
// 0.280 -> 0.265 ms per frame, pipelined step unchanged.  16 waves per CU (-DPN_BF_WAVES=8 with PN_NERF_BLOCKS=512) are faster alone, 0.249 ms, and cost
// the pipelined step 5 %: the march kernels want the wave slots).
// MINW = waves per SIMD the register allocator must leave room for; LU = how many hash levels' gathers are in flight per lane.
#ifndef PN_BF_WAVES
#define PN_BF_WAVES 8
#endif
// X: the fp16 hi/lo form of the dense layers (pn_common.h: pn_net::wx; `wsplit` is then that image, sf, rsf = the device words pn_net::x_scales)
template <int MINW, int LU, bool X = false>
__global__ void __launch_bounds__(PN_BF_WAVES * 64, MINW) k_nerf_forward(const PnByteLevel* __restrict__ lv, const float* __restrict__ emb,
                                                                          const uint4* __restrict__ wsplit, float bound, const float* __restrict__ xyzs,
                                                                          const float* __restrict__ dirs, const int* __restrict__ list,
                                                                          const int* __restrict__ count_dev, uint32_t M_arg, float density_scale,
                                                                          float* __restrict__ sigmas, float* __restrict__ rgbs,
                                                                          float* __restrict__ geo, int sigma_only, const float* __restrict__ x_scales = nullptr) {
    extern __shared__ __attribute__((aligned(16))) uint4 wimg[];  // the weight image, then the 16 level records
    const float sf = X ? x_scales[0] : 1.0f, rsf = X ? x_scales[1] : 1.0f;  // device words beside the image: a captured launch follows pn_net_update
    constexpr int PN_IMG_BYTES = X ? PN_NET_X_BYTES : PN_NET_SPLIT_BYTES;
    const uint32_t M = count_dev ? (uint32_t)*count_dev : M_arg;
    const uint32_t n_tiles = (M + 31) / 32;
    const uint32_t waves_total = gridDim.x * PN_BF_WAVES;
    const uint32_t wave = blockIdx.x * PN_BF_WAVES + (threadIdx.x >> 6);
    if (blockIdx.x * PN_BF_WAVES >= n_tiles) return;  // no tile for any wave of this block
    for (int i = threadIdx.x; i < PN_IMG_BYTES / 16; i += PN_BF_WAVES * 64) wimg[i] = wsplit[i];
    if (threadIdx.x < 16 * sizeof(PnFusedLevel) / 16) wimg[PN_IMG_BYTES / 16 + threadIdx.x] = reinterpret_cast<const uint4*>(lv)[threadIdx.x];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int s = lane & 31, half = lane >> 5;
    const uint4* __restrict__ wl = wimg + lane;
    const PnByteLevel* lds_lv = reinterpret_cast<const PnByteLevel*>(wimg + PN_IMG_BYTES / 16) + 8 * half;
    const float inv2b = 1.0f / (2 * bound);

    for (uint32_t tile = wave; tile < n_tiles; tile += waves_total) {
        const uint32_t li = tile * 32 + s;
        const bool valid = li < M;
        const uint32_t slot = valid ? (list ? (uint32_t)list[li] : li) : 0u;
        float x = 0.f, y = 0.f, z = 0.f, dx = 0.f, dy = 0.f, dz = 1.f;
        if (valid) {
            x = xyzs[slot * 3]; y = xyzs[slot * 3 + 1]; z = xyzs[slot * 3 + 2];
            dx = dirs[slot * 3]; dy = dirs[slot * 3 + 1]; dz = dirs[slot * 3 + 2];
        }
        f32x16 h2;
        if (X) h2 = tile_sigma_net_x<LU>(lds_lv, emb, wl, half, bound, inv2b, x, y, z, sf);   // xs[2] * outputs
        else h2 = tile_sigma_net<LU>(lds_lv, emb, wl, half, bound, inv2b, x, y, z);
        const float sigma_logit = X ? h2[0] * rsf : h2[0];  // row 0 lives in the low half's register 0
        if (geo || sigma_only) {  // NeRFNetwork.density (network.py:129-146): sigma = exp(h[0]), geo_feat = h[1:16]; no colour net (kernel-uniform)
            if (valid) {
                if (half == 0) sigmas[slot] = tile_sigma_out(density_scale, sigma_logit);
                if (geo) {
                    float* __restrict__ g = geo + (size_t)slot * 15;
#pragma unroll
                    for (int r = 0; r < 8; r++) {  // this lane's rows (r&3) + 8*(r>>2) + 4*half of the 16 outputs
                        const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
                        if (row >= 1) g[row - 1] = X ? h2[r] * rsf : h2[r];
                    }
                }
            }
            continue;
        }
        __builtin_amdgcn_sched_barrier(0);
        float e[3];
        if (X) tile_color_net_x(wl, wimg, half, h2, dx, dy, dz, e);
        else tile_color_net(wl, wimg, half, h2, dx, dy, dz, e);
        if (valid && half == 0) {
            sigmas[slot] = tile_sigma_out(density_scale, sigma_logit);
            rgbs[slot * 3 + 0] = tile_rgb_out(e[0]);
            rgbs[slot * 3 + 1] = tile_rgb_out(e[1]);
            rgbs[slot * 3 + 2] = tile_rgb_out(e[2]);
        }
    }
}

#define PN_H_WAVES 4
template <int MINW, int LU>
__global__ void __launch_bounds__(PN_H_WAVES * 64, MINW) k_nerf_forward_h(const PnFusedLevel* __restrict__ lv, const uint32_t* __restrict__ emb_h,
                                                                           const uint4* __restrict__ whalf, float bound, const float* __restrict__ xyzs,
                                                                           const float* __restrict__ dirs, const int* __restrict__ list,
                                                                           const int* __restrict__ count_dev, uint32_t M_arg, float density_scale,
                                                                           float* __restrict__ sigmas, float* __restrict__ rgbs,
                                                                           float* __restrict__ geo, uint32_t emb_bytes, int sigma_only) {
    extern __shared__ __attribute__((aligned(16))) uint4 wimg[];  // PN_NET_HALF_BYTES
    // raw buffer over the whole fp16 table: stride 0, num_records = bytes (out-of-range offsets read 0), gfx9 dword-format flags
    const __amdgpu_buffer_rsrc_t emb_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(emb_h), 0, (int)emb_bytes, 0x00020000);
    const uint32_t M = count_dev ? (uint32_t)*count_dev : M_arg;
    const uint32_t n_tiles = (M + 31) / 32;
    const uint32_t waves_total = gridDim.x * PN_H_WAVES;
    const uint32_t wave = blockIdx.x * PN_H_WAVES + (threadIdx.x >> 6);
    if (blockIdx.x * PN_H_WAVES >= n_tiles) return;
    for (int i = threadIdx.x; i < PN_NET_HALF_BYTES / 16; i += PN_H_WAVES * 64) wimg[i] = whalf[i];
    if (threadIdx.x < 16 * sizeof(PnFusedLevel) / 16) wimg[PN_NET_HALF_BYTES / 16 + threadIdx.x] = reinterpret_cast<const uint4*>(lv)[threadIdx.x];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int s = lane & 31, half = lane >> 5;
    const uint4* __restrict__ wl = wimg + lane;

    for (uint32_t tile = wave; tile < n_tiles; tile += waves_total) {
        const uint32_t li = tile * 32 + s;
        const bool valid = li < M;
        const uint32_t slot = valid ? (list ? (uint32_t)list[li] : li) : 0u;
        float x = 0.f, y = 0.f, z = 0.f, dx = 0.f, dy = 0.f, dz = 1.f;
        if (valid) {
            x = xyzs[slot * 3]; y = xyzs[slot * 3 + 1]; z = xyzs[slot * 3 + 2];
            dx = dirs[slot * 3]; dy = dirs[slot * 3 + 1]; dz = dirs[slot * 3 + 2];
        }
        float g2[8];  // this lane's 8 of the 16 outputs, rounded to half: rows (r&3) + 8*(r>>2) + 4*half
        tile_sigma_net_h<LU>(lv, wimg, emb_rsrc, wl, half, bound, x, y, z, g2);
        const float sigma_logit = g2[0];  // row 0 lives in the low half's register 0; trunc_exp computes in float
        if (geo || sigma_only) {  // NeRFNetwork.density
            if (valid) {
                if (half == 0) sigmas[slot] = tile_sigma_out(density_scale, sigma_logit);
                if (geo) {
                    float* __restrict__ g = geo + (size_t)slot * 15;
#pragma unroll
                    for (int r = 0; r < 8; r++) {
                        const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
                        if (row >= 1) g[row - 1] = g2[r];
                    }
                }
            }
            continue;
        }
        __builtin_amdgcn_sched_barrier(0);
        float e[3];
        tile_color_net_h(wl, wimg, half, g2, dx, dy, dz, e);
        if (valid && half == 0) {
            sigmas[slot] = tile_sigma_out(density_scale, sigma_logit);
#pragma unroll
            for (int o = 0; o < 3; o++) rgbs[slot * 3 + o] = tile_rgb_out_h(e[o]);
        }
    }
}


int pn_nerf_forward_launch(const pn_net* net, const float* xyzs, const float* dirs, const int* list, const int* ctl_count, uint32_t M_max,
                           float density_scale, float* sigmas, float* rgbs, int half, hipStream_t stream, uint32_t blocks_cap) {
    if (M_max == 0) return PN_OK;
    const uint32_t tiles = pn_div_up(M_max, 32);
    if (half) {
        PN_REQUIRE(net->emb_half);  // pn_net_enable_half first
        static const uint32_t max_blocks_h = pn_env_u32("PN_NERF_BLOCKS_H", 1024);  // 4 workgroups per CU
        uint32_t blocks = std::min(pn_div_up(tiles, PN_H_WAVES), max_blocks_h);
        if (blocks_cap) blocks = std::min(blocks, blocks_cap);
        k_nerf_forward_h<4, 4><<<blocks, PN_H_WAVES * 64, PN_NET_HALF_BYTES + 16 * sizeof(PnFusedLevel), stream>>>((const PnFusedLevel*)net->fused_levels, (const uint32_t*)net->emb_half,
                                                                                     (const uint4*)net->whalf, net->bound, xyzs, dirs, list, ctl_count,
                                                                                     M_max, density_scale, sigmas, rgbs, nullptr, net->n_entries * 4u, 0);
        PN_LAUNCH_CHECK();
        return PN_OK;
    }
    static const uint32_t max_blocks = pn_env_u32("PN_NERF_BLOCKS", 2048 / PN_BF_WAVES);  // 8 waves per CU x 256 CUs; waves stride over tiles
    uint32_t blocks = pn_div_up(tiles, PN_BF_WAVES);
    if (blocks > max_blocks) blocks = max_blocks;
    if (blocks_cap) blocks = std::min(blocks, blocks_cap);
    if (net->x_ok)
        k_nerf_forward<2, PN_BF_LU, true><<<blocks, PN_BF_WAVES * 64, PN_NET_X_BYTES + 16 * sizeof(PnFusedLevel), stream>>>(
            (const PnByteLevel*)net->byte_levels, net->embeddings, (const uint4*)net->wx, net->bound, xyzs, dirs, list, ctl_count, M_max, density_scale, sigmas, rgbs,
            nullptr, 0, net->x_scales);
    else
    k_nerf_forward<2, PN_BF_LU><<<blocks, PN_BF_WAVES * 64, PN_NET_SPLIT_BYTES + 16 * sizeof(PnFusedLevel), stream>>>((const PnByteLevel*)net->byte_levels, net->embeddings,
                                                                                  (const uint4*)net->wsplit, net->bound, xyzs, dirs, list, ctl_count,
                                                                                  M_max, density_scale, sigmas, rgbs, nullptr, 0);
    PN_LAUNCH_CHECK();
    return PN_OK;
}

static int density_launch(const pn_net* net, const float* xyzs, uint32_t M, float scale, float* sigmas, float* geo_feat, int half, hipStream_t st) {
    const uint32_t tiles = pn_div_up(M, 32);
    const int sigma_only = geo_feat == nullptr;
    // dirs is only read by the colour net, which this mode never reaches: any readable buffer of >= 3 M floats will do
    if (half) {
        PN_REQUIRE(net->emb_half);
        k_nerf_forward_h<4, 4><<<std::min(pn_div_up(tiles, PN_H_WAVES), 1024u), PN_H_WAVES * 64, PN_NET_HALF_BYTES + 16 * sizeof(PnFusedLevel), st>>>(
            (const PnFusedLevel*)net->fused_levels, (const uint32_t*)net->emb_half, (const uint4*)net->whalf, net->bound, xyzs, xyzs, nullptr, nullptr, M, scale,
            sigmas, nullptr, geo_feat, net->n_entries * 4u, sigma_only);
    } else if (net->x_ok) {
        k_nerf_forward<2, PN_BF_LU, true><<<std::min(pn_div_up(tiles, PN_BF_WAVES), 2048u / PN_BF_WAVES), PN_BF_WAVES * 64, PN_NET_X_BYTES + 16 * sizeof(PnFusedLevel), st>>>(
            (const PnByteLevel*)net->byte_levels, net->embeddings, (const uint4*)net->wx, net->bound, xyzs, xyzs, nullptr, nullptr, M, scale, sigmas, nullptr, geo_feat,
            sigma_only, net->x_scales);
    } else {
        k_nerf_forward<2, PN_BF_LU><<<std::min(pn_div_up(tiles, PN_BF_WAVES), 2048u / PN_BF_WAVES), PN_BF_WAVES * 64, PN_NET_SPLIT_BYTES + 16 * sizeof(PnFusedLevel), st>>>(
            (const PnByteLevel*)net->byte_levels, net->embeddings, (const uint4*)net->wsplit, net->bound, xyzs, xyzs, nullptr, nullptr, M, scale, sigmas,
            nullptr, geo_feat, sigma_only);
    }
    PN_LAUNCH_CHECK();
    return PN_OK;
}

extern "C" int pn_nerf_density(const pn_net* net, const float* xyzs, uint32_t M, float* sigmas, float* geo_feat, void* stream) {
    if (M == 0) return PN_OK;  // empty tensors have null data pointers
    PN_REQUIRE(net && xyzs && sigmas && geo_feat);
    return density_launch(net, xyzs, M, 1.0f, sigmas, geo_feat, 0, (hipStream_t)stream);
}

extern "C" int pn_nerf_density_half(const pn_net* net, const float* xyzs, uint32_t M, float* sigmas, float* geo_feat, void* stream) {
    if (M == 0) return PN_OK;
    PN_REQUIRE(net && xyzs && sigmas && geo_feat);
    return density_launch(net, xyzs, M, 1.0f, sigmas, geo_feat, 1, (hipStream_t)stream);
}

extern "C" int pn_nerf_sigma(const pn_net* net, const float* xyzs, uint32_t M, float density_scale, float* sigmas, int half, void* stream) {
    if (M == 0) return PN_OK;
    PN_REQUIRE(net && xyzs && sigmas);
    return density_launch(net, xyzs, M, density_scale, sigmas, nullptr, half, (hipStream_t)stream);
}

extern "C" int pn_nerf_forward(const pn_net* net, const float* xyzs, const float* dirs, uint32_t M, float density_scale, float* sigmas, float* rgbs,
                               void* stream) {
    if (M == 0) return PN_OK;  // empty tensors have null data pointers
    PN_REQUIRE(net && xyzs && dirs && sigmas && rgbs);
    return pn_nerf_forward_launch(net, xyzs, dirs, nullptr, nullptr, M, density_scale, sigmas, rgbs, 0, (hipStream_t)stream);
}

extern "C" int pn_nerf_forward_half(const pn_net* net, const float* xyzs, const float* dirs, uint32_t M, float density_scale, float* sigmas,
                                    float* rgbs, void* stream) {
    if (M == 0) return PN_OK;
    PN_REQUIRE(net && xyzs && dirs && sigmas && rgbs);
    return pn_nerf_forward_launch(net, xyzs, dirs, nullptr, nullptr, M, density_scale, sigmas, rgbs, 1, (hipStream_t)stream);
}

extern "C" int pn_host_float_to_half(const float* in_host, uint16_t* out_host, uint32_t n) {
    PN_REQUIRE(in_host && out_host);
    for (uint32_t i = 0; i < n; i++) out_host[i] = pn_f2h_bits(in_host[i]);
    return PN_OK;
}
