// The network of one 32-sample tile (hash-grid encoder + SH + the five dense layers of NeRFNetwork.forward, nerf/network.py:98-127), as device
// functions shared by the stand-alone network kernels (pn_nerf_forward.hip) and the fused trip kernel of the frame driver (pn_trips_fused.h).
// gfx950 only.  Moved here unchanged from pn_nerf_forward.hip (round 4); the wave layout is described at the top of that file.
#pragma once
#include <math.h>

#include "pn_common.h"
#include "pn_encoders.h"

// Everything in this header is network arithmetic and is compiled with FMA contraction, as pn_nerf_forward.hip has always been (-ffp-contract=fast):
// the pragma makes that a property of the code instead of the translation unit, so that the fused trip kernel — whose unit, pn_render_ops.hip, is
// built with -ffp-contract=off for the bit-exact ray march — runs the very same instructions (fused vs per-trip frames are compared bit for bit).
// A unit built without contraction defines PN_TU_FP_CONTRACT_OFF before including this header; the end of the header then switches back.
#pragma clang fp contract(fast)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Half(w * float(v)) as c10::Half computes it: the float product is rounded to float FIRST and then to half (two roundings).  Written
// naively, `(_Float16)(w * (float)v)` is fused by hipcc into v_fma_mixlo_f16, which rounds the exact product ONCE, straight to half — a
// different result whenever the float rounding lands on a half tie (tests/test_gpu_half.py caught it: isolated features one half ulp off the oracle).
// The empty asm keeps the float product a value of its own.
__device__ __forceinline__ _Float16 half_of_product(float w, _Float16 v) {
    float p = w * (float)v;
    asm volatile("" : "+v"(p));
    return (_Float16)p;
}
template <typename T>
__device__ __forceinline__ T rounded_product(float w, T v);
template <>
__device__ __forceinline__ float rounded_product<float>(float w, float v) { return w * v; }
template <>
__device__ __forceinline__ _Float16 rounded_product<_Float16>(float w, _Float16 v) { return half_of_product(w, v); }

// ------------------------------------------------------------------------------------------------ SH (degree <= 4)
// ------------------------------------------------------------------------------------------------ SH (degree <= 4)
__device__ __forceinline__ void sh16(float x, float y, float z, float* o) {
    const float xy = x * y, xz = x * z, yz = y * z, x2 = x * x, y2 = y * y, z2 = z * z;
    o[0] = SH_C0;
    o[1] = -SH_C1 * y;
    o[2] = SH_C1 * z;
    o[3] = -SH_C1 * x;
    o[4] = SH_C2A * xy;
    o[5] = -SH_C2A * yz;
    o[6] = SH_C2B * z2 - SH_C2C;
    o[7] = -SH_C2A * xz;
    o[8] = SH_C2D * x2 - SH_C2D * y2;
    o[9] = SH_C3A * y * (-3.0f * x2 + y2);
    o[10] = SH_C3B * xy * z;
    o[11] = SH_C3C * y * (1.0f - 5.0f * z2);
    o[12] = SH_C3D * z * (5.0f * z2 - 3.0f);
    o[13] = SH_C3C * x * (1.0f - 5.0f * z2);
    o[14] = SH_C3E * z * (x2 - y2);
    o[15] = SH_C3A * x * (-x2 + 3.0f * y2);
}

// 8 hash levels for one lane: feat[2j + c] = level (8h + j), channel c   (kernel_grid<float,3,2>, gridencoder.cu:87-197).  Every level is either
// fully dense (index = g0 + g1 s + g2 s^2, provably < table size) or hashed into a power-of-two table (index = (g0 ^ g1 P1 ^ g2 P2) & mask) —
// pn_net_create rejects anything else.
// Build knobs of the fp16 kernel's encoder (encode_levels_h below; the fp32 kernels' encoder has one form):
//   PN_ENC_LDS_LV   the per-level constants read per lane from an LDS copy instead of two scalar loads and a v_mov + v_mov + v_cndmask per constant
//   PN_ENC_UNIFIED  one branch-free index form for dense and hashed levels instead of the compiler's exec-mask flow per corner
//   PN_ENC_PK       the corner weights on the packed fp32 pipe (v_pk_mul_f32: the same roundings)
#ifndef PN_ENC_PK
#define PN_ENC_PK 1
#endif
#ifndef PN_ENC_LDS_LV
#define PN_ENC_LDS_LV 1
#endif
#ifndef PN_ENC_UNIFIED
#define PN_ENC_UNIFIED 1
#endif
#ifndef PN_SPLIT_PK
#define PN_SPLIT_PK 0
#endif
#ifndef PN_BF_LU
#define PN_BF_LU 4
#endif
// One level for one lane.  `L` = the lane's level constants in byte units (PnByteLevel, pn_common.h): the table entry of corner (i, c) — i: x / x + 1,
// c: the (y, z) pair — sits at byte ((t0[i] ^ X[c]) + S[c]) of the embeddings, ONE v_xad_u32 per corner and no 64-bit address arithmetic (the load takes
// the 32-bit offset beside the scalar base):
//   dense level   X = 0, S[c] = 8 (g1' s + g2' s^2 + offset), t0 = 8 g0'                          (index g0 + g1 s + g2 s^2, provably inside the level)
//   hashed level  S = 8 offset, X[c] = (8 g1' P1 ^ 8 g2' P2) & 8 mask, t0 = 8 g0' & 8 mask       (index (g0 ^ g1 P1 ^ g2 P2) & mask: masks and shifts
//                 distribute over xor, products wrap mod 2^32 either way)
// with both parts computed on every level (xmb = 0 switches the hash part off, m1d = m2d = 0 the dense part): the lane halves of a wave work on different
// levels.  The dense strides fit 24 bits (v_mad_u32_u24, full rate; v_mul_lo_u32 is quarter rate).  Same entries as kernel_grid (gridencoder.cu:87-197).
__device__ __forceinline__ void encode_level(const PnByteLevel& L, const float* __restrict__ emb, float u0, float u1, float u2, bool oob, float* out2) {
    const float scale = L.scale;
    float p0 = fmaf(u0, scale, 0.5f), p1 = fmaf(u1, scale, 0.5f), p2 = fmaf(u2, scale, 0.5f);
    const float f0 = floorf(p0), f1 = floorf(p1), f2 = floorf(p2);
    p0 -= f0; p1 -= f1; p2 -= f2;
    const uint32_t g1 = (uint32_t)f1, g2 = (uint32_t)f2, a0 = (uint32_t)f0 << 3;
    const uint32_t t0[2] = {a0 & L.mb, (a0 + 8u) & L.mb};
    uint32_t S[4], X[4];
    S[0] = __umul24(g2, L.m2d) + (__umul24(g1, L.m1d) + L.off_b);
    S[1] = S[0] + L.m1d;
    S[2] = S[0] + L.m2d;
    S[3] = S[1] + L.m2d;
    const uint32_t h1 = g1 * L.p1b, h2 = g2 * L.p2b, h1n = h1 + L.p1b, h2n = h2 + L.p2b;
    X[0] = (h1 ^ h2) & L.xmb;
    X[1] = (h1n ^ h2) & L.xmb;
    X[2] = (h1 ^ h2n) & L.xmb;
    X[3] = (h1n ^ h2n) & L.xmb;
    f32x2 v[8];
#pragma unroll
    for (int idx = 0; idx < 8; idx++) {
        const uint32_t boff = (t0[idx & 1] ^ X[idx >> 1]) + S[idx >> 1];
        const float2 e = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(emb) + boff);
        v[idx] = f32x2{e.x, e.y};
    }
    // the eight corner weights ((1 * tx) * ty) * tz and the two channels' running sums, corner after corner as kernel_grid does them, on the
    // packed fp32 pipe: (w_even, w_odd) pairs from v_pk_mul_f32, both channels of a corner in one v_pk_fma_f32 (the same roundings)
    const f32x2 q0 = {1 - p0, p0};
    const float n1 = 1 - p1, n2 = 1 - p2;
    const f32x2 qa = q0 * n1, qb = q0 * p1;
    const f32x2 w2[4] = {qa * n2, qb * n2, qa * p2, qb * p2};
    f32x2 r = {0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 4; c++) {
        r = __builtin_elementwise_fma(f32x2{w2[c].x, w2[c].x}, v[2 * c], r);
        r = __builtin_elementwise_fma(f32x2{w2[c].y, w2[c].y}, v[2 * c + 1], r);
    }
    out2[0] = oob ? 0.f : r.x;
    out2[1] = oob ? 0.f : r.y;
}

// `lds_lv`: this lane half's 8 levels in LDS (lane half h: levels 8h .. 8h + 7; read per lane — as scalar loads of both halves' records every constant
// would cost a v_mov + v_mov + v_cndmask).  Fully unrolled in groups of LU levels (a partly unrolled loop writes feat[] through s_set_gpr_idx, four
// instructions per value); the scheduling barrier between the groups bounds the gathers in flight per lane at 8 LU.
template <int LU>
__device__ __forceinline__ void encode8(const PnByteLevel* lds_lv, const float* __restrict__ emb, float u0, float u1, float u2, bool oob, float* feat) {
#pragma unroll
    for (int g = 0; g < 8; g += LU) {
#pragma unroll
        for (int j = g; j < g + LU; j++) {
            const PnByteLevel L = lds_lv[j];
            encode_level(L, emb, u0, u1, u2, oob, feat + 2 * j);
        }
        if (g + LU < 8) __builtin_amdgcn_sched_barrier(0);
    }
}

__device__ __forceinline__ f32x16 relu16(f32x16 v) {
#pragma unroll
    for (int r = 0; r < 16; r++) v[r] = __int_as_float(max(__float_as_int(v[r]), 0));  // ReLU as ONE v_max_i32 (fmaxf: two v_max_f32, NaN canonicalisation)
    return v;
}

// ------------------------------------------------------------------------------------------------ fused kernel
// The dense layers run on the bf16 matrix pipe at fp32 accuracy.  An fp32 value is cut, by truncation, into three bf16 pieces
// x = hi + mid + lo (8 + 8 + 8 significant bits, exact), weights likewise on the host; a product x*w is the six partial
// products whose weight is >= 2^-16 (hi*hi, hi*mid, mid*hi, mid*mid, hi*lo, lo*hi; the dropped mid*lo, lo*mid, lo*lo are
// <= 2^-23 |x*w|, the size of fp32's own product rounding), accumulated in the fp32 accumulator of
// v_mfma_f32_32x32x16_bf16.  Six bf16 MFMAs of 32 cycles do the work of sixteen v_mfma_f32_32x32x2_f32 of 64 cycles, and — unlike
// the f32-input MFMA, which occupies the fp32 vector ALUs — they run beside the VALU work of the SIMD's other waves.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
struct Split8 { uint4 hi, mid, lo; };

__device__ __forceinline__ uint32_t hi_pair(float a, float b) {  // bf16 (truncated) of a in the low half, of b in the high half
    return __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u);
}
// x - (bf16 piece of x), two values per v_pk_add_f32
__device__ __forceinline__ f32x2 drop_hi(f32x2 x) {
    const f32x2 h = __builtin_bit_cast(f32x2, __builtin_bit_cast(u32x2, x) & 0xffff0000u);
#if PN_SPLIT_PK
    f32x2 r;  // written out: the vector combiner turns half of these subtractions back into scalar v_add_f32
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(x), "v"(h));
    return r;
#else
    return f32x2{x.x - h.x, x.y - h.y};
#endif
}
__device__ __forceinline__ Split8 split8(float x0, float x1, float x2, float x3, float x4, float x5, float x6, float x7) {
    Split8 o;
    f32x2 a = {x0, x1}, b = {x2, x3}, c = {x4, x5}, d = {x6, x7};
    o.hi = make_uint4(hi_pair(a.x, a.y), hi_pair(b.x, b.y), hi_pair(c.x, c.y), hi_pair(d.x, d.y));
    a = drop_hi(a); b = drop_hi(b); c = drop_hi(c); d = drop_hi(d);
    o.mid = make_uint4(hi_pair(a.x, a.y), hi_pair(b.x, b.y), hi_pair(c.x, c.y), hi_pair(d.x, d.y));
    a = drop_hi(a); b = drop_hi(b); c = drop_hi(c); d = drop_hi(d);
    o.lo = make_uint4(hi_pair(a.x, a.y), hi_pair(b.x, b.y), hi_pair(c.x, c.y), hi_pair(d.x, d.y));
    return o;
}
#define PN_BMFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, (a)), __builtin_bit_cast(bf16x8, (b)), (c), 0, 0, 0)

// acc += W(group G) · x for one K chunk: the six partial products.  The first MFMA of a chain (acc = literal 0) gets a freshly
// allocated destination, and hipcc (ROCm 7.2) does not treat the destination of v_mfma_f32_32x32x16_bf16 as early-clobber: an A or
// B operand that dies in that instruction may be given the same registers (seen in the ISA of an earlier ordering).  Leading with
// hi*hi, both of whose operands are used again below, keeps every source of a first MFMA live and therefore disjoint from its
// destination; tests/test_host.py scans the shipped ISA for such overlaps.  (A precaution, not a fix: round 1 saw run-to-run corruption in an
// earlier build of this kernel and blamed first this, then packed-fp32 VALU of co-resident waves; neither reproduced in isolation
// (tools/repro_mfma_overlap.hip: 32.7 M overlapping MFMAs, tools/repro_pk_mfma.hip: 819 M MFMAs beside v_pk_* streams, 0 differences) and
// both claims are withdrawn)
__device__ __forceinline__ f32x16 split_mac(const uint4* __restrict__ wl, int G, const Split8& x, f32x16 acc) {
    const uint4 wh = wl[(G * 3 + 0) * 64], wm = wl[(G * 3 + 1) * 64], wo = wl[(G * 3 + 2) * 64];
    acc = PN_BMFMA(wh, x.hi, acc);
    acc = PN_BMFMA(wh, x.mid, acc);
    acc = PN_BMFMA(wm, x.hi, acc);
    acc = PN_BMFMA(wm, x.mid, acc);
    acc = PN_BMFMA(wh, x.lo, acc);
    acc = PN_BMFMA(wo, x.hi, acc);
    return acc;
}
__device__ __forceinline__ Split8 split8_of(const f32x16& v, int r0) {
    return split8(v[r0], v[r0 + 1], v[r0 + 2], v[r0 + 3], v[r0 + 4], v[r0 + 5], v[r0 + 6], v[r0 + 7]);
}

// ---- one tile = 32 samples of one wave, two lanes per sample (lane half h: hash levels [8h, 8h + 8), out-rows (r&3) + 8 (r>>2) + 4h of the D
// layout).  The two halves of NeRFNetwork.forward as functions of the lane's sample, so that the stand-alone network kernel (pn_nerf_forward.hip)
// and the fused trip kernel of the frame driver (pn_trips_fused.h, a translation unit built with -ffp-contract=off) run the same instructions:
// the pragma gives these bodies the contraction the network kernels have always been built with.
// `wl` = the workgroup's LDS weight image + lane, `lds_lv` = this lane half's 8 level records in LDS.
template <int LU>
__device__ __forceinline__ f32x16 tile_sigma_net(const PnByteLevel* lds_lv, const float* __restrict__ emb,
                                                 const uint4* __restrict__ wl, int half, float bound, float inv2b, float x, float y, float z) {
#pragma clang fp contract(fast)
    // GridEncoder.forward: inputs = (x + bound) / (2 * bound)  (gridencoder/grid.py:149) — a tensor divided by a Python scalar, which torch's CUDA
    // kernel computes as a multiplication by the scalar's fp32 reciprocal (ATen BinaryDivTrueKernel.cu, the CPU-scalar case); for the power-of-two
    // bounds of every option set of the reference the two are the same number.  inv2b = 1.0f / (2 * bound).
    const float u0 = (x + bound) * inv2b, u1 = (y + bound) * inv2b, u2 = (z + bound) * inv2b;
    const bool oob = (u0 < 0 || u0 > 1 || u1 < 0 || u1 > 1 || u2 < 0 || u2 > 1);
    float feat[16];
    encode8<LU>(lds_lv, emb, oob ? 0.f : u0, oob ? 0.f : u1, oob ? 0.f : u2, oob, feat);
    __builtin_amdgcn_sched_barrier(0);  // keep the next layer's LDS weight reads from being hoisted (register pressure)
    // ---- sigma net layer 0: 32 -> 64, ReLU   (groups 0..3 = tile*2 + chunk)
    f32x16 a0 = {0}, a1 = {0};
#pragma unroll
    for (int kc = 0; kc < 2; kc++) {
        const Split8 b = split8(feat[8 * kc], feat[8 * kc + 1], feat[8 * kc + 2], feat[8 * kc + 3], feat[8 * kc + 4], feat[8 * kc + 5],
                                feat[8 * kc + 6], feat[8 * kc + 7]);
        a0 = split_mac(wl, 0 + kc, b, a0);
        a1 = split_mac(wl, 2 + kc, b, a1);
    }
    a0 = relu16(a0);
    a1 = relu16(a1);
    __builtin_amdgcn_sched_barrier(0);
    // ---- sigma net layer 1: 64 -> 16   (groups 4..7)
    f32x16 h2 = {0};
#pragma unroll
    for (int kc = 0; kc < 4; kc++) h2 = split_mac(wl, 4 + kc, split8_of(kc < 2 ? a0 : a1, (kc & 1) * 8), h2);
    return h2;  // row 0 (the sigma logit) lives in the low half's register 0
}

// colour net on [SH16(dir) | geo15]: e[0..2] = the three logits of the last Linear (valid on both lanes of the pair).  `wimg` = the LDS weight image.
__device__ __forceinline__ void tile_color_net(const uint4* __restrict__ wl, const uint4* wimg, int half, const f32x16& h2, float dx, float dy, float dz,
                                               float (&e)[3]) {
#pragma clang fp contract(fast)
    // ---- colour net input: 16 values per lane (see PN_MAPL / PN_MAPU)
    float sh[16];
    sh16(dx, dy, dz, sh);
    float v[16];
    // `half ? arr[i] : arr[j]` is rewritten by the compiler into arr[half ? i : j], a dynamic register index that it then lowers
    // to a 16-way compare + v_cndmask chain (~200 VALU instructions per tile); the empty asm pins both operands in registers
    // so that each select stays one v_cndmask
    auto pick = [half](float a, float b) {
        asm volatile("" : "+v"(a), "+v"(b));
        return half ? a : b;
    };
#pragma unroll
    for (int k = 0; k < 7; k++) v[k] = pick(h2[k], h2[k + 1]);
    v[7] = pick(h2[7], sh[0]);
#pragma unroll
    for (int k = 8; k < 15; k++) v[k] = pick(sh[k + 1], sh[k - 7]);
    v[15] = pick(0.0f, sh[8]);
    __builtin_amdgcn_sched_barrier(0);
    // ---- colour layer 0: 31 -> 64, ReLU   (groups 8..11)
    f32x16 c0 = {0}, c1 = {0};
#pragma unroll
    for (int kc = 0; kc < 2; kc++) {
        const Split8 b = split8(v[8 * kc], v[8 * kc + 1], v[8 * kc + 2], v[8 * kc + 3], v[8 * kc + 4], v[8 * kc + 5], v[8 * kc + 6], v[8 * kc + 7]);
        c0 = split_mac(wl, 8 + kc, b, c0);
        c1 = split_mac(wl, 10 + kc, b, c1);
    }
    c0 = relu16(c0);
    c1 = relu16(c1);
    __builtin_amdgcn_sched_barrier(0);
    // ---- colour layer 1: 64 -> 64, ReLU   (groups 12..19 = tile*4 + chunk)
    f32x16 d0 = {0}, d1 = {0};
#pragma unroll
    for (int kc = 0; kc < 4; kc++) {
        const Split8 b = split8_of(kc < 2 ? c0 : c1, (kc & 1) * 8);
        d0 = split_mac(wl, 12 + kc, b, d0);
        d1 = split_mac(wl, 16 + kc, b, d1);
    }
    d0 = relu16(d0);
    d1 = relu16(d1);
    __builtin_amdgcn_sched_barrier(0);
    // ---- colour layer 2: 64 -> 3 on the vector ALU: this lane holds 32 of the 64 hidden values of its sample (D layout), its
    // partner lane (l ^ 32) the other 32; 3 x 32 FMAs with broadcast LDS weights, then one cross-half add per output
    e[0] = e[1] = e[2] = 0.f;
    {
        const float* __restrict__ wlast = reinterpret_cast<const float*>(wimg) + PN_NET_SPLIT_W_BYTES / 4 + half * 96;
#pragma unroll
        for (int q4 = 0; q4 < 8; q4++) {
            const float4 wa = *reinterpret_cast<const float4*>(wlast + q4 * 12);
            const float4 wb = *reinterpret_cast<const float4*>(wlast + q4 * 12 + 4);
            const float4 wc = *reinterpret_cast<const float4*>(wlast + q4 * 12 + 8);
            const f32x16& src = (q4 < 4) ? d0 : d1;
            const int r = (q4 & 3) * 4;
            e[0] = fmaf(wa.x, src[r], e[0]); e[1] = fmaf(wa.y, src[r], e[1]); e[2] = fmaf(wa.z, src[r], e[2]);
            e[0] = fmaf(wa.w, src[r + 1], e[0]); e[1] = fmaf(wb.x, src[r + 1], e[1]); e[2] = fmaf(wb.y, src[r + 1], e[2]);
            e[0] = fmaf(wb.z, src[r + 2], e[0]); e[1] = fmaf(wb.w, src[r + 2], e[1]); e[2] = fmaf(wc.x, src[r + 2], e[2]);
            e[0] = fmaf(wc.y, src[r + 3], e[0]); e[1] = fmaf(wc.z, src[r + 3], e[1]); e[2] = fmaf(wc.w, src[r + 3], e[2]);
        }
#pragma unroll
        for (int o = 0; o < 3; o++) e[o] += __shfl_xor(e[o], 32);
    }
}
// ------------------------------------------------------------------------------------------------ fp16 hi/lo form of the same two functions (pn_common.h: pn_net::wx)
// x = hi + lo with hi = f16(x), lo = f16(x - hi) (v_cvt_pk_f16_f32: round to nearest even, two values per instruction); x * w = hi*hi + hi*lo + lo*hi on
// v_mfma_f32_32x32x16_f16 (the dropped lo*lo is <= 2^-22 |x w|).  `c`: a power-of-two factor applied before the split (exact).
typedef _Float16 pn_h16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 pn_h16x8 __attribute__((ext_vector_type(8)));
struct Split8x { uint4 hi, lo; };
// (Written out as v_cvt_pk_f16_f32 / v_fma_mix_f32 inline assembly — four instructions per pair instead of the ~nine the compiler makes of this, 1 319 instead of
// 1 439 vector instructions per tile — the stand-alone kernel kept its accuracy but frames of the fused launch differed from the stand-alone kernel's by more
// than 1e-5: the hazard recogniser does not look into inline assembly next to the matrix instructions.  Left to the compiler.)
__device__ __forceinline__ void split_pair_x(float a, float b, uint32_t& hi, uint32_t& lo) {
    float m1 = -1.0f;
    asm("" : "+s"(m1));   // opaque: fma(h, -1, a) with a literal -1 is rewritten as a - h, which is selected as v_cvt_f32_f16 + v_sub_f32 + a conversion per value
    const pn_h16x2 h0 = {(_Float16)a, (_Float16)b};
    hi = __builtin_bit_cast(uint32_t, h0);
    asm("" : "+v"(hi));   // opaque: otherwise each half is converted a second time on its own instead of being read out of the packed register
    const pn_h16x2 h = __builtin_bit_cast(pn_h16x2, hi);
    const pn_h16x2 l = {(_Float16)__builtin_fmaf((float)h.x, m1, a), (_Float16)__builtin_fmaf((float)h.y, m1, b)};
    lo = __builtin_bit_cast(uint32_t, l);
}
__device__ __forceinline__ Split8x split8x(float x0, float x1, float x2, float x3, float x4, float x5, float x6, float x7, float c) {
    Split8x o;
    split_pair_x(x0 * c, x1 * c, o.hi.x, o.lo.x);
    split_pair_x(x2 * c, x3 * c, o.hi.y, o.lo.y);
    split_pair_x(x4 * c, x5 * c, o.hi.z, o.lo.z);
    split_pair_x(x6 * c, x7 * c, o.hi.w, o.lo.w);
    return o;
}
__device__ __forceinline__ Split8x split8x_of(const f32x16& v, int r0, float c) {
    return split8x(v[r0], v[r0 + 1], v[r0 + 2], v[r0 + 3], v[r0 + 4], v[r0 + 5], v[r0 + 6], v[r0 + 7], c);
}
#define PN_XMFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(pn_h16x8, (a)), __builtin_bit_cast(pn_h16x8, (b)), (c), 0, 0, 0)
// acc += W(group G) . x for one K chunk (hi*hi first: see split_mac)
__device__ __forceinline__ f32x16 split_mac_x(const uint4* __restrict__ wl, int G, const Split8x& x, f32x16 acc) {
    const uint4 wh = wl[(G * 2 + 0) * 64], wo = wl[(G * 2 + 1) * 64];
    acc = PN_XMFMA(wh, x.hi, acc);
    acc = PN_XMFMA(wh, x.lo, acc);
    acc = PN_XMFMA(wo, x.hi, acc);
    return acc;
}
// sf: the features' power-of-two scale xs[0]; every later layer's scale sits in the weight image (pn_nerf_forward.hip: net_choose_form), so the result is
// xs[2] * (sigma logit, geometry features): the colour net below takes it as it is, whoever wants the values multiplies by pn_net::x_rscale = 1 / xs[2].
template <int LU>
__device__ __forceinline__ f32x16 tile_sigma_net_x(const PnByteLevel* lds_lv, const float* __restrict__ emb,
                                                   const uint4* __restrict__ wl, int half, float bound, float inv2b, float x, float y, float z, float sf) {
#pragma clang fp contract(fast)
    const float u0 = (x + bound) * inv2b, u1 = (y + bound) * inv2b, u2 = (z + bound) * inv2b;   // see tile_sigma_net
    const bool oob = (u0 < 0 || u0 > 1 || u1 < 0 || u1 > 1 || u2 < 0 || u2 > 1);
    float feat[16];
    encode8<LU>(lds_lv, emb, oob ? 0.f : u0, oob ? 0.f : u1, oob ? 0.f : u2, oob, feat);
    __builtin_amdgcn_sched_barrier(0);
    f32x16 a0 = {0}, a1 = {0};
#pragma unroll
    for (int kc = 0; kc < 2; kc++) {
        const Split8x b = split8x(feat[8 * kc], feat[8 * kc + 1], feat[8 * kc + 2], feat[8 * kc + 3], feat[8 * kc + 4], feat[8 * kc + 5],
                                  feat[8 * kc + 6], feat[8 * kc + 7], sf);
        a0 = split_mac_x(wl, 0 + kc, b, a0);
        a1 = split_mac_x(wl, 2 + kc, b, a1);
    }
    a0 = relu16(a0);
    a1 = relu16(a1);
    __builtin_amdgcn_sched_barrier(0);
    f32x16 h2 = {0};
#pragma unroll
    for (int kc = 0; kc < 4; kc++) h2 = split_mac_x(wl, 4 + kc, split8x_of(kc < 2 ? a0 : a1, (kc & 1) * 8, 1.0f), h2);
    return h2;
}
__device__ __forceinline__ void tile_color_net_x(const uint4* __restrict__ wl, const uint4* wimg, int half, const f32x16& h2, float dx, float dy, float dz,
                                                 float (&e)[3]) {
#pragma clang fp contract(fast)
    float sh[16];
    sh16(dx, dy, dz, sh);
    float v[16];
    auto pick = [half](float a, float b) {
        asm volatile("" : "+v"(a), "+v"(b));
        return half ? a : b;
    };
#pragma unroll
    for (int k = 0; k < 7; k++) v[k] = pick(h2[k], h2[k + 1]);
    v[7] = pick(h2[7], sh[0]);
#pragma unroll
    for (int k = 8; k < 15; k++) v[k] = pick(sh[k + 1], sh[k - 7]);
    v[15] = pick(0.0f, sh[8]);
    __builtin_amdgcn_sched_barrier(0);
    f32x16 c0 = {0}, c1 = {0};
#pragma unroll
    for (int kc = 0; kc < 2; kc++) {
        const Split8x b = split8x(v[8 * kc], v[8 * kc + 1], v[8 * kc + 2], v[8 * kc + 3], v[8 * kc + 4], v[8 * kc + 5], v[8 * kc + 6], v[8 * kc + 7], 1.0f);
        c0 = split_mac_x(wl, 8 + kc, b, c0);
        c1 = split_mac_x(wl, 10 + kc, b, c1);
    }
    c0 = relu16(c0);
    c1 = relu16(c1);
    __builtin_amdgcn_sched_barrier(0);
    f32x16 d0 = {0}, d1 = {0};
#pragma unroll
    for (int kc = 0; kc < 4; kc++) {
        const Split8x b = split8x_of(kc < 2 ? c0 : c1, (kc & 1) * 8, 1.0f);
        d0 = split_mac_x(wl, 12 + kc, b, d0);
        d1 = split_mac_x(wl, 16 + kc, b, d1);
    }
    d0 = relu16(d0);
    d1 = relu16(d1);
    __builtin_amdgcn_sched_barrier(0);
    e[0] = e[1] = e[2] = 0.f;
    {
        const float* __restrict__ wlast = reinterpret_cast<const float*>(wimg) + PN_NET_X_W_BYTES / 4 + half * 96;
#pragma unroll
        for (int q4 = 0; q4 < 8; q4++) {
            const float4 wa = *reinterpret_cast<const float4*>(wlast + q4 * 12);
            const float4 wb = *reinterpret_cast<const float4*>(wlast + q4 * 12 + 4);
            const float4 wc = *reinterpret_cast<const float4*>(wlast + q4 * 12 + 8);
            const f32x16& src = (q4 < 4) ? d0 : d1;
            const int r = (q4 & 3) * 4;
            e[0] = fmaf(wa.x, src[r], e[0]); e[1] = fmaf(wa.y, src[r], e[1]); e[2] = fmaf(wa.z, src[r], e[2]);
            e[0] = fmaf(wa.w, src[r + 1], e[0]); e[1] = fmaf(wb.x, src[r + 1], e[1]); e[2] = fmaf(wb.y, src[r + 1], e[2]);
            e[0] = fmaf(wb.z, src[r + 2], e[0]); e[1] = fmaf(wb.w, src[r + 2], e[1]); e[2] = fmaf(wc.x, src[r + 2], e[2]);
            e[0] = fmaf(wc.y, src[r + 3], e[0]); e[1] = fmaf(wc.z, src[r + 3], e[1]); e[2] = fmaf(wc.w, src[r + 3], e[2]);
        }
#pragma unroll
        for (int o = 0; o < 3; o++) e[o] += __shfl_xor(e[o], 32);
    }
}

// exp(x) for the network's two activations, written out: the device library's expf is inlined bitcode whose multiply-adds contract or not with the
// flags of the translation unit it lands in, and the same sample must give the same bits in the stand-alone network kernels (-ffp-contract=fast) and in
// the fused trip kernel (a -ffp-contract=off unit).  exp(x) = 2^e * 2^a with x log2(e) = e + a evaluated with a two-piece log2(e) (the product's
// rounding error is carried in `pl`), |a| <= 1/2 on v_exp_f32 (1 ulp), the scaling by v_ldexp_f32 (exact; overflow -> inf, underflow -> 0).
__device__ __forceinline__ float pn_expf(float x) {
    x = fminf(fmaxf(x, -110.0f), 90.0f);  // beyond: 0 / inf through the ldexp
    const float L = 1.44269502162933349609375f, Ll = 1.92596299112661746e-8f;  // log2(e) = L + Ll
    float ph = x * L;
    asm volatile("" : "+v"(ph));  // a value of its own: the product must not be contracted into the subtractions below
    const float pl = __builtin_fmaf(x, Ll, __builtin_fmaf(x, L, -ph));
    const float e = rintf(ph);
    const float a = (ph - e) + pl;
    return ldexpf(__builtin_amdgcn_exp2f(a), (int)e);
}
// what the network writes for a sample (nerf/network.py:98-127): sigma = density_scale * trunc_exp(logit), rgb = sigmoid(logits)
__device__ __forceinline__ float tile_sigma_out(float density_scale, float sigma_logit) {
    return density_scale * pn_expf(sigma_logit);        // trunc_exp forward = exp (nerf/activation.py:8-10)
}
__device__ __forceinline__ float tile_rgb_out(float logit) {
    return 1.0f / (1.0f + pn_expf(-logit));            // torch.sigmoid
}

// ------------------------------------------------------------------------------------------------ fp16 form of the fused kernel
// NeRFNetwork.forward as the reference runs it under torch.cuda.amp.autocast (trainer.py:561 with fp16=True; BASELINE configs[4]):
//   * gridencoder/grid.py:43-44: embeddings.to(torch.half); kernel_grid<at::Half,3,2> (gridencoder.cu:87-197) keeps positions and weights
//     in float and accumulates `results[ch] += w * grid[index + ch]` in at::Half: the float product is rounded to half, the running sum is
//     a half + half addition (c10::Half operators) — restated literally by encode8_h;
//   * nn.Linear under autocast: half inputs x half weights, fp32 accumulation, half output -> v_mfma_f32_32x32x16_f16 and ONE rounding of
//     the accumulator to fp16 per output (hi-precision accumulate order is the matrix core's, cuBLAS's in the reference: unpinnable, hence a
//     tolerance of a few half ulps in the tests);
//   * trunc_exp casts its half input to float (activation.py:7); SH stays float (sphere_harmonics.py:16) and is rounded to half when the
//     concatenated colour-net input enters the first colour Linear; torch.sigmoid of a half tensor rounds its float result to half.
// Same wave layout as k_nerf_forward (32 samples per wave, two lanes per sample, D layout of one layer = B layout of the next), no split:
// 20 MFMAs per tile instead of 120, a 21 KB LDS weight image instead of 61 KB, 4-byte corner gathers instead of 8-byte ones.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
#define PN_HMFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)

// levels J0 .. J0+NJ-1 of this lane's 8 (fully unrolled: every feat index is a compile-time constant, so feat stays in registers).
// The table is read through a buffer resource (uniform base in SGPRs + one 32-bit byte offset per lane): a corner costs ONE address VGPR
// and no 64-bit pointer arithmetic — with flat 64-bit addresses the 32 dword gathers in flight needed 64 address registers and the
// allocator serialised them behind spills.
typedef int pn_rsrc_t __attribute__((ext_vector_type(4)));
template <int J0, int NJ>
__device__ __forceinline__ void encode_levels_h(const PnFusedLevel* __restrict__ lv, const PnFusedLevel* lds_lv, __amdgpu_buffer_rsrc_t emb_rsrc, int half,
                                                float u0, float u1, float u2, bool oob, _Float16* feat) {
#pragma unroll
    for (int j = J0; j < J0 + NJ; j++) {
#if PN_ENC_LDS_LV
        const PnFusedLevel L = lds_lv[j];  // this lane half's level j (see encode8)
#else
        const PnFusedLevel A = lv[j], B = lv[j + 8];  // wave-uniform
        PnFusedLevel L;
        L.scale = half ? B.scale : A.scale;
        L.offset = half ? B.offset : A.offset;
        L.m1 = half ? B.m1 : A.m1; L.m2 = half ? B.m2 : A.m2; L.mask = half ? B.mask : A.mask;
        L.dense = half ? B.dense : A.dense;
        L.dm = half ? B.dm : A.dm; L.xm = half ? B.xm : A.xm;
#endif
        const float scale = L.scale;
        const uint32_t m1 = L.m1, m2 = L.m2, mask = L.mask;
        const bool dense = L.dense != 0;
        const uint32_t base = L.offset;  // entries before this level
        float p0 = fmaf(u0, scale, 0.5f), p1 = fmaf(u1, scale, 0.5f), p2 = fmaf(u2, scale, 0.5f);
        const float f0 = floorf(p0), f1 = floorf(p1), f2 = floorf(p2);
        p0 -= f0; p1 -= f1; p2 -= f2;
        const uint32_t t0[2] = {(uint32_t)f0, (uint32_t)f0 + 1u};
        const uint32_t t1a = (uint32_t)f1 * m1, t2a = (uint32_t)f2 * m2;
        const uint32_t t1[2] = {t1a, t1a + m1}, t2[2] = {t2a, t2a + m2};
        uint32_t v[8];
#if PN_ENC_UNIFIED
        const uint32_t dm = L.dm, M = L.xm;  // see encode_level
        uint32_t S[4], X[4];
#pragma unroll
        for (int c = 0; c < 4; c++) {
            S[c] = (t1[c & 1] + t2[c >> 1]) & dm;
            X[c] = (t1[c & 1] ^ t2[c >> 1]) & ~dm;
        }
#pragma unroll
        for (int idx = 0; idx < 8; idx++) {
            const uint32_t index = ((t0[idx & 1] + S[idx >> 1]) ^ X[idx >> 1]) & M;
            v[idx] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(emb_rsrc, (int)((base + index) << 2), 0, 0);
        }
#else
#pragma unroll
        for (int idx = 0; idx < 8; idx++) {
            const uint32_t a0 = t0[idx & 1], a1 = t1[(idx >> 1) & 1], a2 = t2[(idx >> 2) & 1];
            const uint32_t index = dense ? (a0 + a1 + a2) : ((a0 ^ a1 ^ a2) & mask);
            v[idx] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(emb_rsrc, (int)((base + index) << 2), 0, 0);
        }
#endif
        _Float16 r0 = (_Float16)0.0f, r1 = (_Float16)0.0f;
#if PN_ENC_PK
        const f32x2 q0 = {1 - p0, p0};
        const float n1 = 1 - p1, n2 = 1 - p2;
        const f32x2 qa = q0 * n1, qb = q0 * p1;
        const f32x2 w2[4] = {qa * n2, qb * n2, qa * p2, qb * p2};  // ((1 * tx) * ty) * tz, two corners per v_pk_mul_f32
#endif
#pragma unroll
        for (int idx = 0; idx < 8; idx++) {
#if PN_ENC_PK
            const float w = (idx & 1) ? w2[idx >> 1].y : w2[idx >> 1].x;
#else
            float w = 1;
            w *= (idx & 1) ? p0 : 1 - p0;
            w *= (idx & 2) ? p1 : 1 - p1;
            w *= (idx & 4) ? p2 : 1 - p2;
#endif
            const f16x2 e = __builtin_bit_cast(f16x2, v[idx]);
            r0 = r0 + half_of_product(w, e[0]);  // Half(float * Half) then Half + Half, gridencoder.cu:184
            r1 = r1 + half_of_product(w, e[1]);
        }
        feat[2 * j] = oob ? (_Float16)0.0f : r0;
        feat[2 * j + 1] = oob ? (_Float16)0.0f : r1;
    }
}
// 8 hash levels of one lane, LU levels' gathers (8 x LU dword loads) in flight at a time
template <int LU>
__device__ __forceinline__ void encode8_h(const PnFusedLevel* __restrict__ lv, const PnFusedLevel* lds_lv, __amdgpu_buffer_rsrc_t emb_h, int half, float u0, float u1,
                                          float u2, bool oob, _Float16* feat) {
    static_assert(LU == 2 || LU == 4 || LU == 8, "LU");
    if (LU == 8) { encode_levels_h<0, 8>(lv, lds_lv, emb_h, half, u0, u1, u2, oob, feat); return; }
    if (LU == 4) {
        encode_levels_h<0, 4>(lv, lds_lv, emb_h, half, u0, u1, u2, oob, feat);
        __builtin_amdgcn_sched_barrier(0);
        encode_levels_h<4, 4>(lv, lds_lv, emb_h, half, u0, u1, u2, oob, feat);
        return;
    }
    encode_levels_h<0, 2>(lv, lds_lv, emb_h, half, u0, u1, u2, oob, feat);
    __builtin_amdgcn_sched_barrier(0);
    encode_levels_h<2, 2>(lv, lds_lv, emb_h, half, u0, u1, u2, oob, feat);
    __builtin_amdgcn_sched_barrier(0);
    encode_levels_h<4, 2>(lv, lds_lv, emb_h, half, u0, u1, u2, oob, feat);
    __builtin_amdgcn_sched_barrier(0);
    encode_levels_h<6, 2>(lv, lds_lv, emb_h, half, u0, u1, u2, oob, feat);
}

// fp32 accumulators of one layer -> that layer's half output, optionally through ReLU, as the next layer's B operands
__device__ __forceinline__ f16x8 to_half8(const f32x16& v, int r0, bool relu) {
    f16x8 o;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        _Float16 h = (_Float16)v[r0 + i];
        if (relu) h = h > (_Float16)0.0f ? h : (_Float16)0.0f;
        o[i] = h;
    }
    return o;
}

// Accumulator chains start from a zero held in REGISTERS (opaque to the optimiser), not from the literal 0: with a literal C operand
// hipcc (ROCm 7.2) gives the first MFMA of a chain a destination that overlaps its dying A / B operands (seen in this kernel's ISA:
// v_mfma_f32_32x32x16_f16 v[0:15], v[0:3], v[4:7], 0); with C in registers the destination is tied to C, which is live together with
// A and B.  Same precaution as split_mac's operand order above; tests/test_host.py scans the shipped ISA for such overlaps, and
// tools/repro_mfma_overlap.hip measures whether the overlap is actually harmful on gfx950 (16 v_mov per chain is the price).
__device__ __forceinline__ f32x16 zero16() {
    float z = 0.0f;
    asm volatile("" : "+v"(z));
    f32x16 v;
#pragma unroll
    for (int i = 0; i < 16; i++) v[i] = z;
    return v;
}

// The fp16 (autocast) tile: g2[r] = this lane's 8 of the sigma net's 16 outputs, rounded to half (rows (r&3) + 8 (r>>2) + 4 half); g2[0] of the low
// half is the sigma logit.  `wimg` = the LDS image (PN_NET_HALF_BYTES of weights, then the 16 level records).
template <int LU>
__device__ __forceinline__ void tile_sigma_net_h(const PnFusedLevel* __restrict__ lv, const uint4* wimg, __amdgpu_buffer_rsrc_t emb_rsrc,
                                                 const uint4* __restrict__ wl, int half, float bound, float x, float y, float z, float (&g2)[8]) {
#pragma clang fp contract(fast)
    auto W = [&](int G) { return __builtin_bit_cast(f16x8, wl[G * 64]); };
    const float u0 = (x + bound) / (2 * bound), u1 = (y + bound) / (2 * bound), u2 = (z + bound) / (2 * bound);
    const bool oob = (u0 < 0 || u0 > 1 || u1 < 0 || u1 > 1 || u2 < 0 || u2 > 1);
    _Float16 feat[16];
    // the per-level constants are selected per lane half (`half ? B.x : A.x`): with every index a compile-time constant the compiler
    // hoists all 8 x 6 selections out of the tile loop and then spills around the gathers; an opaque copy of `half` per tile keeps them
    // inside the loop (48 v_cndmask per 32 samples)
    int half_t = half;
    asm volatile("" : "+v"(half_t));
    encode8_h<LU>(lv, reinterpret_cast<const PnFusedLevel*>(wimg + PN_NET_HALF_BYTES / 16) + 8 * half_t, emb_rsrc, half_t, oob ? 0.f : u0, oob ? 0.f : u1, oob ? 0.f : u2, oob, feat);
    __builtin_amdgcn_sched_barrier(0);
    // ---- sigma net layer 0: 32 -> 64, ReLU
    f32x16 a0 = zero16(), a1 = zero16();
#pragma unroll
    for (int kc = 0; kc < 2; kc++) {
        f16x8 b;
#pragma unroll
        for (int i = 0; i < 8; i++) b[i] = feat[8 * kc + i];
        a0 = PN_HMFMA(W(0 + kc), b, a0);
        a1 = PN_HMFMA(W(2 + kc), b, a1);
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- sigma net layer 1: 64 -> 16
    f32x16 h2 = zero16();
#pragma unroll
    for (int kc = 0; kc < 4; kc++) h2 = PN_HMFMA(W(4 + kc), to_half8(kc < 2 ? a0 : a1, (kc & 1) * 8, true), h2);
#pragma unroll
    for (int r = 0; r < 8; r++) g2[r] = (float)(_Float16)h2[r];
}

// e[0..2]: the last Linear's float accumulators (the caller rounds them to half: the layer's half output)
__device__ __forceinline__ void tile_color_net_h(const uint4* __restrict__ wl, const uint4* wimg, int half, const float (&g2)[8], float dx, float dy, float dz,
                                                 float (&e)[3]) {
#pragma clang fp contract(fast)
    auto W = [&](int G) { return __builtin_bit_cast(f16x8, wl[G * 64]); };
    // ---- colour net input (PN_MAPL / PN_MAPU), rounded to half by the first colour Linear's input cast
    float sh[16];
    sh16(dx, dy, dz, sh);
    auto pick = [half](float a, float b) {
        asm volatile("" : "+v"(a), "+v"(b));
        return half ? a : b;
    };
    f16x8 vb[2];
#pragma unroll
    for (int k = 0; k < 7; k++) vb[0][k] = (_Float16)pick(g2[k], g2[k + 1]);
    vb[0][7] = (_Float16)pick(g2[7], sh[0]);
#pragma unroll
    for (int k = 8; k < 15; k++) vb[1][k - 8] = (_Float16)pick(sh[k + 1], sh[k - 7]);
    vb[1][7] = (_Float16)pick(0.0f, sh[8]);
    __builtin_amdgcn_sched_barrier(0);
    // ---- colour layer 0: 31 -> 64, ReLU
    f32x16 c0 = zero16(), c1 = zero16();
#pragma unroll
    for (int kc = 0; kc < 2; kc++) {
        c0 = PN_HMFMA(W(8 + kc), vb[kc], c0);
        c1 = PN_HMFMA(W(10 + kc), vb[kc], c1);
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- colour layer 1: 64 -> 64, ReLU
    f32x16 d0 = zero16(), d1 = zero16();
#pragma unroll
    for (int kc = 0; kc < 4; kc++) {
        const f16x8 b = to_half8(kc < 2 ? c0 : c1, (kc & 1) * 8, true);
        d0 = PN_HMFMA(W(12 + kc), b, d0);
        d1 = PN_HMFMA(W(16 + kc), b, d1);
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- colour layer 2: 64 -> 3 on the vector ALU: half-rounded weights and inputs, float accumulation (products of two halves are
    // exact in float), ONE rounding of each output to half
    e[0] = e[1] = e[2] = 0.f;
    {
        const float* __restrict__ wlast = reinterpret_cast<const float*>(wimg) + PN_NET_HALF_W_BYTES / 4 + half * 96;
#pragma unroll
        for (int q4 = 0; q4 < 8; q4++) {
            const float4 wa = *reinterpret_cast<const float4*>(wlast + q4 * 12);
            const float4 wb = *reinterpret_cast<const float4*>(wlast + q4 * 12 + 4);
            const float4 wc = *reinterpret_cast<const float4*>(wlast + q4 * 12 + 8);
            const f32x16& src = (q4 < 4) ? d0 : d1;
            const int r = (q4 & 3) * 4;
            float hv[4];
#pragma unroll
            for (int i = 0; i < 4; i++) { const _Float16 h = (_Float16)src[r + i]; hv[i] = (float)(h > (_Float16)0.0f ? h : (_Float16)0.0f); }
            e[0] = fmaf(wa.x, hv[0], e[0]); e[1] = fmaf(wa.y, hv[0], e[1]); e[2] = fmaf(wa.z, hv[0], e[2]);
            e[0] = fmaf(wa.w, hv[1], e[0]); e[1] = fmaf(wb.x, hv[1], e[1]); e[2] = fmaf(wb.y, hv[1], e[2]);
            e[0] = fmaf(wb.z, hv[2], e[0]); e[1] = fmaf(wb.w, hv[2], e[1]); e[2] = fmaf(wc.x, hv[2], e[2]);
            e[0] = fmaf(wc.y, hv[3], e[0]); e[1] = fmaf(wc.z, hv[3], e[1]); e[2] = fmaf(wc.w, hv[3], e[2]);
        }
#pragma unroll
        for (int o = 0; o < 3; o++) e[o] += __shfl_xor(e[o], 32);
    }
}
__device__ __forceinline__ float tile_rgb_out_h(float e) {
#pragma clang fp contract(fast)
    const float logit = (float)(_Float16)e;                                  // the last Linear's half output
    return (float)(_Float16)(1.0f / (1.0f + pn_expf(-logit)));               // torch.sigmoid on a half tensor
}

#ifdef PN_TU_FP_CONTRACT_OFF
#pragma clang fp contract(off)
#endif
