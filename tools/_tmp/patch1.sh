python - <<'EOF'
p='/root/repo/pienerf_amd/csrc/pn_nerf_forward.hip'
s=open(p).read()
marker='''int pn_nerf_forward_launch(const pn_net* net,'''
new_kernel = r'''// ------------------------------------------------------------------------------------------------ bf16 three-way split MLP
// The dense layers on the bf16 matrix pipe at fp32 accuracy.  An fp32 value is cut, by truncation, into three bf16 pieces
// x = hi + mid + lo (8 + 8 + 8 significant bits, exact), weights likewise on the host; a product x*w is the six partial
// products whose weight is >= 2^-16 (hi*lo, lo*hi, mid*mid, hi*mid, mid*hi, hi*hi; the dropped mid*lo, lo*mid, lo*lo are
// <= 2^-23 |x*w|, the size of fp32's own product rounding), accumulated smallest first in the fp32 accumulator of
// v_mfma_f32_32x32x16_bf16.  Six bf16 MFMAs of 32 cycles replace sixteen 32x32x2 f32 MFMAs of 64 cycles, and — unlike the
// f32-input MFMA, which executes on the fp32 vector ALUs (DESIGN.md 4.2) — they run beside the VALU work of the other waves.
// Lane layout, D = W·X^T and the D-layout-is-the-next-B-layout property are those of k_nerf_forward: the K index of chunk kc,
// lane half h, element e is whatever feature that lane holds in register 8 kc + e, and pn_net_create lays the weights out to match.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
struct Split8 { uint4 hi, mid, lo; };

__device__ __forceinline__ uint32_t hi_pair(float a, float b) {  // bf16 (truncated) of a in the low half, of b in the high half
    return __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u);
}
__device__ __forceinline__ float drop_hi(float x) { return x - __uint_as_float(__float_as_uint(x) & 0xffff0000u); }
__device__ __forceinline__ Split8 split8(float x0, float x1, float x2, float x3, float x4, float x5, float x6, float x7) {
    Split8 o;
    o.hi = make_uint4(hi_pair(x0, x1), hi_pair(x2, x3), hi_pair(x4, x5), hi_pair(x6, x7));
    x0 = drop_hi(x0); x1 = drop_hi(x1); x2 = drop_hi(x2); x3 = drop_hi(x3);
    x4 = drop_hi(x4); x5 = drop_hi(x5); x6 = drop_hi(x6); x7 = drop_hi(x7);
    o.mid = make_uint4(hi_pair(x0, x1), hi_pair(x2, x3), hi_pair(x4, x5), hi_pair(x6, x7));
    x0 = drop_hi(x0); x1 = drop_hi(x1); x2 = drop_hi(x2); x3 = drop_hi(x3);
    x4 = drop_hi(x4); x5 = drop_hi(x5); x6 = drop_hi(x6); x7 = drop_hi(x7);
    o.lo = make_uint4(hi_pair(x0, x1), hi_pair(x2, x3), hi_pair(x4, x5), hi_pair(x6, x7));
    return o;
}
#define PN_BMFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, (a)), __builtin_bit_cast(bf16x8, (b)), (c), 0, 0, 0)

// acc += W(group G) · x for one K chunk: the six partial products, smallest first
__device__ __forceinline__ f32x16 split_mac(const uint4* __restrict__ wl, int G, const Split8& x, f32x16 acc) {
    const uint4 wh = wl[(G * 3 + 0) * 64], wm = wl[(G * 3 + 1) * 64], wo = wl[(G * 3 + 2) * 64];
    acc = PN_BMFMA(wo, x.hi, acc);
    acc = PN_BMFMA(wh, x.lo, acc);
    acc = PN_BMFMA(wm, x.mid, acc);
    acc = PN_BMFMA(wm, x.hi, acc);
    acc = PN_BMFMA(wh, x.mid, acc);
    acc = PN_BMFMA(wh, x.hi, acc);
    return acc;
}
__device__ __forceinline__ Split8 split8_of(const f32x16& v, int r0) {
    return split8(v[r0], v[r0 + 1], v[r0 + 2], v[r0 + 3], v[r0 + 4], v[r0 + 5], v[r0 + 6], v[r0 + 7]);
}

#define PN_BF_WAVES 6  // waves per workgroup: 2 workgroups x 61 KB LDS image per CU = 3 waves per SIMD
template <int MINW, int LU>
__global__ void __launch_bounds__(PN_BF_WAVES * 64, MINW) k_nerf_forward_bf(const PnFusedLevel* __restrict__ lv, const float* __restrict__ emb,
                                                                          const uint4* __restrict__ wsplit, float bound, const float* __restrict__ xyzs,
                                                                          const float* __restrict__ dirs, const int* __restrict__ list,
                                                                          const int* __restrict__ count_dev, uint32_t M_arg, float density_scale,
                                                                          float* __restrict__ sigmas, float* __restrict__ rgbs) {
    extern __shared__ __attribute__((aligned(16))) uint4 wimg[];  // PN_NET_SPLIT_BYTES
    const uint32_t M = count_dev ? (uint32_t)*count_dev : M_arg;
    const uint32_t n_tiles = (M + 31) / 32;
    const uint32_t waves_total = gridDim.x * PN_BF_WAVES;
    const uint32_t wave = blockIdx.x * PN_BF_WAVES + (threadIdx.x >> 6);
    if (blockIdx.x * PN_BF_WAVES >= n_tiles) return;  // no tile for any wave of this block
    for (int i = threadIdx.x; i < PN_NET_SPLIT_BYTES / 16; i += PN_BF_WAVES * 64) wimg[i] = wsplit[i];
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
        // GridEncoder.forward: inputs = (x + bound) / (2 * bound)  (gridencoder/grid.py:149)
        const float u0 = (x + bound) / (2 * bound), u1 = (y + bound) / (2 * bound), u2 = (z + bound) / (2 * bound);
        const bool oob = (u0 < 0 || u0 > 1 || u1 < 0 || u1 > 1 || u2 < 0 || u2 > 1);
        float feat[16];
        encode8<LU>(lv, emb, half, oob ? 0.f : u0, oob ? 0.f : u1, oob ? 0.f : u2, oob, feat);
        __builtin_amdgcn_sched_barrier(0);
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
        const float sigma_logit = h2[0];  // row 0 lives in the low half's register 0
        __builtin_amdgcn_sched_barrier(0);
        // ---- colour net input: 16 values per lane (see PN_MAPL / PN_MAPU)
        float sh[16];
        sh16(dx, dy, dz, sh);
        float v[16];
        auto pick = [half](float a, float b) {  // see k_nerf_forward
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
        // ---- colour layer 2: 64 -> 3 on the vector ALU (see k_nerf_forward)
        float e[3] = {0.f, 0.f, 0.f};
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
        if (valid && half == 0) {
            sigmas[slot] = density_scale * expf(sigma_logit);           // trunc_exp forward = exp (nerf/activation.py:8-10)
            rgbs[slot * 3 + 0] = 1.0f / (1.0f + expf(-e[0]));          // torch.sigmoid
            rgbs[slot * 3 + 1] = 1.0f / (1.0f + expf(-e[1]));
            rgbs[slot * 3 + 2] = 1.0f / (1.0f + expf(-e[2]));
        }
    }
}

'''
s=s.replace(marker, new_kernel+marker)
s=s.replace('''    const size_t lds = sizeof(float) * PN_NET_MFMAS * 64;
    static int variant = -1;''','''    static const uint32_t mlp = pn_env_u32("PN_NERF_MLP", 1);  // 1: bf16 three-way split MFMA, 0: f32-input MFMA
    if (mlp) {
        static const uint32_t bf_blocks = pn_env_u32("PN_NERF_BF_BLOCKS", 512);  // 2 workgroups per CU x 256 CUs
        static bool attr_set = false;
        if (!attr_set) {
            PN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_nerf_forward_bf<2, 4>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                             PN_NET_SPLIT_BYTES));
            attr_set = true;
        }
        uint32_t nb = pn_div_up(tiles, PN_BF_WAVES);
        if (nb > bf_blocks) nb = bf_blocks;
        k_nerf_forward_bf<2, 4><<<nb, PN_BF_WAVES * 64, PN_NET_SPLIT_BYTES, stream>>>((const PnFusedLevel*)net->fused_levels, net->embeddings,
                                                                                     (const uint4*)net->wsplit, net->bound, xyzs, dirs, list, ctl_count,
                                                                                     M_max, density_scale, sigmas, rgbs);
        PN_LAUNCH_CHECK();
        return PN_OK;
    }
    const size_t lds = sizeof(float) * PN_NET_MFMAS * 64;
    static int variant = -1;''')
open(p,'w').write(s)
EOF
python -m pienerf_amd.build 2>&1 | grep -E "error|warning" | head; cd /tmp && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fhip-fp32-correctly-rounded-divide-sqrt -ffp-contract=fast --cuda-device-only -S -o /tmp/nf.s /root/repo/pienerf_amd/csrc/pn_nerf_forward.hip 2>/dev/null; grep -E "^\s+\.(vgpr_count|private_segment_fixed_size|name):|\.vgpr_spill_count|\.agpr_count" /tmp/nf.s | paste - - - - - | grep nerf_forward