// Stand-alone reproducer for the claim of round 1 (DESIGN.md 4.2 "gfx950 finding 1"): packed-fp32 VALU instructions (v_pk_fma_f32 /
// v_pk_mul_f32 / v_pk_add_f32) issued while a v_mfma_f32_32x32x16_bf16 is in flight on the same SIMD corrupt the MFMA's result.
//
//   hipcc --offload-arch=gfx950 -O2 tools/repro_pk_mfma.hip -o tools/bin/repro_pk_mfma && tools/bin/repro_pk_mfma [iters]
//
// Workgroups of 512 threads = 8 waves = two waves per SIMD.  Waves 0-3 ("matrix") loop  acc = A x B + 0  on fixed bf16 operands and
// compare every result, bit for bit, with the first one; waves 4-7 ("vector") loop one of
//   mode 0: nothing (exit at once)            mode 1: v_fma_f32 (plain)           mode 2: v_pk_fma_f32 + v_pk_mul_f32 + v_pk_add_f32
// beside them.  Mode 3: ONE wave per SIMD that interleaves the packed instructions between its own MFMAs (independent registers).
// Output: one JSON line per mode with the number of MFMAs executed and the number whose accumulator differed.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

__global__ void __launch_bounds__(512) k_repro(int mode, int iters, unsigned long long* bad, unsigned long long* done, float* sink) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool matrix = (mode == 3) ? true : (wave < 4);
    if (mode == 3 && wave >= 4) return;  // one wave per SIMD
    if (matrix) {
        // bf16 operands with a spread of magnitudes, fixed per (block, wave, lane)
        uint4 au, bu;
        uint32_t s = mix(blockIdx.x * 9781u + wave * 131u + lane + 1u);
        uint32_t* ap = reinterpret_cast<uint32_t*>(&au);
        uint32_t* bp = reinterpret_cast<uint32_t*>(&bu);
        for (int i = 0; i < 4; i++) {
            s = mix(s); const uint32_t lo = 0x3f00u + (s & 0xffu), hi = 0xbf00u + ((s >> 8) & 0xffu);
            ap[i] = lo | (hi << 16);
            s = mix(s); const uint32_t lo2 = 0x3e80u + (s & 0xffu), hi2 = 0x3f80u + ((s >> 8) & 0x7fu);
            bp[i] = lo2 | (hi2 << 16);
        }
        const bf16x8 A = __builtin_bit_cast(bf16x8, au), B = __builtin_bit_cast(bf16x8, bu);
        f32x16 zero = {0};
        f32x16 first = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, zero, 0, 0, 0);
        f32x2 p = {1.0f + lane * 1e-3f, 0.5f}, q = {0.999f, 1.0001f}, r = {1e-4f, -1e-4f};
        unsigned long long nbad = 0;
        for (int it = 0; it < iters; it++) {
            uint32_t a0 = au.x, a1 = au.y, a2 = au.z, a3 = au.w;
            asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));  // opaque per iteration: the MFMA is not loop-invariant for the optimiser
            const uint4 at = make_uint4(a0, a1, a2, a3);
            f32x16 acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, at), B, zero, 0, 0, 0);
            if (mode == 3) {  // this wave's own packed VALU between its MFMAs, on registers the MFMA does not touch
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2\n\tv_pk_mul_f32 %0, %0, %1\n\tv_pk_add_f32 %0, %0, %2\n\tv_pk_fma_f32 %0, %0, %1, %2"
                             : "+v"(p) : "v"(q), "v"(r));
            }
            bool diff = false;
#pragma unroll
            for (int i = 0; i < 16; i++) diff |= (__float_as_uint(acc[i]) != __float_as_uint(first[i]));
            nbad += __popcll(__ballot(diff)) ? 1 : 0;
        }
        if (lane == 0) {
            atomicAdd(bad, nbad);
            atomicAdd(done, (unsigned long long)iters);
        }
        if (mode == 3) sink[blockIdx.x * 512 + threadIdx.x] = p[0] + p[1];
    } else {
        if (mode == 0) return;
        f32x2 p = {1.0f + lane * 1e-3f, 0.5f}, q = {0.999f, 1.0001f}, r = {1e-4f, -1e-4f};
        float a = 1.0f + lane * 1e-3f, b = 0.999f, c = 1e-4f;
        for (int it = 0; it < iters; it++) {
            if (mode == 2) {
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2\n\tv_pk_mul_f32 %0, %0, %1\n\tv_pk_add_f32 %0, %0, %2\n\tv_pk_fma_f32 %0, %0, %1, %2\n\t"
                             "v_pk_fma_f32 %0, %0, %1, %2\n\tv_pk_mul_f32 %0, %0, %1\n\tv_pk_add_f32 %0, %0, %2\n\tv_pk_fma_f32 %0, %0, %1, %2"
                             : "+v"(p) : "v"(q), "v"(r));
            } else {
                asm volatile("v_fma_f32 %0, %0, %1, %2\n\tv_mul_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %2\n\tv_fma_f32 %0, %0, %1, %2\n\t"
                             "v_fma_f32 %0, %0, %1, %2\n\tv_mul_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %2\n\tv_fma_f32 %0, %0, %1, %2"
                             : "+v"(a) : "v"(b), "v"(c));
            }
        }
        sink[blockIdx.x * 512 + threadIdx.x] = p[0] + p[1] + a;
    }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 400000;
    const int blocks = 512;  // two 512-thread workgroups per CU
    unsigned long long *bad, *done;
    float* sink;
    CK(hipMalloc(&bad, 8)); CK(hipMalloc(&done, 8)); CK(hipMalloc(&sink, (size_t)blocks * 512 * 4));
    const char* names[4] = {"matrix waves alone", "plain fp32 VALU in the co-resident waves", "packed fp32 VALU in the co-resident waves",
                            "packed fp32 VALU interleaved in the MFMA wave itself"};
    for (int mode = 0; mode < 4; mode++) {
        CK(hipMemset(bad, 0, 8)); CK(hipMemset(done, 0, 8));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0));
        k_repro<<<blocks, 512>>>(mode, iters, bad, done, sink);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long hb = 0, hd = 0;
        CK(hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hd, done, 8, hipMemcpyDeviceToHost));
        printf("{\"tool\": \"repro_pk_mfma\", \"mode\": %d, \"what\": \"%s\", \"mfma_executed\": %llu, \"mfma_results_differing\": %llu, \"ms\": %.2f}\n", mode,
               names[mode], hd, hb, ms);
    }
    return 0;
}
