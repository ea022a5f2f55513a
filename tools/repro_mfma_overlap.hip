// Is a v_mfma_f32_32x32x16_{f16,bf16} whose DESTINATION registers overlap its A / B source registers (C = literal 0) computed correctly on
// gfx950?  hipcc (ROCm 7.2) emits such encodings — e.g. `v_mfma_f32_32x32x16_f16 v[0:15], v[0:3], v[4:7], 0` was found in an earlier
// build of k_nerf_forward_h — and round 1 avoided them "as a precaution" (finding 2) without evidence either way.
//
//   hipcc --offload-arch=gfx950 -O2 tools/repro_mfma_overlap.hip -o tools/bin/repro_mfma_overlap && tools/bin/repro_mfma_overlap
//
// Each wave computes D = A x B three ways on the same random fp16 operands, with hand-placed registers:
//   ref : v_mfma ... v[16:31], v[0:3], v[4:7], 0      (disjoint)
//   ovA : v_mfma ... v[0:15],  v[0:3], v[4:7], 0      (destination covers A and B — the encoding hipcc produced)
//   ovB : v_mfma ... v[2:17],  v[0:3], v[4:7], 0      (partial overlap)
// and reports how many waves got a different result.  Many waves per SIMD run concurrently so that timing varies.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define MOV_IN "v_mov_b32 v0, %16\n\tv_mov_b32 v1, %17\n\tv_mov_b32 v2, %18\n\tv_mov_b32 v3, %19\n\tv_mov_b32 v4, %20\n\tv_mov_b32 v5, %21\n\tv_mov_b32 v6, %22\n\tv_mov_b32 v7, %23\n\ts_nop 4\n\t"
#define WAIT "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\t"
#define OUT16(b) "v_mov_b32 %0, v" #b "\n\t"
#define CLOB "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", \
             "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31"

#define RUN(NAME, MFMA, R0, R1, R2, R3, R4, R5, R6, R7, R8, R9, R10, R11, R12, R13, R14, R15)                                                       \
    __device__ __forceinline__ void NAME(const uint32_t* a, const uint32_t* b, float* o) {                                                         \
        asm volatile(MOV_IN MFMA "\n\t" WAIT                                                                                                       \
                     "v_mov_b32 %0, v" #R0 "\n\tv_mov_b32 %1, v" #R1 "\n\tv_mov_b32 %2, v" #R2 "\n\tv_mov_b32 %3, v" #R3 "\n\t"                    \
                     "v_mov_b32 %4, v" #R4 "\n\tv_mov_b32 %5, v" #R5 "\n\tv_mov_b32 %6, v" #R6 "\n\tv_mov_b32 %7, v" #R7 "\n\t"                    \
                     "v_mov_b32 %8, v" #R8 "\n\tv_mov_b32 %9, v" #R9 "\n\tv_mov_b32 %10, v" #R10 "\n\tv_mov_b32 %11, v" #R11 "\n\t"                \
                     "v_mov_b32 %12, v" #R12 "\n\tv_mov_b32 %13, v" #R13 "\n\tv_mov_b32 %14, v" #R14 "\n\tv_mov_b32 %15, v" #R15                   \
                     : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7]), "=&v"(o[8]), "=&v"(o[9]), \
                       "=&v"(o[10]), "=&v"(o[11]), "=&v"(o[12]), "=&v"(o[13]), "=&v"(o[14]), "=&v"(o[15])                                           \
                     : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3])                                       \
                     : CLOB);                                                                                                                       \
    }

RUN(run_ref, "v_mfma_f32_32x32x16_f16 v[16:31], v[0:3], v[4:7], 0", 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31)
RUN(run_ovA, "v_mfma_f32_32x32x16_f16 v[0:15], v[0:3], v[4:7], 0", 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15)
RUN(run_ovB, "v_mfma_f32_32x32x16_f16 v[2:17], v[0:3], v[4:7], 0", 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17)
RUN(run_ref_bf, "v_mfma_f32_32x32x16_bf16 v[16:31], v[0:3], v[4:7], 0", 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31)
RUN(run_ovA_bf, "v_mfma_f32_32x32x16_bf16 v[0:15], v[0:3], v[4:7], 0", 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15)

__device__ __forceinline__ uint32_t mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

__global__ void __launch_bounds__(256) k_overlap(int iters, unsigned long long* counts /* [5]: waves run, ovA diff, ovB diff, bf16 ovA diff, nonzero check */) {
    const int lane = threadIdx.x & 63;
    unsigned long long dA = 0, dB = 0, dAbf = 0, nz = 0;
    for (int it = 0; it < iters; it++) {
        uint32_t a[4], b[4];
        uint32_t s = mix((blockIdx.x * 256u + threadIdx.x) * 2654435761u + it * 40503u + 7u);
        for (int i = 0; i < 4; i++) {  // fp16 / bf16 pairs of moderate magnitude (exponent field in a safe range for both formats)
            s = mix(s); a[i] = (0x3800u + (s & 0x7ffu)) | ((0xb800u + ((s >> 11) & 0x7ffu)) << 16);
            s = mix(s); b[i] = (0x3400u + (s & 0x7ffu)) | ((0x3a00u + ((s >> 11) & 0x3ffu)) << 16);
        }
        float r[16], x[16], y[16], rb[16], xb[16];
        run_ref(a, b, r); run_ovA(a, b, x); run_ovB(a, b, y); run_ref_bf(a, b, rb); run_ovA_bf(a, b, xb);
        bool da = false, db = false, dab = false, any = false;
        for (int i = 0; i < 16; i++) {
            da |= __float_as_uint(r[i]) != __float_as_uint(x[i]);
            db |= __float_as_uint(r[i]) != __float_as_uint(y[i]);
            dab |= __float_as_uint(rb[i]) != __float_as_uint(xb[i]);
            any |= r[i] != 0.0f;
        }
        dA += __ballot(da) ? 1 : 0; dB += __ballot(db) ? 1 : 0; dAbf += __ballot(dab) ? 1 : 0; nz += __ballot(any) ? 1 : 0;
    }
    if (lane == 0) {
        atomicAdd(counts, (unsigned long long)iters); atomicAdd(counts + 1, dA); atomicAdd(counts + 2, dB); atomicAdd(counts + 3, dAbf); atomicAdd(counts + 4, nz);
    }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

int main() {
    unsigned long long* c;
    CK(hipMalloc(&c, 40)); CK(hipMemset(c, 0, 40));
    k_overlap<<<4096, 256>>>(2000, c);
    CK(hipDeviceSynchronize());
    unsigned long long h[5];
    CK(hipMemcpy(h, c, 40, hipMemcpyDeviceToHost));
    printf("{\"tool\": \"repro_mfma_overlap\", \"wave_mfma_triples\": %llu, \"f16_dst_covers_A_and_B_differs\": %llu, \"f16_dst_partial_overlap_differs\": %llu, "
           "\"bf16_dst_covers_A_and_B_differs\": %llu, \"reference_nonzero\": %llu}\n", h[0], h[1], h[2], h[3], h[4]);
    return 0;
}
