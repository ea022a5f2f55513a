// Throughput of scattered per-lane gathers on gfx950, the hash-grid encoder's access pattern: every lane of a wave loads a few bytes from an address of its
// own, B loads in flight per lane, every CU full.  Compares the return path through the vector registers with LDS-DMA (global_load_lds_dwordx4 + ds_read),
// on a table that fits the L2s and on one that does not.
//   hipcc --offload-arch=gfx950 -O3 -o bin/calib_gather calib_gather.hip && bin/calib_gather
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITERS 64
#define B 8
// MODE 0: dwordx2 to VGPRs (8 B per lane)   1: dwordx4 to VGPRs (16 B per lane)   2: LDS-DMA dwordx4 + ds_read_b64   3: LDS-DMA dword x 2 + ds_read_b64
//      4: dwordx2 to VGPRs, lanes of a quad adjacent (one 32 B segment per quad)
template <int MODE>
__global__ void k(const float2* __restrict__ buf, unsigned mask, float* out) {
    __shared__ __attribute__((aligned(16))) float4 lds[4][B][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned r = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    float acc = 0.f;
    for (int it = 0; it < ITERS; it++) {
        unsigned idx[B];
#pragma unroll
        for (int b = 0; b < B; b++) {
            r = r * 1664525u + 1013904223u;
            unsigned v = r >> 3;
            if (MODE == 4) v = (__shfl((int)v, lane & ~3) & ~3u) + (lane & 3);
            idx[b] = v & mask & ~1u;   // even entry: 16-byte aligned
        }
        if (MODE == 0 || MODE == 4) {
            float2 v[B];
#pragma unroll
            for (int b = 0; b < B; b++) v[b] = buf[idx[b]];
#pragma unroll
            for (int b = 0; b < B; b++) acc += v[b].x + v[b].y;
        } else if (MODE == 1) {
            float4 v[B];
#pragma unroll
            for (int b = 0; b < B; b++) v[b] = *reinterpret_cast<const float4*>(buf + idx[b]);
#pragma unroll
            for (int b = 0; b < B; b++) acc += v[b].x + v[b].w;
        } else if (MODE == 2) {
#pragma unroll
            for (int b = 0; b < B; b++)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(buf + idx[b]), (__attribute__((address_space(3))) void*)lds[wave][b], 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int b = 0; b < B; b++) { const float2 v = *reinterpret_cast<const float2*>(&lds[wave][b][lane]); acc += v.x + v.y; }
        } else if (MODE == 3) {
            float* l = reinterpret_cast<float*>(lds[wave]);
#pragma unroll
            for (int b = 0; b < B; b++) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(reinterpret_cast<const float*>(buf + idx[b])), (__attribute__((address_space(3))) void*)(l + b * 128), 4, 0, 0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(reinterpret_cast<const float*>(buf + idx[b]) + 1), (__attribute__((address_space(3))) void*)(l + b * 128 + 64), 4, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int b = 0; b < B; b++) acc += l[b * 128 + lane] + l[b * 128 + 64 + lane];
        }
        asm volatile("" ::"v"(acc) : "memory");
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
template <int MODE>
static void run(const char* name, const float2* buf, unsigned mask, float* out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int cfg = 0; cfg < 3; cfg++) {
        const int blocks = cfg == 0 ? 256 : (cfg == 1 ? 768 : 2048);  // 1, 3, 8 waves per SIMD
        k<MODE><<<blocks, 256>>>(buf, mask, out);
        hipEventRecord(e0); k<MODE><<<blocks, 256>>>(buf, mask, out); hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double instr_per_cu = (double)blocks * 4 / 256 * ITERS * B;   // gathers (of one entry per lane) issued on one CU
        printf("%-52s %2d waves/CU: %8.1f us, %6.1f ns per wave-gather per CU = %5.1f cycles at 2.4 GHz\n", name, blocks * 4 / 256, ms * 1e3, ms * 1e6 / instr_per_cu, ms * 1e6 / instr_per_cu * 2.4);
    }
}
int main() {
    const size_t big = 64u << 20;
    float2* buf; float* out;
    hipMalloc(&buf, big); hipMemset(buf, 0, big); hipMalloc(&out, 2048 * 256 * 4);
    for (int pass = 0; pass < 2; pass++) {
        const unsigned mask = pass == 0 ? ((2u << 20) / 8 - 1) : (unsigned)(big / 8 - 1);
        printf("---- table of %s\n", pass == 0 ? "2 MB (L2 resident)" : "64 MB (the chair's hash tables)");
        run<0>("0 dwordx2 -> VGPR, 64 scattered 8 B", buf, mask, out);
        run<1>("1 dwordx4 -> VGPR, 64 scattered 16 B", buf, mask, out);
        run<2>("2 LDS-DMA dwordx4 (64 scattered 16 B) + ds_read_b64", buf, mask, out);
        run<3>("3 LDS-DMA dword x 2 + ds_read", buf, mask, out);
        run<4>("4 dwordx2 -> VGPR, quads adjacent", buf, mask, out);
    }
    return 0;
}
