// Cost of one vector-memory instruction on gfx950 by access pattern (L2-resident data, every CU full: 4096 waves), in CU cycles.
//   hipcc --offload-arch=gfx950 -O3 -o bin/calib_tcp calib_tcp.hip && bin/calib_tcp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define ITERS 256
// mode: address pattern of the 64 lanes
//  0 coalesced dwordx4 (lane i -> base + 16 i)        1 broadcast dwordx4 (all lanes same 16 B)
//  2 8 groups of 8 lanes, one 16 B per group (scattered groups)   3 divergent dwordx4 (64 scattered 16 B)
//  4 quads: each quad reads one scattered 64 B (lane q -> +16 q)   5 divergent dword (64 scattered 4 B)
//  6 coalesced dword    7 glds dwordx4 coalesced    8 glds dwordx4 quads scattered 64 B   9 dwordx3 per lane, consecutive lanes contiguous (12 B stride)
template <int MODE>
__global__ void k(const float4* __restrict__ buf, int n_vec, float* out, unsigned long long* ticks) {
    __shared__ float4 lds[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned r = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    float acc = 0.f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITERS; it++) {
        r = r * 1664525u + 1013904223u;
        const unsigned wr = __builtin_amdgcn_readfirstlane(r);            // wave-uniform random
        const unsigned gr = __shfl((int)r, lane & ~7);                    // per 8-lane group
        const unsigned qr = __shfl((int)r, lane & ~3);                    // per quad
        unsigned idx;
        if (MODE == 0 || MODE == 7) idx = (wr % (n_vec - 64)) + lane;
        else if (MODE == 1) idx = wr % n_vec;
        else if (MODE == 2) idx = gr % n_vec;
        else if (MODE == 3) idx = r % n_vec;
        else if (MODE == 4 || MODE == 8) idx = ((qr % (n_vec / 4)) * 4) + (lane & 3);
        else idx = 0;
        if (MODE <= 4) { const float4 v = buf[idx]; acc += v.x + v.w; }
        else if (MODE == 5) { acc += reinterpret_cast<const float*>(buf)[r % (n_vec * 4)]; }
        else if (MODE == 6) { acc += reinterpret_cast<const float*>(buf)[(wr % (n_vec * 4 - 64)) + lane]; }
        else if (MODE == 7 || MODE == 8) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(buf + idx), (__attribute__((address_space(3))) void*)lds[wave], 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            acc += lds[wave][63 - lane].x;
        } else if (MODE == 9) {
            const float* p = reinterpret_cast<const float*>(buf) + (wr % (n_vec * 4 - 256)) + lane * 3;
            struct __attribute__((packed, aligned(4))) F3 { float x, y, z; };
            const F3 v = *reinterpret_cast<const F3*>(p);
            acc += v.x + v.z;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::"v"(acc) : "memory");  // one dependent instruction at a time per wave
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0) ticks[blockIdx.x * 4 + wave] = t1 - t0;
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
template <int MODE>
static void run(const char* name, const float4* buf, int n_vec, float* out, unsigned long long* ticks) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int cfg = 0; cfg < 2; cfg++) {
        const int blocks = cfg == 0 ? 256 : 1024;  // 1 or 4 waves per SIMD
        k<MODE><<<blocks, 256>>>(buf, n_vec, out, ticks);
        hipEventRecord(e0); k<MODE><<<blocks, 256>>>(buf, n_vec, out, ticks); hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        // CU cycles per instruction = kernel time * clock / (instructions issued on one CU)
        const double per_cu = (double)blocks * 4 / 256 * ITERS;
        printf("%-44s %2d waves/CU: %7.1f us, %6.1f ns per instr per CU (latency %5.0f ticks)\n", name, blocks * 4 / 256, ms * 1e3, ms * 1e6 / per_cu, 0.0);
    }
}
int main() {
    const int n_vec = (2 << 20) / 16;  // 2 MB: L2 resident
    float4* buf; float* out; unsigned long long* ticks;
    hipMalloc(&buf, n_vec * 16); hipMemset(buf, 0, n_vec * 16); hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&ticks, 8 * 4096);
    run<0>("0 coalesced dwordx4 (1 KB contiguous)", buf, n_vec, out, ticks);
    run<1>("1 broadcast dwordx4 (one 16 B)", buf, n_vec, out, ticks);
    run<2>("2 8 groups x one scattered 16 B", buf, n_vec, out, ticks);
    run<3>("3 divergent dwordx4 (64 scattered 16 B)", buf, n_vec, out, ticks);
    run<4>("4 16 quads x one scattered 64 B", buf, n_vec, out, ticks);
    run<5>("5 divergent dword (64 scattered 4 B)", buf, n_vec, out, ticks);
    run<6>("6 coalesced dword (256 B contiguous)", buf, n_vec, out, ticks);
    run<7>("7 glds dwordx4 coalesced", buf, n_vec, out, ticks);
    run<8>("8 glds dwordx4, 16 quads x scattered 64 B", buf, n_vec, out, ticks);
    run<9>("9 dwordx3, 768 B contiguous", buf, n_vec, out, ticks);
    return 0;
}
