// Calibration of __builtin_readcyclecounter() (s_memtime) on gfx950 against instruction issue and memory latency.
//   hipcc --offload-arch=gfx950 -O3 -o bin/calib_clock calib_clock.hip && bin/calib_clock
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k_alu(float* out, unsigned long long* ticks, int iters) {
    float x = threadIdx.x * 1e-3f, y = 1.0001f;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 64; u++) x = x * y + 0.5f;  // dependent chain
    }
    asm volatile("s_nop 0" ::"v"(x));
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x % 64 == 0) ticks[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}
__global__ void k_chase(const int* next, int* out, unsigned long long* ticks, int hops) {
    int p = threadIdx.x + blockIdx.x * blockDim.x;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < hops; i++) p = next[p];
    asm volatile("s_waitcnt vmcnt(0)" ::"v"(p));
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x % 64 == 0) ticks[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
    out[threadIdx.x + blockIdx.x * blockDim.x] = p;
}
int main() {
    float* out; unsigned long long* ticks; int *next, *iout;
    const int maxw = 1 << 16;
    hipMalloc(&out, sizeof(float) * maxw * 64); hipMalloc(&ticks, sizeof(unsigned long long) * maxw);
    std::vector<unsigned long long> h(maxw);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int cfg = 0; cfg < 3; cfg++) {
        const int blocks = cfg == 0 ? 1 : (cfg == 1 ? 256 : 256 * 4), threads = cfg == 0 ? 64 : 256, iters = 256;
        k_alu<<<blocks, threads>>>(out, ticks, iters);
        hipEventRecord(e0); k_alu<<<blocks, threads>>>(out, ticks, iters); hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const int nw = blocks * threads / 64;
        hipMemcpy(h.data(), ticks, sizeof(unsigned long long) * nw, hipMemcpyDeviceToHost);
        double s = 0; for (int i = 0; i < nw; i++) s += h[i];
        printf("alu: %d waves (%d per SIMD): %.1f ticks per dependent v_fma per wave, kernel %.1f us -> %.2f ns per instr per wave\n", nw,
               cfg == 2 ? 4 : 1, s / nw / (iters * 64.0), ms * 1e3, ms * 1e6 / (iters * 64.0));
    }
    // pointer chase: random permutation over 64 MB (beyond L2), then over 1 MB (L2 resident)
    for (int sz = 0; sz < 2; sz++) {
        const int n = sz == 0 ? (16 << 20) : (256 << 10);
        std::vector<int> perm(n); for (int i = 0; i < n; i++) perm[i] = i;
        unsigned long long r = 88172645463325252ull;
        for (int i = n - 1; i > 0; i--) { r ^= r << 13; r ^= r >> 7; r ^= r << 17; int j = r % (i + 1); int t = perm[i]; perm[i] = perm[j]; perm[j] = t; }
        std::vector<int> nx(n); for (int i = 0; i < n; i++) nx[perm[i]] = perm[(i + 1) % n];
        hipMalloc(&next, sizeof(int) * n); hipMalloc(&iout, sizeof(int) * maxw * 64);
        hipMemcpy(next, nx.data(), sizeof(int) * n, hipMemcpyHostToDevice);
        for (int cfg = 0; cfg < 3; cfg++) {
            const int blocks = cfg == 0 ? 1 : (cfg == 1 ? 256 : 1024), threads = cfg == 0 ? 64 : 256, hops = 64;
            k_chase<<<blocks, threads>>>(next, iout, ticks, hops);
            hipEventRecord(e0); k_chase<<<blocks, threads>>>(next, iout, ticks, hops); hipEventRecord(e1); hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const int nw = blocks * threads / 64;
            hipMemcpy(h.data(), ticks, sizeof(unsigned long long) * nw, hipMemcpyDeviceToHost);
            double s = 0; for (int i = 0; i < nw; i++) s += h[i];
            printf("chase %d MB: %d waves: %.0f ticks per dependent 64-lane gather, %.2f us per hop (kernel %.1f us)\n", n >> 18, nw, s / nw / hops,
                   ms * 1e3 / hops, ms * 1e3);
        }
        hipFree(next); hipFree(iout);
    }
    return 0;
}
