// Throughput of same-address global atomics on gfx950 (one per wave), returning and not, vs atomics spread over many cache lines.
//   hipcc --offload-arch=gfx950 -O3 -o bin/calib_atomic calib_atomic.hip && bin/calib_atomic
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>  // 0: returning, same address   1: non-returning, same address   2: returning, 256 lines   3: non-returning, 256 lines   4: no atomic
__global__ void k(int* counters, int* out, int work) {
    const int lane = threadIdx.x & 63, wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    float x = lane * 0.001f;
    for (int i = 0; i < work; i++) x = x * 1.0001f + 0.5f;  // some ALU work first (~work * 6 cycles)
    int r = (int)x;
    int* c = counters + ((MODE >= 2) ? (wave & 255) * 32 : 0);
    if (lane == 0) {
        if (MODE == 0 || MODE == 2) r += atomicAdd(c, 1);
        else if (MODE == 1 || MODE == 3) atomicAdd(c, 1);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int MODE>
static void run(const char* name, int* counters, int* out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("%-40s", name);
    for (int blocks : {256, 1250, 5000, 20000}) {
        k<MODE><<<blocks, 256>>>(counters, out, 500);
        hipEventRecord(e0); k<MODE><<<blocks, 256>>>(counters, out, 500); hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("  %6d waves: %7.1f us", blocks * 4, ms * 1e3);
    }
    printf("\n");
}
int main() {
    int *counters, *out;
    hipMalloc(&counters, 256 * 32 * 4); hipMemset(counters, 0, 256 * 32 * 4); hipMalloc(&out, 20000 * 256 * 4);
    run<4>("no atomic", counters, out);
    run<0>("returning, one address", counters, out);
    run<1>("non-returning, one address", counters, out);
    run<2>("returning, 256 cache lines", counters, out);
    run<3>("non-returning, 256 cache lines", counters, out);
    return 0;
}
