// What a device-wide barrier inside a persistent kernel costs on gfx950 (8 XCDs, L2 per XCD, not coherent with each other), against the
// boundary between two dependent kernels of a captured graph.  The quantity that decides whether the substep (4 dependent launches per
// local/global iteration) should become one cooperative kernel with 3 barriers per iteration.
//   hipcc --offload-arch=gfx950 -O3 -o bin/calib_barrier calib_barrier.hip && bin/calib_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

// flat: every workgroup bumps ONE counter (device-scope atomic = executed at the memory side) and polls it
__global__ void k_flat(int* ctr, int n_bar, float* sink) {
    float x = threadIdx.x;
    for (int b = 0; b < n_bar; b++) {
        x = x * 1.0001f + 1.0f;
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int want = (int)gridDim.x * (b + 1);
            while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
    }
    if (x == 12345.f) *sink = x;
}
// hierarchical: workgroup i runs on XCD i % 8; arrivals are counted per XCD (8 counters on separate cache lines), the last arrival of an
// XCD bumps the global counter, the last XCD publishes the generation, everybody polls the generation word
__global__ void k_hier(int* xcd_ctr, int* glob, int* gen, int n_bar, float* sink) {
    float x = threadIdx.x;
    const int xcd = blockIdx.x & 7, per_xcd = ((int)gridDim.x + 7 - xcd) / 8;
    for (int b = 0; b < n_bar; b++) {
        x = x * 1.0001f + 1.0f;
        __syncthreads();
        if (threadIdx.x == 0) {
            const int a = __hip_atomic_fetch_add(xcd_ctr + xcd * 32, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (a == per_xcd * (b + 1) - 1) {
                const int g = __hip_atomic_fetch_add(glob, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (g == 8 * (b + 1) - 1) __hip_atomic_store(gen, b + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            while (__hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < b + 1) __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
    }
    if (x == 12345.f) *sink = x;
}
// the same with release / acquire semantics (what a barrier that orders ordinary loads and stores around it needs: L2 write-back + invalidate)
__global__ void k_flat_fenced(int* ctr, int n_bar, float* data, float* sink) {
    float x = threadIdx.x;
    for (int b = 0; b < n_bar; b++) {
        data[(blockIdx.x * blockDim.x + threadIdx.x)] = x;   // something dirty in this XCD's L2
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            const int want = (int)gridDim.x * (b + 1);
            while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) __builtin_amdgcn_s_sleep(1);
            __atomic_thread_fence(__ATOMIC_ACQUIRE);
        }
        __syncthreads();
        x += data[((blockIdx.x + 1) % gridDim.x) * blockDim.x + threadIdx.x];   // read a neighbour's value written before the barrier
    }
    if (x == 12345.f) *sink = x;
}
__global__ void k_small(float* data, int n) {  // a dependent small kernel: read what the previous launch wrote
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) data[i] = data[(i + 256) % n] * 1.0001f + 1.0f;
}

int main() {
    int *ctr; float *sink, *data;
    hipMalloc(&ctr, 4096); hipMalloc(&sink, 4); hipMalloc(&data, 1024 * 256 * 4); hipMemset(data, 0, 1024 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipStream_t st; hipStreamCreate(&st);
    const int NB = 200;
    for (int blocks : {64, 128, 256, 512}) {
        float t0, t1, t2;
        hipMemsetAsync(ctr, 0, 4096, st); k_flat<<<blocks, 256, 0, st>>>(ctr, 1, sink);
        hipMemsetAsync(ctr, 0, 4096, st); hipEventRecord(e0, st); k_flat<<<blocks, 256, 0, st>>>(ctr, NB, sink); hipEventRecord(e1, st); hipStreamSynchronize(st);
        hipEventElapsedTime(&t0, e0, e1);
        hipMemsetAsync(ctr, 0, 4096, st); hipEventRecord(e0, st); k_hier<<<blocks, 256, 0, st>>>(ctr, ctr + 512, ctr + 768, NB, sink); hipEventRecord(e1, st); hipStreamSynchronize(st);
        hipEventElapsedTime(&t1, e0, e1);
        hipMemsetAsync(ctr, 0, 4096, st); hipEventRecord(e0, st); k_flat_fenced<<<blocks, 256, 0, st>>>(ctr, NB, data, sink); hipEventRecord(e1, st); hipStreamSynchronize(st);
        hipEventElapsedTime(&t2, e0, e1);
        printf("%4d workgroups x 256: flat barrier %6.2f us   hierarchical (per-XCD counters) %6.2f us   flat with release/acquire + data exchange %6.2f us\n", blocks,
               t0 * 1e3 / NB, t1 * 1e3 / NB, t2 * 1e3 / NB);
    }
    // kernel boundaries: NB dependent small launches, eager and as one graph
    for (int blocks : {16, 256}) {
        const int n = blocks * 256;
        for (int i = 0; i < 10; i++) k_small<<<blocks, 256, 0, st>>>(data, n);
        hipEventRecord(e0, st);
        for (int i = 0; i < NB; i++) k_small<<<blocks, 256, 0, st>>>(data, n);
        hipEventRecord(e1, st); hipStreamSynchronize(st);
        float te; hipEventElapsedTime(&te, e0, e1);
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
        for (int i = 0; i < NB; i++) k_small<<<blocks, 256, 0, st>>>(data, n);
        hipStreamEndCapture(st, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        hipGraphLaunch(ge, st); hipStreamSynchronize(st);
        hipEventRecord(e0, st); hipGraphLaunch(ge, st); hipEventRecord(e1, st); hipStreamSynchronize(st);
        float tg; hipEventElapsedTime(&tg, e0, e1);
        printf("%4d-workgroup dependent kernels: %6.2f us per launch eager, %6.2f us inside a graph\n", blocks, te * 1e3 / NB, tg * 1e3 / NB);
    }
    return 0;
}
