// What ONE all-to-all data exchange between the workgroups of a persistent kernel costs on gfx950 when nothing is fenced: the producers
// write their doubles with relaxed agent-scope stores (sc1: written through to the memory side, no L2 write-back of anything else), wait for the
// stores' acknowledgement (s_waitcnt vmcnt(0)), publish a per-workgroup generation tag, the consumers poll all tags with ONE coalesced load per
// round and then read the data with relaxed agent-scope loads (sc1: never served from this XCD's possibly stale L2 line).  This is the primitive
// of the persistent substep (pn_sim.hip: k_substep_persistent); tools/calib_barrier.hip measured the counting barriers and the fenced form.
//   hipcc --offload-arch=gfx950 -O3 -o bin/calib_exchange calib_exchange.hip && bin/calib_exchange
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ void st_agent(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ld_agent(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// every workgroup writes `own` doubles per exchange and reads `want` doubles (a window of the shared vector starting at its own block).
// MODE 0: one tag per workgroup, everybody polls all G tags;  MODE 1: the producer writes its tag into 8 replicas (one per XCD), a consumer polls
// the replica of its own XCD (blockIdx & 7): 32 readers per line instead of 256;  MODE 2: no tags, the hierarchical counting barrier of
// calib_barrier.hip (per-XCD arrival counters, a generation word) after the stores were acknowledged.
template <int THREADS, int MODE>
__global__ void __launch_bounds__(THREADS) k_exchange(double* buf, int* tags, int own, int want, int total, int n_ex, int* bad, double* sink) {
    __shared__ int ok_s;
    double acc = 0.0;
    const int G = gridDim.x, w = blockIdx.x, xcd = w & 7;
    int* xcd_ctr = tags + 4096; int* glob = tags + 4096 + 512; int* gen = tags + 4096 + 768;
    const int per_xcd = (G + 7 - xcd) / 8;
    for (int g = 1; g <= n_ex; g++) {
        double* cur = buf + (size_t)(g % 3) * total;  // three buffers in rotation, like the three exchanges of a local/global iteration
        for (int i = threadIdx.x; i < own; i += THREADS) st_agent(cur + (size_t)w * own + i, (double)(g * 1000 + w) + acc * 1e-30);
        __builtin_amdgcn_s_waitcnt(0);  // stores acknowledged
        __syncthreads();
        if (MODE == 0) { if (threadIdx.x == 0) __hip_atomic_store(tags + w, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        if (MODE == 1) { if (threadIdx.x < 8) __hip_atomic_store(tags + threadIdx.x * 512 + w, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        int spins = 0;
        if (MODE == 2) {
            if (threadIdx.x == 0) {
                const int a = __hip_atomic_fetch_add(xcd_ctr + xcd * 32, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (a == per_xcd * g - 1) {
                    const int gg = __hip_atomic_fetch_add(glob, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (gg == 8 * g - 1) __hip_atomic_store(gen, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                while (__hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < g && ++spins < (1 << 20)) __builtin_amdgcn_s_sleep(1);
                if (spins >= (1 << 20)) atomicAdd(bad, 1);
            }
            __syncthreads();
        } else {
            const int* my = MODE == 1 ? tags + xcd * 512 : tags;
            while (true) {  // thread t watches tag t (G <= THREADS)
                if (threadIdx.x == 0) ok_s = 1;
                __syncthreads();
                const bool late = (int)threadIdx.x < G && __hip_atomic_load(my + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < g;
                if (late) ok_s = 0;
                __syncthreads();
                const int ok = ok_s;
                __syncthreads();
                if (ok) break;
                if (++spins > (1 << 18)) { if (threadIdx.x == 0) atomicAdd(bad, 1); return; }
            }
        }
        int wrong = 0;
        for (int i0 = threadIdx.x; i0 < want; i0 += THREADS * 8) {  // eight independent loads in flight per thread
            double v[8]; int src[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int i = i0 + u * THREADS;
                const int j = (int)(((size_t)w * own + (i < want ? i : 0)) % (size_t)total);
                src[u] = j / own;
                v[u] = ld_agent(cur + j);
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                if (v[u] < (double)(g * 1000 + src[u]) - 0.5 || v[u] > (double)(g * 1000 + src[u]) + 0.5) wrong++;
                acc += v[u];
            }
        }
        if (wrong) atomicAdd(bad + 1, wrong);
        __syncthreads();
    }
    if (acc == 12345.0) *sink = acc;
}

template <int MODE>
static void run(const char* name, double* buf, int* tags, int* bad, double* sink, hipStream_t st, hipEvent_t e0, hipEvent_t e1) {
    const int NB = 300;
    for (int G : {64, 256}) {
        for (int own : {16, 128}) {
            for (int frac : {8, 1}) {  // each workgroup reads total / frac doubles
                const int total = G * own, want = total / frac;
                hipMemsetAsync(tags, 0, 32768, st); hipMemsetAsync(bad, 0, 8, st);
                k_exchange<512, MODE><<<G, 512, 0, st>>>(buf, tags, own, want, total, 2, bad, sink);
                hipMemsetAsync(tags, 0, 32768, st);
                hipEventRecord(e0, st);
                k_exchange<512, MODE><<<G, 512, 0, st>>>(buf, tags, own, want, total, NB, bad, sink);
                hipEventRecord(e1, st); hipStreamSynchronize(st);
                float t; hipEventElapsedTime(&t, e0, e1);
                int hb[2]; hipMemcpy(hb, bad, 8, hipMemcpyDeviceToHost);
                printf("%-34s %4d workgroups x 512: each writes %4d doubles, reads %6d (%5.1f KB): %6.2f us per exchange   timeouts %d  stale/wrong values %d\n", name, G, own,
                       want, want * 8 / 1024.0, t * 1e3 / NB, hb[0], hb[1]);
            }
        }
    }
}

int main() {
    double *buf, *sink; int *tags, *bad;
    hipMalloc(&buf, 8 << 20); hipMalloc(&sink, 8); hipMalloc(&tags, 32768); hipMalloc(&bad, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipStream_t st; hipStreamCreate(&st);
    run<0>("tags, one copy", buf, tags, bad, sink, st, e0, e1);
    run<1>("tags, a replica per XCD", buf, tags, bad, sink, st, e0, e1);
    run<2>("hierarchical counting barrier", buf, tags, bad, sink, st, e0, e1);
    return 0;
}
