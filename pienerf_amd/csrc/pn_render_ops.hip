// Ray-side kernels of the render path + the whole-frame driver (gfx950).
// Built with -ffp-contract=off (see pn_march_math.h).  Reference citations are relative to /root/reference.
#include <float.h>

#include "pn_march_window.h"

thread_local char pn_err_buf[512] = {0};

// Device-side record driving one loop trip of rund_cuda (nerf/renderer.py:836-891).
struct PnTrip {
    // written by the previous trip's compaction (trip 0: k_frame_rays), read-only while this trip's kernels run
    int n_alive;    // rays entering this trip
    int n_step;     // max(min(N // n_alive, 8), 1)
    int step_base;  // renderer's `step` before this trip
    int dense;      // see below
    int pad0[28];
    // counters the march updates with atomics: a cache line of their own, so that the waves reading the fields above (every wave's first
    // instruction) do not queue behind them
    int n_samples;  // entries of the sample list the network kernel reads (list trips: filled by atomics; dense trips: n_alive * n_step)
    int n_emitted;  // dense trips only: samples really emitted (statistics)
    int pad1[30];
};
static_assert(sizeof(PnTrip) == 256, "two cache lines");
// Dense trips (frame driver of the deformed render, every trip after the first): there nearly every alive ray fills all its n_step slots
// (measured on the chair: 98-99 %), so the sample list is the identity over the n_alive * n_step slots — written by the march without the
// returning atomic a compact list costs every wave (one more dependent memory round trip at the end of a latency-bound kernel: -24 % on
// trip 0's k_march without it) — and the few unfilled slots are zero-filled and run through the network as well; composite never reads
// them (their delta is 0).  The compaction kernel presets n_samples for such a trip.
__device__ __forceinline__ bool trip_is_dense(const PnTrip* t) { return t->dense != 0; }

// Ray groups (pn_render_opts::ray_batch > 0): the frame rendered "in ray batches of B" (max_ray_batch, get_opts.py:24; the staging loop of
// renderer.py:562-576) WITHOUT one launch chain per batch.  Rays are independent, so what a batch changes is only its own trip schedule: batch
// b = rays [b B, (b + 1) B) marches n_step_b = max(min(N_b // n_alive_b, 8), 1) samples per ray and trip, stops when none of ITS rays is alive or
// ITS step count reaches max_steps.  Stable compaction keeps the alive list sorted by ray id, so the batches are contiguous runs of it; every
// trip kernel handles all of them in one launch and looks up, per ray, its group's (first alive position, n_step, first sample slot).  One
// record per group and trip parity, written by the previous trip's compaction (trip 0: k_frame_rays).  The sample slots of a trip stay dense:
// slot_base is the running sum of n_alive_b * n_step_b.  n_step == 0 marks a group that ran into max_steps: composite retires its rays.
struct PnGroup {
    int alive_base, n_step, slot_base, step_base;
};
// n_step / first sample slot of the ray at alive position n (ray id `index`); groups == nullptr: one schedule for all rays (slot0 = n * n_step)
__device__ __forceinline__ void ray_slots(const PnGroup* __restrict__ groups, uint32_t group_rays, int index, uint32_t n, uint32_t& n_step, uint32_t& slot0) {
    if (groups) {
        const PnGroup g = groups[(uint32_t)index / group_rays];
        n_step = (uint32_t)g.n_step;
        slot0 = (uint32_t)g.slot_base + (n - (uint32_t)g.alive_base) * (uint32_t)g.n_step;
    } else {
        slot0 = n * n_step;
    }
}

// Per-frame device record of the frame drivers (pn_render_deformed / pn_render_static).
struct PnFrameDev {
    float aabb[6];      // bbmin = aabb, bbmax = aabb + 3   (aabb = cat(bbmin, bbmax), renderer.py:796)
    int resolution[4];  // [3] = n_grid
    int err;
    int unfinished;     // rays left alive by fixed-trip renders since the last reset, summed (staged batches are checked once per frame)
    int trips_run;      // loop trips the last render (or continuation) on this workspace has enqueued: written by its epilogue, so that it is
                        // also right after a HIP-graph REPLAY, which the host-side bookkeeping never sees
    int nb_alloc;       // candidate-list entries handed out so far (k_frame_prologue bumps it once per 32 cells; cleared by k_frame_tables)
    int fused_trips;    // trips the last k_trips_fused launch ran (pn_trips_fused.h); k_frame_finish adds them to trips_run and clears the field
    // summary of the trip records, written by k_frame_finish (the host reads this record instead of every trip's):
    int stat_trips;     // trips that had rays
    int alive_at_exit;  // rays alive behind the last trip enqueued
    int pad0;
    long long stat_samples;  // samples marched (dense trips: emitted; list trips: listed)
    // cells [ip_lo, ip_hi] per axis hold every integration point (k_frame_tables); with --cut the search grid spans +-bound (67^3 cells on the trex option
    // set) while the points fill a fortieth of it: k_frame_prologue builds candidate lists for the cells within one cell of that box only
    int ip_lo[3], ip_hi[3];
    int pad1[2];
};

// ------------------------------------------------------------------------------------------------ sph_from_ray
// kernel_sph_from_ray, raymarching.cu:165-202: where the ray leaves the sphere of `radius` (the larger root), as (theta, phi) scaled to [-1, 1] —
// the texture coordinate of the background model (renderer.py:246, bg_radius > 0).  atan2f / sqrtf of the device library, like the reference.
__global__ void __launch_bounds__(256) k_sph_from_ray(const float* __restrict__ rays_o, const float* __restrict__ rays_d, float radius, uint32_t N,
                                                      float* __restrict__ coords) {
    const uint32_t n = threadIdx.x + blockIdx.x * blockDim.x;
    if (n >= N) return;
    const float ox = rays_o[n * 3], oy = rays_o[n * 3 + 1], oz = rays_o[n * 3 + 2];
    const float dx = rays_d[n * 3], dy = rays_d[n * 3 + 1], dz = rays_d[n * 3 + 2];
    const float A = dx * dx + dy * dy + dz * dz;
    const float B = ox * dx + oy * dy + oz * dz;  // B / 2 of the quadratic
    const float Cq = ox * ox + oy * oy + oz * oz - radius * radius;
    const float t = (-B + sqrtf(B * B - A * Cq)) / A;
    const float x = ox + t * dx, y = oy + t * dy, z = oz + t * dz;
    const float theta = atan2f(sqrtf(x * x + z * z), y);  // y is up
    const float phi = atan2f(z, x);
    const float rpi = 0.3183098861837907f;
    coords[n * 2] = 2 * theta * rpi - 1;
    coords[n * 2 + 1] = phi * rpi;
}

extern "C" int pn_sph_from_ray(const float* rays_o, const float* rays_d, float radius, uint32_t N, float* coords, void* stream) {
    if (N == 0) return PN_OK;
    PN_REQUIRE(rays_o && rays_d && coords);
    k_sph_from_ray<<<pn_div_up(N, 256), 256, 0, (hipStream_t)stream>>>(rays_o, rays_d, radius, N, coords);
    PN_LAUNCH_CHECK();
    return PN_OK;
}

// ------------------------------------------------------------------------------------------------ near/far
// kernel_near_far_from_aabb, raymarching.cu:91-159
__global__ void __launch_bounds__(256) k_near_far(const float* __restrict__ rays_o, const float* __restrict__ rays_d, const float* __restrict__ aabb,
                                                  uint32_t N, float min_near, float* __restrict__ nears, float* __restrict__ fars,
                                                  float* __restrict__ rays_t) {
    const uint32_t n = threadIdx.x + blockIdx.x * blockDim.x;
    if (n >= N) return;
    const float ox = rays_o[n * 3], oy = rays_o[n * 3 + 1], oz = rays_o[n * 3 + 2];
    const float dx = rays_d[n * 3], dy = rays_d[n * 3 + 1], dz = rays_d[n * 3 + 2];
    const float rdx = 1 / dx, rdy = 1 / dy, rdz = 1 / dz;
    float near = (aabb[0] - ox) * rdx, far = (aabb[3] - ox) * rdx;
    if (near > far) { float c = near; near = far; far = c; }
    float near_y = (aabb[1] - oy) * rdy, far_y = (aabb[4] - oy) * rdy;
    if (near_y > far_y) { float c = near_y; near_y = far_y; far_y = c; }
    bool miss = (near > far_y || near_y > far);
    if (!miss) {
        if (near_y > near) near = near_y;
        if (far_y < far) far = far_y;
        float near_z = (aabb[2] - oz) * rdz, far_z = (aabb[5] - oz) * rdz;
        if (near_z > far_z) { float c = near_z; near_z = far_z; far_z = c; }
        miss = (near > far_z || near_z > far);
        if (!miss) {
            if (near_z > near) near = near_z;
            if (far_z < far) far = far_z;
            if (near < min_near) near = min_near;
        }
    }
    if (miss) near = far = FLT_MAX;
    nears[n] = near;
    fars[n] = far;
    if (rays_t) rays_t[n] = near;  // frame driver: rays_t = nears.clone() (renderer.py:829)
}

extern "C" int pn_near_far_from_aabb(const float* rays_o, const float* rays_d, const float* aabb, uint32_t N, float min_near, float* nears,
                                     float* fars, void* stream) {
    if (N == 0) return PN_OK;  // empty tensors have null data pointers
    PN_REQUIRE(rays_o && rays_d && aabb && nears && fars);
    k_near_far<<<pn_div_up(N, 256), 256, 0, (hipStream_t)stream>>>(rays_o, rays_d, aabb, N, min_near, nears, fars, nullptr);
    PN_LAUNCH_CHECK();
    return PN_OK;
}

// ------------------------------------------------------------------------------------------------ get_rays
// nerf/utils.py:54-138 (N = -1): pixel p -> (i = p%W + .5, j = p/W + .5).  pose: device pointer, row-major 4x4 cam2world
// (the reference's `poses` is a device tensor too, so no host round trip is needed).
__global__ void __launch_bounds__(256) k_get_rays(const float* __restrict__ pose, float fx, float fy, float cx, float cy, int HW, int W,
                                                  float* __restrict__ rays_o, float* __restrict__ rays_d) {
    const int p = threadIdx.x + blockIdx.x * blockDim.x;
    if (p >= HW) return;
    const float i = (float)(p % W) + 0.5f, j = (float)(p / W) + 0.5f;
    const float xs = (i - cx) / fx, ys = (j - cy) / fy, zs = 1.0f;
    const float nrm = sqrtf(xs * xs + ys * ys + zs * zs);
    const float d0 = xs / nrm, d1 = ys / nrm, d2 = zs / nrm;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        rays_d[p * 3 + c] = d0 * pose[c * 4] + d1 * pose[c * 4 + 1] + d2 * pose[c * 4 + 2];
        rays_o[p * 3 + c] = pose[c * 4 + 3];
    }
}

extern "C" int pn_get_rays(const float* pose, float fx, float fy, float cx, float cy, int H, int W, float* rays_o, float* rays_d, void* stream) {
    PN_REQUIRE(pose && rays_o && rays_d && H > 0 && W > 0);
    k_get_rays<<<pn_div_up((uint64_t)H * W, 256), 256, 0, (hipStream_t)stream>>>(pose, fx, fy, cx, cy, H * W, W, rays_o, rays_d);
    PN_LAUNCH_CHECK();
    return PN_OK;
}

// ------------------------------------------------------------------------------------------------ spatial hash of IPs
// p2g, nerf/utils.py:389-407
__device__ __forceinline__ int p2g(const float* __restrict__ p, const float* __restrict__ bbmin, float hgs, const int* __restrict__ res, int n_grid) {
    const int g0 = (int)floorf((p[0] - bbmin[0]) / hgs);
    const int g1 = (int)floorf((p[1] - bbmin[1]) / hgs);
    const int g2 = (int)floorf((p[2] - bbmin[2]) / hgs);
    const int gid = g2 * res[1] * res[0] + g1 * res[0] + g0;
    return (gid < 0 || gid >= n_grid) ? -1 : gid;
}

__global__ void __launch_bounds__(256) k_pig_zero(int* __restrict__ cnt, int n_grid_max, const int* __restrict__ n_grid_dev) {
    const int n_grid = n_grid_dev ? min(*n_grid_dev, n_grid_max) : n_grid_max;
    for (int g = threadIdx.x + blockIdx.x * blockDim.x; g < n_grid; g += gridDim.x * blockDim.x) cnt[g] = 0;
}

// get_pig_cnt, nerf/utils.py:410-424
__global__ void __launch_bounds__(256) k_pig_count(int n_vtx, int n_grid_max, const int* __restrict__ n_grid_dev, const float* __restrict__ pnts,
                                                   const float* __restrict__ bbmin, float hgs, const int* __restrict__ res, int* cnt,
                                                   int* err_flag) {
    const int p = threadIdx.x + blockIdx.x * blockDim.x;
    if (p >= n_vtx) return;
    const int n_grid = n_grid_dev ? min(*n_grid_dev, n_grid_max) : n_grid_max;
    const int gid = p2g(pnts + p * 3, bbmin, hgs, res, n_grid);
    if (gid >= 0) atomicAdd(cnt + gid, 1);
    else if (err_flag) atomicOr(err_flag, 2);
}

// pig_bgn = cumsum(cnt) - cnt (nerf/utils.py:369), one workgroup of 1024 threads, 16 cells per thread per tile (16 384 cells per round: a 300 k-cell
// grid — --cut with bound 2 — is 19 rounds of one barrier each; with 4 cells per thread and three barriers per round it was 74 rounds, 0.23 ms per scan).
// The running carry lives in a register of every thread (each adds the same 16 wave totals), the wave totals alternate between two LDS rows.
__device__ __forceinline__ void block_scan_1024(int n_grid, const int* __restrict__ cnt, int* __restrict__ bgn, int* __restrict__ cursor) {
    __shared__ int wsum[2][16];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    int carry = 0, buf = 0;
    for (int base = 0; base < n_grid; base += 16384, buf ^= 1) {
        const int i0 = base + threadIdx.x * 16;
        int v[16];
        if (i0 + 16 <= n_grid) {  // (cnt + i0 is 64-byte aligned: the tables come from hipMalloc and i0 is a multiple of 16)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int4 w = *reinterpret_cast<const int4*>(cnt + i0 + 4 * q);
                v[4 * q] = w.x; v[4 * q + 1] = w.y; v[4 * q + 2] = w.z; v[4 * q + 3] = w.w;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 16; k++) v[k] = (i0 + k < n_grid) ? cnt[i0 + k] : 0;
        }
        int tsum = 0;
#pragma unroll
        for (int k = 0; k < 16; k++) tsum += v[k];
        int inc = tsum;  // inclusive wave scan
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int u = __shfl_up(inc, o);
            if (lane >= o) inc += u;
        }
        if (lane == 63) wsum[buf][wid] = inc;
        __syncthreads();
        int woff = 0, total = 0;
#pragma unroll
        for (int w = 0; w < 16; w++) {
            const int x = wsum[buf][w];
            woff += (w < wid) ? x : 0;
            total += x;
        }
        int run = carry + woff + inc - tsum;
        if (i0 + 16 <= n_grid) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                int4 o4;
                o4.x = run; run += v[4 * q];
                o4.y = run; run += v[4 * q + 1];
                o4.z = run; run += v[4 * q + 2];
                o4.w = run; run += v[4 * q + 3];
                *reinterpret_cast<int4*>(bgn + i0 + 4 * q) = o4;
                *reinterpret_cast<int4*>(cursor + i0 + 4 * q) = o4;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 16; k++) {
                if (i0 + k < n_grid) { bgn[i0 + k] = run; cursor[i0 + k] = run; }
                run += v[k];
            }
        }
        carry += total;
    }
}
__global__ void __launch_bounds__(1024) k_pig_scan(int n_grid_max, const int* __restrict__ n_grid_dev, const int* __restrict__ cnt,
                                                   int* __restrict__ bgn, int* __restrict__ cursor) {
    const int n_grid = n_grid_dev ? min(*n_grid_dev, n_grid_max) : n_grid_max;
    block_scan_1024(n_grid, cnt, bgn, cursor);
}

// Large grids (--cut with bound 2: 300 k cells): the same exclusive scan in three launches over 4096-cell tiles — tile sums, one
// workgroup scanning the <= 1024 tile sums, per-tile scan + offset.  The tile's sum / offset travels in bgn[first cell of the tile],
// so no scratch buffer is needed.  (One workgroup walking 74 tiles one after the other took 0.23-0.30 ms per scan.)
__device__ __forceinline__ int block_sum_1024(int v, int* wsum) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if (lane == 0) wsum[wid] = v;
    __syncthreads();
    int total = 0;
    for (int w = 0; w < 16; w++) total += wsum[w];
    __syncthreads();
    return total;
}
__global__ void __launch_bounds__(1024) k_scan_tile_sum(int n_grid_max, const int* __restrict__ n_grid_dev, const int* __restrict__ cnt,
                                                        int* __restrict__ bgn) {
    __shared__ int wsum[16];
    const int n_grid = n_grid_dev ? min(*n_grid_dev, n_grid_max) : n_grid_max;
    const int base = (int)blockIdx.x * 4096;
    if (base >= n_grid) return;
    const int i0 = base + threadIdx.x * 4;
    int v = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) v += (i0 + k < n_grid) ? __hip_atomic_load(cnt + i0 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
    const int total = block_sum_1024(v, wsum);
    if (threadIdx.x == 0) bgn[base] = total;
}
__global__ void __launch_bounds__(1024) k_scan_tile_offsets(int n_grid_max, const int* __restrict__ n_grid_dev, int* __restrict__ bgn) {
    __shared__ int wsum[16];
    const int n_grid = n_grid_dev ? min(*n_grid_dev, n_grid_max) : n_grid_max;
    const int n_tiles = (n_grid + 4095) / 4096;  // <= 1024 (checked by the launcher)
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
    const int v = t < n_tiles ? __hip_atomic_load(bgn + (size_t)t * 4096, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int u = __shfl_up(inc, o);
        if (lane >= o) inc += u;
    }
    if (lane == 63) wsum[wid] = inc;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wid; w++) woff += wsum[w];
    if (t < n_tiles) bgn[(size_t)t * 4096] = woff + inc - v;
}
__global__ void __launch_bounds__(1024) k_scan_tile_apply(int n_grid_max, const int* __restrict__ n_grid_dev, const int* __restrict__ cnt,
                                                          int* __restrict__ bgn, int* __restrict__ cursor) {
    __shared__ int wsum[16];
    __shared__ int off_s;
    const int n_grid = n_grid_dev ? min(*n_grid_dev, n_grid_max) : n_grid_max;
    const int base = (int)blockIdx.x * 4096;
    if (base >= n_grid) return;
    if (threadIdx.x == 0) off_s = __hip_atomic_load(bgn + base, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int i0 = base + threadIdx.x * 4;
    int v[4];
#pragma unroll
    for (int k = 0; k < 4; k++) v[k] = (i0 + k < n_grid) ? __hip_atomic_load(cnt + i0 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
    const int tsum = v[0] + v[1] + v[2] + v[3];
    int inc = tsum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int u = __shfl_up(inc, o);
        if (lane >= o) inc += u;
    }
    if (lane == 63) wsum[wid] = inc;
    __syncthreads();  // also orders thread 0's read of the tile offset before any write to bgn[base]
    int woff = 0;
    for (int w = 0; w < wid; w++) woff += wsum[w];
    int run = off_s + woff + inc - tsum;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (i0 + k < n_grid) { bgn[i0 + k] = run; cursor[i0 + k] = run; }
        run += v[k];
    }
}
// exclusive scan cnt -> bgn, cursor over up to n_grid_max cells (the live count may come from device memory)
static void launch_cell_scan(int n_grid_max, const int* n_grid_dev, const int* cnt, int* bgn, int* cursor, hipStream_t st) {
    const int tiles = (int)pn_div_up(n_grid_max, 4096);
    // Measured on the trex option set (300 k cells): the tiled form (three launches) shortens a single frame (1.04 vs 1.15 ms eager) and — since the
    // pipeline's rate is its lanes' chain latency (round 4: frames per second = lanes / latency of a lane's launches) — the pipelined step as well:
    // 1 464 -> 1 535 steps/s (profiles/r04_trex_scan.txt).  (Round 1 measured the opposite, 1.30 vs 1.13 ms per step, when a lane's chain was ~45
    // launches and the one long workgroup hid behind the other lanes.)  PN_TILED_SCAN=0: the one-workgroup scan.
    static const bool tiled = pn_env_u32("PN_TILED_SCAN", 1) != 0;
    if (!tiled || tiles <= 16 || tiles > 1024) {  // small grids: one workgroup is faster than three launches
        k_pig_scan<<<1, 1024, 0, st>>>(n_grid_max, n_grid_dev, cnt, bgn, cursor);
        return;
    }
    k_scan_tile_sum<<<tiles, 1024, 0, st>>>(n_grid_max, n_grid_dev, cnt, bgn);
    k_scan_tile_offsets<<<1, 1024, 0, st>>>(n_grid_max, n_grid_dev, bgn);
    k_scan_tile_apply<<<tiles, 1024, 0, st>>>(n_grid_max, n_grid_dev, cnt, bgn, cursor);
}

// get_pig_idx, nerf/utils.py:427-443 — slots claimed through a per-cell cursor ...
__global__ void __launch_bounds__(256) k_pig_fill(int n_vtx, int n_grid_max, const int* __restrict__ n_grid_dev, const float* __restrict__ pnts,
                                                  const float* __restrict__ bbmin, float hgs, const int* __restrict__ res, int* cursor,
                                                  int* __restrict__ idx) {
    const int p = threadIdx.x + blockIdx.x * blockDim.x;
    if (p >= n_vtx) return;
    const int n_grid = n_grid_dev ? min(*n_grid_dev, n_grid_max) : n_grid_max;
    const int gid = p2g(pnts + p * 3, bbmin, hgs, res, n_grid);
    if (gid >= 0) idx[atomicAdd(cursor + gid, 1)] = p;
}
// ... then each cell's few entries are put in ascending point id, which makes the table independent of atomic order.
__global__ void __launch_bounds__(256) k_pig_sort(int n_grid_max, const int* __restrict__ n_grid_dev, const int* __restrict__ cnt,
                                                  const int* __restrict__ bgn, int* __restrict__ idx) {
    const int n_grid = n_grid_dev ? min(*n_grid_dev, n_grid_max) : n_grid_max;
    for (int g = threadIdx.x + blockIdx.x * blockDim.x; g < n_grid; g += gridDim.x * blockDim.x) {
        const int c = __hip_atomic_load(cnt + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (c < 2) continue;
        int* a = idx + bgn[g];
        for (int i = 1; i < c; i++) {
            const int v = __hip_atomic_load(a + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int j = i - 1;
            while (j >= 0) {
                const int u = __hip_atomic_load(a + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (u <= v) break;
                __hip_atomic_store(a + j + 1, u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                j--;
            }
            __hip_atomic_store(a + j + 1, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

static int pig_build(int n_vtx, int n_grid_max, const int* n_grid_dev, const float* pnts, const float* bbmin, float hgs, const int* res, int* cnt,
                     int* bgn, int* idx, int* cursor, int* err_flag, hipStream_t st) {
    const int gz = (int)pn_div_up(n_grid_max, 256) < 1024 ? (int)pn_div_up(n_grid_max, 256) : 1024;
    k_pig_zero<<<gz, 256, 0, st>>>(cnt, n_grid_max, n_grid_dev);
    k_pig_count<<<pn_div_up(n_vtx, 256), 256, 0, st>>>(n_vtx, n_grid_max, n_grid_dev, pnts, bbmin, hgs, res, cnt, err_flag);
    launch_cell_scan(n_grid_max, n_grid_dev, cnt, bgn, cursor, st);
    k_pig_fill<<<pn_div_up(n_vtx, 256), 256, 0, st>>>(n_vtx, n_grid_max, n_grid_dev, pnts, bbmin, hgs, res, cursor, idx);
    k_pig_sort<<<gz, 256, 0, st>>>(n_grid_max, n_grid_dev, cnt, bgn, idx);
    PN_LAUNCH_CHECK();
    return PN_OK;
}

extern "C" int pn_pnts_in_grids(int n_vtx, int n_grid, const float* pnts, const float* bbmin, float hgs, const int* resolution, int* pig_cnt,
                                int* pig_bgn, int* pig_idx, int* err_flag, void* stream) {
    PN_REQUIRE(n_vtx > 0 && n_grid > 0 && pnts && bbmin && resolution && pig_cnt && pig_bgn && pig_idx);
    int* cursor = nullptr;
    hipStream_t st = (hipStream_t)stream;
    PN_HIP_CHECK(hipMallocAsync((void**)&cursor, sizeof(int) * (size_t)n_grid, st));
    int rc = pig_build(n_vtx, n_grid, nullptr, pnts, bbmin, hgs, resolution, pig_cnt, pig_bgn, pig_idx, cursor, err_flag, st);
    PN_HIP_CHECK(hipFreeAsync(cursor, st));
    return rc;
}

// ------------------------------------------------------------------------------------------------ march
// Side tables of the cooperative march (pn_march_tables.h): per-cell candidate lists and packed IP records.
__device__ __forceinline__ void nb_cell_coords(int c, int r0, int r1, int& g0, int& g1, int& g2) {
    g0 = c % r0;
    g1 = (c / r0) % r1;
    g2 = c / (r0 * r1);
}
// neighbour k of cell (g0,g1,g2) in the visiting order of find_closest_IPs (offset applied as (g0+a, g1+b, g2+c), raymarching.cu:1095-1102)
// or, for num_seek_IP == 1, of find_closest_IP (offset applied as (g2+a, g1+b, g0+c), :1018-1025).  Returns -1 when out of the grid.
__device__ __forceinline__ int nb_neighbour(int k, int swap, int g0, int g1, int g2, int r0, int r1, int r2) {
    const int a = pnm::NBR26[k][0], b = pnm::NBR26[k][1], c = pnm::NBR26[k][2];
    const int n0 = g0 + (swap ? c : a), n1 = g1 + b, n2 = g2 + (swap ? a : c);
    if (n0 < 0 || n0 >= r0 || n1 < 0 || n1 >= r1 || n2 < 0 || n2 >= r2) return -1;
    return n2 * r1 * r0 + n1 * r0 + n0;
}

__global__ void __launch_bounds__(256) k_nb_count(int n_grid_max, const int* __restrict__ n_grid_dev, const int* __restrict__ res,
                                                  const int* __restrict__ pig_cnt, int swap, int* __restrict__ nb_cnt) {
    const int n_grid = n_grid_dev ? min(*n_grid_dev, n_grid_max) : n_grid_max;
    const int r0 = res[0], r1 = res[1], r2 = res[2];
    for (int c = threadIdx.x + blockIdx.x * blockDim.x; c < n_grid; c += gridDim.x * blockDim.x) {
        int g0, g1, g2;
        nb_cell_coords(c, r0, r1, g0, g1, g2);
        int s = pig_cnt[c];
        for (int k = 0; k < 26; k++) {
            const int nbc = nb_neighbour(k, swap, g0, g1, g2, r0, r1, r2);
            if (nbc >= 0) s += pig_cnt[nbc];
        }
        nb_cnt[c] = s;
    }
}

__global__ void __launch_bounds__(256) k_nb_fill(int n_grid_max, const int* __restrict__ n_grid_dev, const int* __restrict__ res,
                                                 const int* __restrict__ pig_cnt, const int* __restrict__ pig_bgn, const int* __restrict__ pig_idx,
                                                 const float* __restrict__ p_def, int swap, const int* __restrict__ nb_cnt, const int* __restrict__ nb_bgn,
                                                 float4* __restrict__ nb, int nb_capacity, int* err_flag, int2* __restrict__ nb_rng) {
    const int n_grid = n_grid_dev ? min(*n_grid_dev, n_grid_max) : n_grid_max;
    const int r0 = res[0], r1 = res[1], r2 = res[2];
    for (int c = threadIdx.x + blockIdx.x * blockDim.x; c < n_grid; c += gridDim.x * blockDim.x) {
        int w = nb_bgn[c];
        const bool fits = w + nb_cnt[c] <= nb_capacity;
        nb_rng[c] = fits ? make_int2(w, w + nb_cnt[c]) : make_int2(0, 0);
        if (nb_cnt[c] == 0) continue;
        if (!fits) { if (err_flag) atomicOr(err_flag, 8); continue; }
        int g0, g1, g2;
        nb_cell_coords(c, r0, r1, g0, g1, g2);
        for (int k = -1; k < 26; k++) {
            const int cell = (k < 0) ? c : nb_neighbour(k, swap, g0, g1, g2, r0, r1, r2);
            if (cell < 0) continue;
            const int n = pig_cnt[cell], b = pig_bgn[cell];
            for (int i = 0; i < n; i++) {
                const int ip = pig_idx[b + i];
                nb[w++] = make_float4(p_def[ip * 3], p_def[ip * 3 + 1], p_def[ip * 3 + 2], __int_as_float(ip));
            }
        }
    }
}

// rec[ip]: see pn_march_tables.h (pack_ip_float)
__global__ void __launch_bounds__(256) k_pack_ip(int n_vtx, const float* __restrict__ p_ori, const float* __restrict__ p_def,
                                                 const float* __restrict__ F_IP, const float* __restrict__ dF_IP, float* __restrict__ rec) {
    const int t = threadIdx.x + blockIdx.x * blockDim.x;
    const int ip = t / PN_REC_FLOATS, j = t % PN_REC_FLOATS;
    if (ip >= n_vtx) return;
    rec[t] = pnm2::pack_ip_float(j, ip, p_ori, p_def, F_IP, dF_IP);
}

struct MarchSide {  // device buffers of the side tables
    int *nb_cnt, *nb_bgn, *nb_cursor;  // [n_grid_max + 1] op-level build only (count -> scan -> fill); the frame driver allocates list space by bumping a counter
    int2* nb_rng;                       // [n_grid_max]
    float4* nb;                         // [nb_capacity]
    float* rec;                         // [n_vtx * 44]
    int nb_capacity;
};

static int march_side_build(const MarchSide& s, int n_vtx, int n_grid_max, const int* n_grid_dev, const int* res, const int* pig_cnt,
                            const int* pig_bgn, const int* pig_idx, const float* p_def, const float* p_ori, const float* F_IP, const float* dF_IP,
                            int num_seek_IP, int* err_flag, hipStream_t st) {
    const int swap = (num_seek_IP == 1) ? 1 : 0;
    const int gz = (int)pn_div_up(n_grid_max, 256) < 1024 ? (int)pn_div_up(n_grid_max, 256) : 1024;
    k_nb_count<<<gz, 256, 0, st>>>(n_grid_max, n_grid_dev, res, pig_cnt, swap, s.nb_cnt);
    launch_cell_scan(n_grid_max, n_grid_dev, s.nb_cnt, s.nb_bgn, s.nb_cursor, st);
    k_nb_fill<<<gz, 256, 0, st>>>(n_grid_max, n_grid_dev, res, pig_cnt, pig_bgn, pig_idx, p_def, swap, s.nb_cnt, s.nb_bgn, s.nb, s.nb_capacity, err_flag, s.nb_rng);
    k_pack_ip<<<pn_div_up((uint64_t)n_vtx * PN_REC_FLOATS, 256), 256, 0, st>>>(n_vtx, p_ori, p_def, F_IP, dF_IP, s.rec);
    PN_LAUNCH_CHECK();
    return PN_OK;
}

struct MarchIO {
    uint32_t n_alive, n_step;
    const int* rays_alive;
    float *xyzs, *dirs, *deltas;
    const float* noises;
    // frame-driver mode (trip != nullptr): counts come from device memory, valid sample slots are appended to `list`
    PnTrip* trip;
    int* list;
    float* t_resume;  // optional [n_alive]: written by k_march_skip, read by k_march (pn_march_tables.h: skip_empty_cells)
    // optional tail pass: rays unfinished after `max_rounds` windows in k_march are appended here (counters zeroed by the caller)
    struct TailEntry* tail;
    int* tail_counts;   // segmented (see PN_SEGS): rays with a long way to go, appended from the front of the segment's region
    int* tail_back;     // segmented: the others, appended from the back (the tail pass starts the long ones first)
    int* tail_cursors;  // segmented: next unprocessed entry (the tail pass hands its rays out dynamically)
    int tail_seg_cap;
    int max_rounds;
    // optional (trip 0 of the frame driver): k_march_skip lists the alive slots that still have something to march — nine rays in ten miss the
    // object's bounding box or run out of it inside the IP-free cells — and k_march walks that list instead of all n_alive slots
    int* active;
    int* active_counts;  // segmented
    int active_seg_cap;
    // frame-driver mode, list trips: the sample list is appended in segments (list_seg, samp_counts) and packed into `list` by k_list_pack;
    // dense trips: emit_parts collects the number of samples really emitted
    int* list_seg;
    int* samp_counts;
    int list_seg_cap;
    int* emit_parts;
    // optional: one bit per search cell, set when the cell has candidates (frame driver); k_march_skip keeps it in LDS when launched with
    // cell_bits_words * 4 bytes of dynamic shared memory
    const uint32_t* cell_bits;
    int cell_bits_words;
    // optional (with cell_bits): the cells within one cell of a cell with candidates, and where k_march_skip writes each ray's shortened end
    // (pn_march_window.h: ray_end_of_candidates); the march kernels then run with MarchParams::fars = fars_eff
    const uint32_t* cell_bits2;
    float* fars_eff;
    // optional (frame driver with ray groups, see PnGroup): this trip's group records
    const PnGroup* groups;
    uint32_t group_rays;
    int lane_per_ray;           // k_march: one lane per ray instead of eight (the throughput form of a frame's first trip)
    int dda_start, hop_budget;  // k_march_skip: restart the hop chain just before the first cell with candidates; hops before a ray is handed on (pn_march_window.h)
    // optional (--cut frames): the region map of pn_march_window.h (region_dda) — one bit per 8^3-voxel block of the top cascade level, set when a point of
    // the region can meet an occupied voxel on any level or the cut box (k_frame_prologue); k_march_skip keeps it in LDS
    const uint32_t* grid_regions;
    int grid_regions_words;
    int grid_regions_R;   // regions per axis: H / 8 (8^3-voxel regions) or H / 4
};

// Append lists are SEGMENTED: PN_SEGS independent (counter, region) pairs, every counter on a cache line of its own, the producer picking
// its segment from its workgroup / wave id.  Atomics on ONE address are served one at a time by the memory side — measured 11.4 ns each on
// gfx950, returning or not, however many waves issue them (tools/calib_atomic.hip: 5 000 waves x 1 atomic = +50 us) — and the march used one
// per wave (sample list) or per ray (tail list): that serialisation, not ALU work or memory latency, was 50-120 us of every march launch.
// Consumers need no prefix over the segments: workgroup b (wave w) takes segment b % PN_SEGS (w % PN_SEGS) and strides over its entries.
#define PN_SEGS 64
#define PN_SEG_STRIDE 32  // ints between counters: 128 B
__device__ __forceinline__ int seg_count(const int* counts, int seg) {
    return __hip_atomic_load(counts + seg * PN_SEG_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// entries a segment must hold when its producers are the workgroups (32 rays each) / waves (64 rays) with id % PN_SEGS == segment
// workers (workgroups or waves) whose id % PN_SEGS == seg, out of `total` (launches of segment consumers have at least PN_SEGS workers)
__device__ __forceinline__ int seg_workers(int total, int seg) { return max((total - seg + PN_SEGS - 1) / PN_SEGS, 1); }
// ... or, in the one-lane-per-ray form of k_march (G = 1), 256-ray chunks dealt by chunk % PN_SEGS: a segment then gets up to
// ceil(ceil(n / 256) / PN_SEGS) * 256 entries (640 000 rays: 10 240, more than the 64-ray form's 10 112 — round-3 advisor finding); the larger of the two
static uint32_t seg_cap_for(uint32_t n_rays) {
    const uint32_t by64 = (pn_div_up(pn_div_up(n_rays, 64), PN_SEGS) + 1) * 64, by256 = pn_div_up(pn_div_up(n_rays, 256), PN_SEGS) * 256 + 64;
    return std::max(by64, by256);
}

// One lane per ray slot: fast-forward over the leading run of IP-free search cells.
__global__ void __launch_bounds__(256) k_march_skip(pnm::MarchParams a, pnm2::March2Tables tb, MarchIO io) {
    extern __shared__ uint32_t bits_lds[];
    uint32_t n_alive = io.n_alive;
    if (io.trip) n_alive = (uint32_t)io.trip->n_alive;
    const uint32_t n = threadIdx.x + blockIdx.x * blockDim.x;
    const uint32_t *cell_bits = nullptr, *cell_bits2 = nullptr;
    if (io.cell_bits_words > 0) {  // uniform
        const int n_grid = a.resolution[0] * a.resolution[1] * a.resolution[2];
        const int words = min((n_grid + 31) >> 5, io.cell_bits_words);
        for (int w = threadIdx.x; w < words; w += blockDim.x) bits_lds[w] = io.cell_bits[w];
        cell_bits = bits_lds;
        if (io.cell_bits2 && io.fars_eff && !a.cut) {
            for (int w = threadIdx.x; w < words; w += blockDim.x) bits_lds[io.cell_bits_words + w] = io.cell_bits2[w];
            cell_bits2 = bits_lds + io.cell_bits_words;
        }
        __syncthreads();
    }
    const uint32_t* grid_regions = nullptr;
    if (io.grid_regions_words > 0) {  // uniform; behind the cell maps (the launch's dynamic LDS counts it in)
        uint32_t* gb = bits_lds + (io.cell_bits_words > 0 ? io.cell_bits_words * ((io.cell_bits2 && io.fars_eff && !a.cut) ? 2 : 1) : 0);
        for (int w = threadIdx.x; w < io.grid_regions_words; w += blockDim.x) gb[w] = io.grid_regions[w];
        grid_regions = gb;
        __syncthreads();
    }
    bool work = false;
    if (n < n_alive) {
        unsigned n_iter = 0;
        const int index = io.rays_alive[n];
        float far = a.fars[index];
        if (cell_bits2) {  // shorten the ray to where it can still find candidates
            const float near = a.rays_t[index];
            if (near < far) {
                const pnm3::Float3 o = *reinterpret_cast<const pnm3::Float3*>(a.rays_o + (size_t)index * 3),
                                   d = *reinterpret_cast<const pnm3::Float3*>(a.rays_d + (size_t)index * 3);
                far = pnm3::ray_end_of_candidates(a, cell_bits2, o.x, o.y, o.z, d.x, d.y, d.z, near, far);
            }
            io.fars_eff[index] = far;
        }
        const float t = pnm3::skip_empty_cells(a, tb, index, io.noises ? io.noises[n] : 0.0f, &n_iter, cell_bits, cell_bits2 ? far : -1.0f,
                                               io.dda_start ? cell_bits2 : nullptr, io.dda_start && cell_bits2 ? io.hop_budget : 0, grid_regions, io.grid_regions_R);
        io.t_resume[n] = t;
        if (!PN_DBG_PHASES_ON && a.stats && n_iter) atomicAdd(a.stats, (unsigned long long)n_iter);
        work = t < far;
        if (io.active && !work) {  // nothing left to march: k_march will not visit the slot, so its (single, n_step == 1) sample slot is ended here
            const uint32_t n_step = (uint32_t)io.trip->n_step;
            float* dl = io.deltas + (size_t)n * n_step * 2;
            for (uint32_t s2 = 0; s2 < n_step; s2++) { dl[2 * s2] = 0.0f; dl[2 * s2 + 1] = 0.0f; }
        }
    }
    if (io.active) {  // wave-aggregated append to this wave's segment (order is irrelevant: every listed slot is processed independently)
        const unsigned long long m = __ballot(work);
        const int lane = threadIdx.x & 63;
        const int seg = (int)((n >> 6) % PN_SEGS);
        int base = 0;
        if (lane == 0 && m) base = atomicAdd(io.active_counts + seg * PN_SEG_STRIDE, (int)__popcll(m));
        base = __shfl(base, 0);
        if (work) io.active[(size_t)seg * io.active_seg_cap + base + (int)__popcll(m & ((1ull << lane) - 1ull))] = (int)n;
    }
}

// ---- the per-ray march (pn_march_window.h): pass 1 = k_march (8 lanes per ray, bounded number of rounds), pass 2 = k_march_tail
// (one wave per ray that pass 1 left unfinished).
// A ray handed from k_march to k_march_tail, with everything the tail pass needs to go on: fetching the slot's ray through rays_alive ->
// rays_o / rays_d / fars again cost the tail three dependent memory round trips per ray — half of a typical tail ray's time (phase clocks).
struct __attribute__((aligned(16))) TailEntry {
    int n;            // alive slot
    float t, last_t;  // pnm3::RayState
    int step;
    float ox, oy, oz, dx;
    float dy, dz, rdx, rdy;
    float rdz, far;
    int slot0, n_step;  // first sample slot and sample budget of the ray in this trip (ray_slots)
};
static_assert(sizeof(TailEntry) == 64, "four 16-byte parts");

// waves per SIMD the march kernels ask for.  What really sets their occupancy is LDS: 12 KB of staging per wave (PN_STAGE_CAP) = three
// workgroups per CU, and the compiler then takes the registers three waves per SIMD leave it (~160 VGPRs, no spills).  One wave per SIMD is
// only 15 % slower for the march alone (a wave is a chain of dependent instructions and round trips), but what a march wave holds while it
// waits is what the other render lanes and the simulator cannot use (launch structure)
#ifndef PN_MARCH_WAVES
#define PN_MARCH_WAVES 4
#endif

// G = 8: 8 lanes per ray (each lane one point of the ray's t-sequence per round), 32 rays per 256-thread block.
// G = 1: ONE lane per ray, 256 rays per block — every evaluated point is a visited one (no speculation: a quarter of the VALU work per visited point of
// the windows, whose lanes evaluate 4.6 elements per voxel hop), at one visited point per round (the windows: ~14).  The throughput form of a frame's
// first trip (pn_render_opts.throughput): the wave-per-ray tail pass that the pipelined step is bound by only gets the rays that outlast the budget.
// WPB != 4 (G = 1 only): the PACKED form of the one-lane pass — WPB waves (12 KB of staging each, dynamic LDS) in ONE workgroup that takes a whole CU, the
// 64-ray chunks dealt per WAVE.  The first trip's ~1 100 busy waves then sit on ~90 CUs, three per SIMD, instead of one workgroup of 4 on every CU of
// the chip: a march wave is a chain of dependent instructions that leaves its SIMD idle four cycles in five, so three of them interleave almost for
// free — and the other frames' fused launches (one 157-KB workgroup per CU, which no CU with a march workgroup on it can take) find the rest of the
// chip free instead of waiting for the march to end.
#ifndef PN_MARCH_PACK_WAVES
#define PN_MARCH_PACK_WAVES 12
#endif
template <int K, bool MULTI, int G, int WPB = 4>
__global__ void __launch_bounds__(WPB * 64, WPB == 4 ? PN_MARCH_WAVES : 1) k_march(pnm::MarchParams a, pnm2::March2Tables tb, MarchIO io) {
    uint32_t n_alive = io.n_alive, n_step_trip = io.n_step;
    bool dense = false;
    if (io.trip) { n_alive = (uint32_t)io.trip->n_alive; n_step_trip = (uint32_t)io.trip->n_step; dense = trip_is_dense(io.trip); }
    static_assert(G == 8 || G == 1, "lanes per ray");
    constexpr bool WAVE_DEAL = WPB != 4;
    static_assert(!WAVE_DEAL || G == 1, "the packed form is the one-lane pass");
    constexpr uint32_t RB = WAVE_DEAL ? 64u : 256u / G;  // rays per chunk: a workgroup's (a wave's) share per step of its loop
    const int lane = threadIdx.x & 63, sub = lane & (G - 1), gbase = lane & ~(G - 1);
    const int budget = io.tail ? io.max_rounds : 0x7fffffff;
    __shared__ float4 stage_mem[WAVE_DEAL ? 1 : 4][WAVE_DEAL ? 1 : PN_STAGE_CAP];
    extern __shared__ __attribute__((aligned(16))) float4 stage_dyn[];   // WAVE_DEAL: WPB x PN_STAGE_CAP entries
    float4* stage = WAVE_DEAL ? stage_dyn + (size_t)(threadIdx.x >> 6) * PN_STAGE_CAP : stage_mem[WAVE_DEAL ? 0 : (threadIdx.x >> 6)];
    // 32-ray chunks are dealt round-robin to a bounded grid: in frame mode the alive count is only known on the device, and a
    // grid sized for all N rays would push ~20 000 mostly empty workgroups through the dispatcher on every trip.  With an active list
    // (trip 0) workgroup b walks segment b % PN_SEGS of it; either way `seg` names the segment this workgroup's own appends go to.
    // (WAVE_DEAL: read "wave" for "workgroup".)
    const uint32_t unit = WAVE_DEAL ? blockIdx.x * WPB + (threadIdx.x >> 6) : blockIdx.x, n_units = WAVE_DEAL ? gridDim.x * WPB : gridDim.x;
    const uint32_t act_seg = unit % PN_SEGS;
    const uint32_t n_work = io.active ? (uint32_t)seg_count(io.active_counts, (int)act_seg) : n_alive;
    const uint32_t k0 = io.active ? unit / PN_SEGS : unit, kstep = io.active ? (uint32_t)seg_workers((int)n_units, (int)act_seg) : n_units;
    PN_PHASE_DECL(pk);
    for (uint32_t chunk = k0; chunk * RB < n_work; chunk += kstep) {
        const uint32_t seg = io.active ? act_seg : chunk % PN_SEGS;
        const uint32_t i_work = chunk * RB + (WAVE_DEAL ? (uint32_t)lane : threadIdx.x / G);
        const uint32_t n = io.active ? (i_work < n_work ? (uint32_t)io.active[(size_t)act_seg * io.active_seg_cap + i_work] : 0xffffffffu) : i_work;
        uint32_t emitted = 0;
        bool deferred = false, have = false;
        float* dl = nullptr;
        pnm3::RayConsts c;
        pnm3::RayState st{0.f, 0.f, 0u};
        uint32_t n_step = n_step_trip, slot0 = 0;  // per ray with ray groups
        if (n < n_alive) {
            const int index = io.rays_alive[n];
            const float noise = io.noises ? io.noises[n] : 0.0f;
            ray_slots(io.groups, io.group_rays, index, n, n_step, slot0);
            dl = io.deltas + (size_t)slot0 * 2;
            pnm3::ray_consts(a, index, c);
            have = pnm3::ray_start(a, c, index, noise, io.t_resume ? io.t_resume + n : nullptr, st);
        }
        PN_PHASE(pk, 0);
        // all 64 lanes enter (the round loop inside is wave-uniform, pn_march_window.h); lanes without a ray idle through it
        const bool done = pnm3::march_window<K, MULTI, G>(a, tb, c, n_step, sub, gbase, lane, stage, io.xyzs + (size_t)slot0 * 3,
                                                          io.dirs + (size_t)slot0 * 3, dl, st, budget, have PN_PHASE_PASS);
        if (n < n_alive) {
            deferred = have && !done;  // still marching after the round budget: continue with a whole wave (k_march_tail)
            emitted = deferred ? 0u : st.step;  // a deferred ray's samples are listed by the tail pass
            if (!PN_DBG_PHASES_ON && a.stats && sub == 0 && emitted) atomicAdd(a.stats + 3, (unsigned long long)emitted);
        }
        if (io.tail) {  // one counter update per wave and class for all its deferred rays
            // class: more than three 64-element windows still to go (longest-first start order shortens the tail pass's critical path)
            const bool is_long = deferred && (c.far - st.t) > 192.0f * pnm3::dtf(a, c, st.t);
            const unsigned long long dm = __ballot(deferred && sub == 0), lm = __ballot(is_long && sub == 0), sm = dm & ~lm;
            if (dm) {
                int posl = 0, poss = 0;
                if (lane == 0 && lm) posl = atomicAdd(io.tail_counts + seg * PN_SEG_STRIDE, (int)__popcll(lm));
                if (lane == 0 && sm) poss = atomicAdd(io.tail_back + seg * PN_SEG_STRIDE, (int)__popcll(sm));
                posl = __shfl(posl, 0);
                poss = __shfl(poss, 0);
                if (deferred && sub < 4) {  // lanes 0..3 of the group write one 16-byte part each (G == 1: the lane writes all four)
                    const unsigned long long below = (1ull << gbase) - 1ull;
                    const int slot = is_long ? posl + (int)__popcll(lm & below) : io.tail_seg_cap - 1 - (poss + (int)__popcll(sm & below));
                    float4* te = reinterpret_cast<float4*>(io.tail + (size_t)seg * io.tail_seg_cap + slot);
#pragma unroll
                    for (int part_i = (G == 1 ? 0 : sub); part_i < (G == 1 ? 4 : sub + 1); part_i++) {
                        float4 part;
                        if (part_i == 0) part = make_float4(__int_as_float((int)n), st.t, st.last_t, __int_as_float((int)st.step));
                        else if (part_i == 1) part = make_float4(c.ox, c.oy, c.oz, c.dx);
                        else if (part_i == 2) part = make_float4(c.dy, c.dz, c.rdx, c.rdy);
                        else part = make_float4(c.rdz, c.far, __int_as_float((int)slot0), __int_as_float((int)n_step));
                        te[part_i] = part;
                    }
                }
            }
        }
        if (io.trip) {
            // slots the ray did not fill end it in composite (delta == 0); the op-level wrapper zero-fills instead (raymarching.py:415-417)
            if (dl && !deferred)
                for (uint32_t s = emitted + sub; s < n_step; s += G) { dl[2 * s] = 0.0f; dl[2 * s + 1] = 0.0f; }
            if (dense) {
                if (dl && !deferred) {
                    float* X = io.xyzs + (size_t)slot0 * 3;
                    float* Dd = io.dirs + (size_t)slot0 * 3;
                    for (uint32_t s = emitted + sub; s < n_step; s += G) { X[3 * s] = X[3 * s + 1] = X[3 * s + 2] = 0.0f; Dd[3 * s] = Dd[3 * s + 1] = Dd[3 * s + 2] = 0.0f; }
                    for (uint32_t s = sub; s < n_step; s += G) io.list[slot0 + s] = (int)(slot0 + s);
                }
                int v = (sub == 0 && dl && !deferred) ? (int)emitted : 0;  // one counter update per wave
                if (G == 1) { v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); }
                v += __shfl_xor(v, 8); v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
                if (lane == 0 && v) atomicAdd(io.emit_parts + seg * PN_SEG_STRIDE, v);
            } else {
            // wave-aggregated append of this wave's valid sample slots (one atomic per wave)
            int inc = (sub == 0) ? (int)emitted : 0;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int u = __shfl_up(inc, o);
                if (lane >= o) inc += u;
            }
            const int total = __shfl(inc, 63);
            int base = 0;
            if (lane == 63 && total > 0) base = atomicAdd(io.samp_counts + seg * PN_SEG_STRIDE, total);
            base = __shfl(base, 63);
            const int first = base + __shfl(inc, gbase) - (int)emitted;  // exclusive prefix of this group's first lane
            int* seg_list = io.list_seg + (size_t)seg * io.list_seg_cap;
            for (uint32_t s = sub; s < emitted; s += G) seg_list[first + s] = (int)(slot0 + s);
            }
        }
        PN_PHASE(pk, 5);
    }
    PN_PHASE_FLUSH(pk, a.stats, 0, lane);
}

// One wave per unfinished ray: windows of 64 sequence elements until the ray is done for this trip.
template <int K, bool MULTI>
__global__ void __launch_bounds__(256, PN_MARCH_WAVES) k_march_tail(pnm::MarchParams a, pnm2::March2Tables tb, MarchIO io) {
    bool dense = false;
    if (io.trip) dense = trip_is_dense(io.trip);
    const int lane = threadIdx.x & 63;
    const int gw = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6), n_waves = (int)gridDim.x * 4;
    const int seg = gw % PN_SEGS;  // this wave's segment of the tail list; its own appends go to the same segment of the sample list
    const int n_long = seg_count(io.tail_counts, seg), total = n_long + seg_count(io.tail_back, seg);
    __shared__ float4 stage_mem[4][PN_STAGE_CAP];
    float4* stage = stage_mem[threadIdx.x >> 6];
    PN_PHASE_DECL(pk);
    // the rays of a segment are handed out one at a time to the waves that serve it: their lengths differ by an order of magnitude (1 to 8
    // windows), and with a fixed assignment the wave that drew several long ones set the kernel's duration
    (void)n_waves;
    for (;;) {
        int e = 0;
        if (lane == 0) e = atomicAdd(io.tail_cursors + seg * PN_SEG_STRIDE, 1);
        e = __builtin_amdgcn_readfirstlane(e);
        if (e >= total) break;
        const TailEntry te = io.tail[(size_t)seg * io.tail_seg_cap + (e < n_long ? e : io.tail_seg_cap - 1 - (e - n_long))];
        const uint32_t slot0 = (uint32_t)te.slot0, n_step = (uint32_t)te.n_step;
        float* dl = io.deltas + (size_t)slot0 * 2;
        pnm3::RayConsts c;
        c.ox = te.ox; c.oy = te.oy; c.oz = te.oz; c.dx = te.dx; c.dy = te.dy; c.dz = te.dz; c.rdx = te.rdx; c.rdy = te.rdy; c.rdz = te.rdz; c.far = te.far;
        pnm3::frame_consts(a, c);
        pnm3::RayState st{te.t, te.last_t, (uint32_t)te.step};
        PN_PHASE(pk, 0);
        pnm3::march_window<K, MULTI, 64>(a, tb, c, n_step, lane, 0, lane, stage, io.xyzs + (size_t)slot0 * 3, io.dirs + (size_t)slot0 * 3,
                                         dl, st, 0x7fffffff, true PN_PHASE_PASS);
        const uint32_t emitted = st.step;
        if (!PN_DBG_PHASES_ON && a.stats && lane == 0 && emitted) atomicAdd(a.stats + 3, (unsigned long long)emitted);
        if (io.trip) {
            for (uint32_t s = emitted + lane; s < n_step; s += 64) { dl[2 * s] = 0.0f; dl[2 * s + 1] = 0.0f; }
            if (dense) {
                float* X = io.xyzs + (size_t)slot0 * 3;
                float* Dd = io.dirs + (size_t)slot0 * 3;
                for (uint32_t s = emitted + lane; s < n_step; s += 64) { X[3 * s] = X[3 * s + 1] = X[3 * s + 2] = 0.0f; Dd[3 * s] = Dd[3 * s + 1] = Dd[3 * s + 2] = 0.0f; }
                for (uint32_t s = lane; s < n_step; s += 64) io.list[slot0 + s] = (int)(slot0 + s);
                if (lane == 0 && emitted) atomicAdd(io.emit_parts + seg * PN_SEG_STRIDE, (int)emitted);
            } else {
                int base = 0;
                if (lane == 0 && emitted > 0) base = atomicAdd(io.samp_counts + seg * PN_SEG_STRIDE, (int)emitted);
                base = __shfl(base, 0);
                int* seg_list = io.list_seg + (size_t)seg * io.list_seg_cap;
                for (uint32_t s = lane; s < emitted; s += 64) seg_list[base + s] = (int)(slot0 + s);
            }
        }
        PN_PHASE(pk, 5);
    }
    PN_PHASE_FLUSH(pk, a.stats, 6, lane);
}

template <int K, bool MULTI>
static void launch_march_km(uint32_t blocks, uint32_t tail_blocks, hipStream_t st, const pnm::MarchParams& a, const pnm2::March2Tables& tb, const MarchIO& io) {
    // PN_MARCH_PACK=1 (experiments): the one-lane pass packed into whole-CU workgroups (see k_march, WPB).  Bit-identical, and measured neutral in the
    // pipeline on all three configurations (profiles/r05/march_pack_ab.txt), so the default stays four waves per workgroup all over the chip
    static const bool pack = pn_env_u32("PN_MARCH_PACK", 0) != 0;
    if (io.lane_per_ray && pack) {
        constexpr int W = PN_MARCH_PACK_WAVES;
        const size_t lds = (size_t)W * PN_STAGE_CAP * sizeof(float4);
        static bool granted[PN_MAX_DEVICES] = {false};  // dynamic LDS above 64 KB is opted into per function and DEVICE
        int dev_id = 0;
        if (hipGetDevice(&dev_id) == hipSuccess && (dev_id < 0 || dev_id >= PN_MAX_DEVICES || !granted[dev_id])) {
            (void)hipFuncSetAttribute((const void*)k_march<K, MULTI, 1, W>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (dev_id >= 0 && dev_id < PN_MAX_DEVICES) granted[dev_id] = true;
        }
        // as many waves as the unpacked grid had (blocks x 4), at least PN_SEGS of them, at most one workgroup per CU: beyond that the waves loop
        static const uint32_t pack_grid = pn_env_u32("PN_MARCH_PACK_GRID", 256);
        const uint32_t wgs = std::max(std::min(pn_div_up(blocks * 4u, (uint32_t)W), pack_grid), pn_div_up((uint32_t)PN_SEGS, (uint32_t)W));
        k_march<K, MULTI, 1, W><<<wgs, W * 64, lds, st>>>(a, tb, io);
    } else if (io.lane_per_ray) k_march<K, MULTI, 1><<<blocks, 256, 0, st>>>(a, tb, io);
    else k_march<K, MULTI, 8><<<blocks, 256, 0, st>>>(a, tb, io);
    if (io.tail) k_march_tail<K, MULTI><<<tail_blocks, 256, 0, st>>>(a, tb, io);
}

// pass 1 over `blocks` workgroups, then (io.tail != nullptr) the tail pass over `tail_blocks`
static void launch_march(int K, uint32_t blocks, uint32_t tail_blocks, hipStream_t st, const pnm::MarchParams& a, const pnm2::March2Tables& tb,
                         const MarchIO& io) {
    const bool multi = a.max_iter_num > 1;
    if (K == 1) { if (multi) launch_march_km<1, true>(blocks, tail_blocks, st, a, tb, io); else launch_march_km<1, false>(blocks, tail_blocks, st, a, tb, io); }
    else if (K == 2) { if (multi) launch_march_km<2, true>(blocks, tail_blocks, st, a, tb, io); else launch_march_km<2, false>(blocks, tail_blocks, st, a, tb, io); }
    else { if (multi) launch_march_km<3, true>(blocks, tail_blocks, st, a, tb, io); else launch_march_km<3, false>(blocks, tail_blocks, st, a, tb, io); }
}

// Rounds of 8 sequence elements a ray gets in k_march before it is handed to the wave-per-ray tail pass (PN_TAIL_ROUNDS overrides).
static int g_skip_dda_override = -1;    // pn_march_set_skip_dda (tests): 0 / 1 replace the default, -1: default (environment PN_SKIP_DDA)
static int g_tail_rounds_override = 0;  // pn_march_set_tail_rounds (tests): > 0 replaces the default below
// Defaults measured on the chair once the append lists were segmented (k_march + tail per trip, us): trip 0 (every ray looks for its first sample)
// 232 / 201 / 210 / 211 for 1 / 2 / 3 / 4 rounds; later trips (alive rays, 8 samples each: most are done after one window) 70 / 76 / 78 / 79.
static uint32_t march_tail_rounds(int trip = -1) {
    static const uint32_t r = pn_env_u32("PN_TAIL_ROUNDS", 0);  // 0: per-trip defaults
    if (g_tail_rounds_override > 0) return (uint32_t)g_tail_rounds_override;
    if (r) return r;
    return trip < 0 ? 4u : (trip == 0 ? 2u : 1u);
}
extern "C" int pn_march_set_skip_dda(int on) {
    PN_REQUIRE(on >= -1 && on <= 1);
    g_skip_dda_override = on;
    return PN_OK;
}
extern "C" int pn_march_set_tail_rounds(int rounds) {
    PN_REQUIRE(rounds >= 0);
    g_tail_rounds_override = rounds;
    return PN_OK;
}

static pnm::MarchParams make_march_params(const int* pig_cnt, const int* pig_bgn, const int* pig_idx, int n_vtx, int n_grid, const float* p_def,
                                          const float* p_ori, const float* F_IP, const float* dF_IP, int max_iter_num, const float* bbmin,
                                          const float* bbmax, float hgs, const int* resolution, int num_seek_IP, float IP_dx, int cut,
                                          const float* cut_bounds, const float* rays_t, const float* rays_o, const float* rays_d, float bound,
                                          float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H, const uint8_t* grid, const float* fars,
                                          int* err_flag) {
    pnm::MarchParams a;
    a.pig_cnt = pig_cnt; a.pig_bgn = pig_bgn; a.pig_idx = pig_idx; a.n_vtx = n_vtx; a.n_grid = n_grid;
    a.p_ori = p_ori; a.p_def = p_def; a.F_IP = F_IP; a.dF_IP = dF_IP; a.max_iter_num = max_iter_num;
    a.bbmin = bbmin; a.bbmax = bbmax; a.hgs = hgs; a.resolution = resolution; a.num_seek_IP = num_seek_IP; a.IP_dx = IP_dx;
    a.cut = cut; a.cut_bounds = cut_bounds; a.rays_t = rays_t; a.rays_o = rays_o; a.rays_d = rays_d;
    a.bound = bound; a.dt_gamma = dt_gamma; a.max_steps = max_steps; a.C = C; a.H = H; a.grid = grid; a.fars = fars; a.err_flag = err_flag;
    a.stats = nullptr;
    return a;
}

extern "C" int pn_march_rays_quadratic_bending(const int* pig_cnt, const int* pig_bgn, const int* pig_idx, int n_vtx, int n_grid,
                                               const float* p_def, const float* p_ori, const float* F_IP, const float* dF_IP, int max_iter_num,
                                               const float* bbmin, const float* bbmax, float hgs, const int* resolution, int num_seek_IP,
                                               float IP_dx, int cut, const float* cut_bounds, uint32_t n_alive, uint32_t n_step,
                                               const int* rays_alive, const float* rays_t, const float* rays_o, const float* rays_d, float bound,
                                               float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H, const uint8_t* grid, const float* nears,
                                               const float* fars, float* xyzs, float* dirs, float* deltas, const float* noises, int* err_flag,
                                               void* stream) {
    (void)nears;
    PN_REQUIRE(pig_cnt && pig_bgn && pig_idx && p_def && p_ori && F_IP && dF_IP && bbmin && bbmax && resolution);
    PN_REQUIRE(rays_alive && rays_t && rays_o && rays_d && grid && fars && xyzs && dirs && deltas);
    PN_REQUIRE(num_seek_IP >= 1 && num_seek_IP <= 3);
    PN_REQUIRE(!cut || cut_bounds);
    PN_REQUIRE(C >= 1 && C <= 8 && H > 0 && n_step >= 1 && n_vtx > 0 && n_grid > 0);
    if (n_alive == 0) return PN_OK;
    hipStream_t st = (hipStream_t)stream;
    // side tables are rebuilt from the caller's spatial hash on every call (stream-ordered pool allocations)
    MarchSide s;
    s.nb_capacity = 27 * n_vtx;
    char* pool = nullptr;
    const size_t ints = ((size_t)n_grid + 1) * 5 * sizeof(int), nbb = (size_t)s.nb_capacity * sizeof(float4), recb = (size_t)n_vtx * PN_REC_FLOATS * sizeof(float);
    const size_t off_nb = (ints + 255) & ~(size_t)255, off_rec = (off_nb + nbb + 255) & ~(size_t)255;
    const size_t off_res = (off_rec + recb + 255) & ~(size_t)255;
    const uint32_t tail_cap = seg_cap_for(n_alive);
    const size_t tail_ctr = (size_t)3 * PN_SEGS * PN_SEG_STRIDE * sizeof(int);  // front counters, back counters, cursors
    const size_t off_tail = (off_res + (size_t)n_alive * sizeof(float) + 255) & ~(size_t)255;  // [segment counters | tail entries]
    PN_HIP_CHECK(hipMallocAsync((void**)&pool, off_tail + tail_ctr + (size_t)PN_SEGS * tail_cap * sizeof(TailEntry), st));
    PN_HIP_CHECK(hipMemsetAsync(pool + off_tail, 0, tail_ctr, st));
    s.nb_cnt = (int*)pool; s.nb_bgn = s.nb_cnt + n_grid + 1; s.nb_cursor = s.nb_bgn + n_grid + 1; s.nb_rng = (int2*)(s.nb_cursor + n_grid + 1);
    s.nb = (float4*)(pool + off_nb); s.rec = (float*)(pool + off_rec);
    int rc = march_side_build(s, n_vtx, n_grid, nullptr, resolution, pig_cnt, pig_bgn, pig_idx, p_def, p_ori, F_IP, dF_IP, num_seek_IP, err_flag, st);
    if (rc == PN_OK) {
        pnm::MarchParams a = make_march_params(pig_cnt, pig_bgn, pig_idx, n_vtx, n_grid, p_def, p_ori, F_IP, dF_IP, max_iter_num, bbmin, bbmax, hgs,
                                               resolution, num_seek_IP, IP_dx, cut, cut_bounds, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps,
                                               C, H, grid, fars, err_flag);
        pnm2::March2Tables tb{s.nb_rng, s.nb, (const float4*)s.rec};
        MarchIO io{n_alive, n_step, rays_alive, xyzs, dirs, deltas, noises, nullptr, nullptr, (float*)(pool + off_res),
                   (TailEntry*)(pool + off_tail + tail_ctr), (int*)(pool + off_tail), (int*)(pool + off_tail) + PN_SEGS * PN_SEG_STRIDE,
                   (int*)(pool + off_tail) + 2 * PN_SEGS * PN_SEG_STRIDE, (int)tail_cap,
                   (int)march_tail_rounds(), nullptr, nullptr, 0,
                   nullptr, nullptr, 0, nullptr, nullptr, 0, nullptr, nullptr, nullptr, 0, 0};
        if (io.t_resume) k_march_skip<<<pn_div_up(n_alive, 256), 256, 0, st>>>(a, tb, io);
        launch_march(num_seek_IP, pn_div_up(n_alive, 32), std::max(std::min(pn_div_up(n_alive, 4), 2048u), (uint32_t)PN_SEGS / 4), st, a, tb, io);
    }
    PN_HIP_CHECK(hipFreeAsync(pool, st));
    if (rc) return rc;
    PN_LAUNCH_CHECK();
    return PN_OK;
}

// ------------------------------------------------------------------------------------------------ composite
// kernel_composite_rays, raymarching.cu:827-923.  __expf -> the gfx950 fast exponential (v_exp_f32 on x*log2e).
__device__ __forceinline__ bool composite_one(int index, uint32_t slot0, uint32_t n_step, float T_thresh, float* rays_t,
                                              const float* __restrict__ sigmas, const float* __restrict__ rgbs, const float* __restrict__ deltas,
                                              float* weights_sum, float* depth, float* image) {
    sigmas += (size_t)slot0;
    rgbs += (size_t)slot0 * 3;
    deltas += (size_t)slot0 * 2;
    float t = rays_t[index];
    float ws = weights_sum[index], d = depth[index];
    float r = image[index * 3], g = image[index * 3 + 1], b = image[index * 3 + 2];
    uint32_t step = 0;
    while (step < n_step) {
        if (deltas[0] == 0) break;
        const float alpha = 1.0f - __expf(-sigmas[0] * deltas[0]);
        const float T = 1 - ws;
        const float w = alpha * T;
        ws += w;
        t += deltas[1];
        d += w * t;
        r += w * rgbs[0];
        g += w * rgbs[1];
        b += w * rgbs[2];
        if (T < T_thresh) break;
        sigmas++; rgbs += 3; deltas += 2; step++;
    }
    const bool alive = !(step < n_step);
    if (alive) rays_t[index] = t;  // (the caller marks a dead ray in rays_alive)
    weights_sum[index] = ws;
    depth[index] = d;
    image[index * 3] = r; image[index * 3 + 1] = g; image[index * 3 + 2] = b;
    return alive;
}

// One 256-ray chunk per block; in frame-driver mode also records the chunk's survivor count for the compaction pass.
// groups / group_cnt (ray groups, see PnGroup): per-ray schedule, and the survivors counted per group — the alive list is sorted by ray id, so the
// lanes of a wave form a few runs of equal group id and every run costs one atomic (group_cnt == nullptr with a single group: its count is the
// chunk total the compaction computes anyway, and one counter for every wave of the launch would serialise, see PN_SEGS).
__global__ void __launch_bounds__(256) k_composite(uint32_t n_alive_arg, uint32_t n_step_arg, float T_thresh, int* rays_alive, float* rays_t,
                                                   const float* __restrict__ sigmas, const float* __restrict__ rgbs,
                                                   const float* __restrict__ deltas, float* weights_sum, float* depth, float* image,
                                                   const PnTrip* trip, int* chunk_counts, const PnGroup* __restrict__ groups, uint32_t group_rays,
                                                   int* group_cnt) {
    uint32_t n_alive = n_alive_arg, n_step_trip = n_step_arg;
    if (trip) { n_alive = (uint32_t)trip->n_alive; n_step_trip = (uint32_t)trip->n_step; }
    for (uint32_t chunk = blockIdx.x; chunk * 256u < n_alive; chunk += gridDim.x) {  // bounded grid, see k_march
        const uint32_t n = threadIdx.x + chunk * 256u;
        bool alive = false;
        int grp = -1;
        if (n < n_alive) {
            const int index = rays_alive[n];
            uint32_t n_step = n_step_trip, slot0;
            ray_slots(groups, group_rays, index, n, n_step, slot0);
            if (groups) grp = (int)((uint32_t)index / group_rays);
            // n_step == 0: its group has reached max_steps: the batch's loop is over (renderer.py:836), the ray is dropped
            if (n_step != 0) alive = composite_one(index, slot0, n_step, T_thresh, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image);
            if (!alive) rays_alive[n] = -1;
        }
        if (group_cnt) {
            const int lane = threadIdx.x & 63;
            const unsigned long long am = __ballot(alive);
            const int prev = __shfl_up(grp, 1);
            const bool head = lane == 0 || grp != prev;
            const unsigned long long hm = __ballot(head);
            if (head && grp >= 0) {  // this run: lanes [lane, next head)
                const unsigned long long above = lane == 63 ? 0ull : hm & ~((2ull << lane) - 1ull);
                const unsigned long long upto = above ? ((1ull << (__ffsll((long long)above) - 1)) - 1ull) : ~0ull;
                const int c = (int)__popcll(am & upto & ~((1ull << lane) - 1ull));
                if (c) atomicAdd(group_cnt + grp, c);
            }
        }
        if (chunk_counts) {
            const int c = __syncthreads_count(alive);
            if (threadIdx.x == 0) chunk_counts[chunk] = c;
        }
    }
}

extern "C" int pn_composite_rays(uint32_t n_alive, uint32_t n_step, float T_thresh, int* rays_alive, float* rays_t, const float* sigmas,
                                 const float* rgbs, const float* deltas, float* weights_sum, float* depth, float* image, void* stream) {
    if (n_alive == 0) return PN_OK;  // empty tensors have null data pointers
    PN_REQUIRE(rays_alive && rays_t && sigmas && rgbs && deltas && weights_sum && depth && image && n_step >= 1);
    k_composite<<<pn_div_up(n_alive, 256), 256, 0, (hipStream_t)stream>>>(n_alive, n_step, T_thresh, rays_alive, rays_t, sigmas, rgbs, deltas,
                                                                         weights_sum, depth, image, nullptr, nullptr, nullptr, 0, nullptr);
    PN_LAUNCH_CHECK();
    return PN_OK;
}

// ------------------------------------------------------------------------------------------------ static (undeformed) inference ops
// SURVEY 8(f) rank 3, inference side: kernel_march_rays (raymarching.cu:703-810), kernel_packbits (:270-292), kernel_morton3D /
// kernel_morton3D_invert (:217-258).  Off the simulate-and-render hot path (the deformed march above replaces kernel_march_rays
// there): one lane per ray / byte / index like the reference, arithmetic restated literally (this file is compiled with
// -ffp-contract=off) so that samples are bit-identical to the oracle.
__global__ void __launch_bounds__(128) k_march_rays_static(uint32_t n_alive, uint32_t n_step, const int* __restrict__ rays_alive,
                                                           const float* __restrict__ rays_t, const float* __restrict__ rays_o,
                                                           const float* __restrict__ rays_d, float bound, float dt_gamma, uint32_t max_steps, uint32_t C,
                                                           uint32_t H, const uint8_t* __restrict__ grid, const float* __restrict__ fars,
                                                           float* __restrict__ xyzs, float* __restrict__ dirs, float* __restrict__ deltas,
                                                           const float* __restrict__ noises) {
    using namespace pnm;
    const uint32_t n = threadIdx.x + blockIdx.x * blockDim.x;
    if (n >= n_alive) return;
    const int index = rays_alive[n];
    const float noise = noises ? noises[n] : 0.0f;
    rays_o += (size_t)index * 3;
    rays_d += (size_t)index * 3;
    xyzs += (size_t)n * n_step * 3;
    dirs += (size_t)n * n_step * 3;
    deltas += (size_t)n * n_step * 2;
    const float ox = rays_o[0], oy = rays_o[1], oz = rays_o[2];
    const float dx = rays_d[0], dy = rays_d[1], dz = rays_d[2];
    const float rdx = 1 / dx, rdy = 1 / dy, rdz = 1 / dz;
    const float rH = 1 / (float)H;
    const float H3 = (float)(H * H * H);
    float t = rays_t[index];
    const float far = fars[index];
    const float dt_min = 2 * 1.73205080757f / max_steps;
    const float dt_max = 2 * 1.73205080757f * (1 << (C - 1)) / H;
    uint32_t step = 0;
    t += clampf(t * dt_gamma, dt_min, dt_max) * noise;
    float last_t = t;
    while (t < far && step < n_step) {
        const float x = clampf(ox + t * dx, -bound, bound);
        const float y = clampf(oy + t * dy, -bound, bound);
        const float z = clampf(oz + t * dz, -bound, bound);
        const float dt = clampf(t * dt_gamma, dt_min, dt_max);
        const int level = max(mip_from_pos(x, y, z, (float)C), mip_from_dt(dt, (float)H, (float)C));
        const float mip_bound = fminf(scalbnf(1, level), bound);
        const float mip_rbound = 1 / mip_bound;
        // `0.5 * (x * mip_rbound + 1) * H` is a double product in the reference; (float)(0.5 * (double)v * (double)H) == v * (0.5f * H)
        // for every float v and power-of-two-free H < 2^24 only when the product is exact, so it is kept in double here (cold path)
        const int nx = (int)clampf((float)(0.5 * (double)(x * mip_rbound + 1) * (double)H), 0.0f, (float)(H - 1));
        const int ny = (int)clampf((float)(0.5 * (double)(y * mip_rbound + 1) * (double)H), 0.0f, (float)(H - 1));
        const int nz = (int)clampf((float)(0.5 * (double)(z * mip_rbound + 1) * (double)H), 0.0f, (float)(H - 1));
        const uint32_t vox = (uint32_t)(level * H3 + (float)morton3D(nx, ny, nz));
        const bool occ = grid[vox / 8] & (1 << (vox % 8));
        if (occ) {
            xyzs[0] = x; xyzs[1] = y; xyzs[2] = z;
            dirs[0] = dx; dirs[1] = dy; dirs[2] = dz;
            t += dt;
            deltas[0] = dt;
            deltas[1] = t - last_t;
            last_t = t;
            xyzs += 3; dirs += 3; deltas += 2;
            step++;
        } else {
            const float tx = (((nx + 0.5f + 0.5f * signf(dx)) * rH * 2 - 1) * mip_bound - x) * rdx;
            const float ty = (((ny + 0.5f + 0.5f * signf(dy)) * rH * 2 - 1) * mip_bound - y) * rdy;
            const float tz = (((nz + 0.5f + 0.5f * signf(dz)) * rH * 2 - 1) * mip_bound - z) * rdz;
            const float tt = t + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
            do { t += clampf(t * dt_gamma, dt_min, dt_max); } while (t < tt);
        }
    }
}

extern "C" int pn_march_rays(uint32_t n_alive, uint32_t n_step, const int* rays_alive, const float* rays_t, const float* rays_o, const float* rays_d,
                             float bound, float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H, const uint8_t* grid, const float* nears,
                             const float* fars, float* xyzs, float* dirs, float* deltas, const float* noises, void* stream) {
    (void)nears;
    if (n_alive == 0) return PN_OK;
    PN_REQUIRE(rays_alive && rays_t && rays_o && rays_d && grid && fars && xyzs && dirs && deltas);
    PN_REQUIRE(C >= 1 && C <= 8 && H > 0 && n_step >= 1 && max_steps > 0);
    k_march_rays_static<<<pn_div_up(n_alive, 128), 128, 0, (hipStream_t)stream>>>(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, dt_gamma,
                                                                                 max_steps, C, H, grid, fars, xyzs, dirs, deltas, noises);
    PN_LAUNCH_CHECK();
    return PN_OK;
}

// Frame-driver form of the static march (pn_render_static): counts come from the trip record, 256-ray chunks are dealt round-robin to a
// bounded grid, unfilled slots are ended (delta = 0) and the valid sample slots are appended to `list` (one atomic per wave).
__device__ __forceinline__ uint32_t march_static_one(uint32_t n, uint32_t n_step, const int* __restrict__ rays_alive, const float* __restrict__ rays_t,
                                                     const float* __restrict__ rays_o, const float* __restrict__ rays_d, float bound, float dt_gamma,
                                                     uint32_t max_steps, uint32_t C, uint32_t H, const uint8_t* __restrict__ grid,
                                                     const float* __restrict__ fars, float* __restrict__ xyzs, float* __restrict__ dirs,
                                                     float* __restrict__ deltas) {
    using namespace pnm;
    const int index = rays_alive[n];
    rays_o += (size_t)index * 3;
    rays_d += (size_t)index * 3;
    xyzs += (size_t)n * n_step * 3;
    dirs += (size_t)n * n_step * 3;
    deltas += (size_t)n * n_step * 2;
    const float ox = rays_o[0], oy = rays_o[1], oz = rays_o[2];
    const float dx = rays_d[0], dy = rays_d[1], dz = rays_d[2];
    const float rdx = 1 / dx, rdy = 1 / dy, rdz = 1 / dz;
    const float rH = 1 / (float)H;
    const float H3 = (float)(H * H * H);
    float t = rays_t[index];
    const float far = fars[index];
    const float dt_min = 2 * 1.73205080757f / max_steps;
    const float dt_max = 2 * 1.73205080757f * (1 << (C - 1)) / H;
    uint32_t step = 0;
    float last_t = t;  // noise = 0 (perturb = False): `t += clamp(...) * noise` leaves t unchanged
    while (t < far && step < n_step) {
        const float x = clampf(ox + t * dx, -bound, bound);
        const float y = clampf(oy + t * dy, -bound, bound);
        const float z = clampf(oz + t * dz, -bound, bound);
        const float dt = clampf(t * dt_gamma, dt_min, dt_max);
        const int level = max(mip_from_pos(x, y, z, (float)C), mip_from_dt(dt, (float)H, (float)C));
        const float mip_bound = fminf(scalbnf(1, level), bound);
        const float mip_rbound = 1 / mip_bound;
        const int nx = (int)clampf((float)(0.5 * (double)(x * mip_rbound + 1) * (double)H), 0.0f, (float)(H - 1));
        const int ny = (int)clampf((float)(0.5 * (double)(y * mip_rbound + 1) * (double)H), 0.0f, (float)(H - 1));
        const int nz = (int)clampf((float)(0.5 * (double)(z * mip_rbound + 1) * (double)H), 0.0f, (float)(H - 1));
        const uint32_t vox = (uint32_t)(level * H3 + (float)morton3D(nx, ny, nz));
        const bool occ = grid[vox / 8] & (1 << (vox % 8));
        if (occ) {
            xyzs[0] = x; xyzs[1] = y; xyzs[2] = z;
            dirs[0] = dx; dirs[1] = dy; dirs[2] = dz;
            t += dt;
            deltas[0] = dt;
            deltas[1] = t - last_t;
            last_t = t;
            xyzs += 3; dirs += 3; deltas += 2;
            step++;
        } else {
            const float tx = (((nx + 0.5f + 0.5f * signf(dx)) * rH * 2 - 1) * mip_bound - x) * rdx;
            const float ty = (((ny + 0.5f + 0.5f * signf(dy)) * rH * 2 - 1) * mip_bound - y) * rdy;
            const float tz = (((nz + 0.5f + 0.5f * signf(dz)) * rH * 2 - 1) * mip_bound - z) * rdz;
            const float tt = t + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
            do { t += clampf(t * dt_gamma, dt_min, dt_max); } while (t < tt);
        }
    }
    for (uint32_t s = step; s < n_step; s++) { deltas[0] = 0.0f; deltas[1] = 0.0f; deltas += 2; }  // the op-level wrapper zero-fills instead
    return step;
}

__global__ void __launch_bounds__(256) k_march_static_trip(PnTrip* trip, const int* __restrict__ rays_alive, const float* __restrict__ rays_t,
                                                           const float* __restrict__ rays_o, const float* __restrict__ rays_d, float bound,
                                                           float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H,
                                                           const uint8_t* __restrict__ grid, const float* __restrict__ fars, float* __restrict__ xyzs,
                                                           float* __restrict__ dirs, float* __restrict__ deltas, int* __restrict__ list) {
    const uint32_t n_alive = (uint32_t)trip->n_alive, n_step = (uint32_t)trip->n_step;
    const int lane = threadIdx.x & 63;
    for (uint32_t chunk = blockIdx.x; chunk * 256u < n_alive; chunk += gridDim.x) {
        const uint32_t n = chunk * 256u + threadIdx.x;
        const uint32_t emitted = n < n_alive ? march_static_one(n, n_step, rays_alive, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, C, H, grid, fars,
                                                                xyzs, dirs, deltas)
                                             : 0u;
        int inc = (int)emitted;  // inclusive wave scan of the sample counts, one atomic per wave
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int u = __shfl_up(inc, o);
            if (lane >= o) inc += u;
        }
        const int total = __shfl(inc, 63);
        int base = 0;
        if (lane == 63 && total > 0) base = atomicAdd(&trip->n_samples, total);
        base = __shfl(base, 63);
        const int first = base + inc - (int)emitted;
        for (uint32_t s = 0; s < emitted; s++) list[first + s] = (int)(n * n_step + s);
    }
}

__global__ void k_set_aabb(PnFrameDev* dev, float a0, float a1, float a2, float a3, float a4, float a5) {
    dev->aabb[0] = a0; dev->aabb[1] = a1; dev->aabb[2] = a2; dev->aabb[3] = a3; dev->aabb[4] = a4; dev->aabb[5] = a5;
    dev->resolution[0] = dev->resolution[1] = dev->resolution[2] = dev->resolution[3] = 0;
    dev->err = 0;
}

__global__ void __launch_bounds__(256) k_packbits(const float* __restrict__ grid, uint32_t N, float density_thresh, uint8_t* __restrict__ bitfield) {
    const uint32_t n = threadIdx.x + blockIdx.x * blockDim.x;
    if (n >= N) return;
    const float4 a = reinterpret_cast<const float4*>(grid)[2 * (size_t)n], b = reinterpret_cast<const float4*>(grid)[2 * (size_t)n + 1];
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    uint32_t bits = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) bits |= (v[i] > density_thresh) ? (1u << i) : 0u;
    bitfield[n] = (uint8_t)bits;
}

extern "C" int pn_packbits(const float* grid, uint32_t N, float density_thresh, uint8_t* bitfield, void* stream) {
    if (N == 0) return PN_OK;
    PN_REQUIRE(grid && bitfield && ((uintptr_t)grid & 15) == 0);
    k_packbits<<<pn_div_up(N, 256), 256, 0, (hipStream_t)stream>>>(grid, N, density_thresh, bitfield);
    PN_LAUNCH_CHECK();
    return PN_OK;
}

__device__ __forceinline__ uint32_t morton3D_invert1(uint32_t x) {  // raymarching.cu:73-81
    x = x & 0x49249249u;
    x = (x | (x >> 2)) & 0xc30c30c3u;
    x = (x | (x >> 4)) & 0x0f00f00fu;
    x = (x | (x >> 8)) & 0xff0000ffu;
    x = (x | (x >> 16)) & 0x0000ffffu;
    return x;
}
__global__ void __launch_bounds__(256) k_morton3D(const int* __restrict__ coords, uint32_t N, int* __restrict__ indices, int invert) {
    const uint32_t n = threadIdx.x + blockIdx.x * blockDim.x;
    if (n >= N) return;
    if (!invert) {
        indices[n] = (int)pnm::morton3D((uint32_t)coords[n * 3], (uint32_t)coords[n * 3 + 1], (uint32_t)coords[n * 3 + 2]);
    } else {  // `coords` is the output here
        const int ind = indices[n];
        int* c = const_cast<int*>(coords) + (size_t)n * 3;
        c[0] = (int)morton3D_invert1((uint32_t)(ind >> 0));
        c[1] = (int)morton3D_invert1((uint32_t)(ind >> 1));
        c[2] = (int)morton3D_invert1((uint32_t)(ind >> 2));
    }
}

extern "C" int pn_morton3D(const int* coords, uint32_t N, int* indices, void* stream) {
    if (N == 0) return PN_OK;
    PN_REQUIRE(coords && indices);
    k_morton3D<<<pn_div_up(N, 256), 256, 0, (hipStream_t)stream>>>(coords, N, indices, 0);
    PN_LAUNCH_CHECK();
    return PN_OK;
}

extern "C" int pn_morton3D_invert(const int* indices, uint32_t N, int* coords, void* stream) {
    if (N == 0) return PN_OK;
    PN_REQUIRE(coords && indices);
    k_morton3D<<<pn_div_up(N, 256), 256, 0, (hipStream_t)stream>>>(coords, N, const_cast<int*>(indices), 1);
    PN_LAUNCH_CHECK();
    return PN_OK;
}

// ------------------------------------------------------------------------------------------------ stable compaction
__global__ void __launch_bounds__(256) k_chunk_count(const int* __restrict__ rays_alive, uint32_t n, int* chunk_counts) {
    const uint32_t i = threadIdx.x + blockIdx.x * blockDim.x;
    const int c = __syncthreads_count(i < n && rays_alive[i] >= 0);
    if (threadIdx.x == 0) chunk_counts[blockIdx.x] = c;
}

// Packs the PN_SEGS segments of a list trip's sample list into the dense list the network kernel reads and publishes the total
// (workgroup s copies segment s behind the segments before it).  A dense trip has nothing to pack.
__global__ void __launch_bounds__(256) k_list_pack(PnTrip* trip, const int* __restrict__ samp_counts, const int* __restrict__ list_seg, int seg_cap,
                                                   int* __restrict__ list) {
    if (trip_is_dense(trip) || trip->n_alive <= 0) return;
    const int s = (int)blockIdx.x, lane = threadIdx.x & 63;
    __shared__ int before_s, total_s;
    if (threadIdx.x < 64) {
        const int c = samp_counts[lane * PN_SEG_STRIDE];
        int pre = lane < s ? c : 0, tot = c;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { pre += __shfl_xor(pre, o); tot += __shfl_xor(tot, o); }
        if (lane == 0) { before_s = pre; total_s = tot; }
    }
    __syncthreads();
    const int n = samp_counts[s * PN_SEG_STRIDE], off = before_s;
    const int* src = list_seg + (size_t)s * seg_cap;
    for (int i = threadIdx.x; i < n; i += 256) list[off + i] = src[i];
    if (s == 0 && threadIdx.x == 0) trip->n_samples = total_s;
}

// End of a trip of the frame driver, executed by ONE wave once every ray of the trip has been composited and `sum` of them survive: folds the
// march's segment counters into the records and clears them, and writes the next trip's record (renderer.py:839-846,891) — with ray groups
// (g_next != nullptr, see PnGroup) also the next trip's group records from this trip's and the per-group survivor counts of the composite:
// N_b // n_alive_b per group, exclusive sums for the first alive position and the first sample slot.
__device__ __forceinline__ void trip_epilogue(int lane, int sum, PnTrip* trip, PnTrip* next, uint32_t N_rays, uint32_t max_steps, int dense_trips,
                                              int* seg_counters, int* tail_diag, const PnGroup* __restrict__ g_cur, PnGroup* __restrict__ g_next,
                                              int* group_cnt, uint32_t group_rays, uint32_t n_groups) {
    if (seg_counters) {
        // this trip's march is over — fold its segment counters (seg_counters = [tail | sample | emitted | cursor | tail back] x PN_SEGS) into
        // the records and clear them for the next trip
        int* tail_c = seg_counters + lane * PN_SEG_STRIDE;
        int* samp_c = tail_c + PN_SEGS * PN_SEG_STRIDE;
        int* emit_c = samp_c + PN_SEGS * PN_SEG_STRIDE;
        int* curs_c = emit_c + PN_SEGS * PN_SEG_STRIDE;
        int* back_c = curs_c + PN_SEGS * PN_SEG_STRIDE;
        int tl = *tail_c + *back_c, em = *emit_c;
        *tail_c = 0; *samp_c = 0; *emit_c = 0; *curs_c = 0; *back_c = 0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { tl += __shfl_xor(tl, o); em += __shfl_xor(em, o); }
        if (lane == 0) { if (tail_diag) *tail_diag = tl; if (trip) trip->n_emitted = em; }
    }
    if (!next) return;
    if (g_next) {
        // 64 groups per round, running sums carried in (uniform) registers
        int alive_run = 0, slot_run = 0, live = 0, step0 = 1;
        for (uint32_t b0 = 0; b0 < n_groups; b0 += 64) {
            const uint32_t b = b0 + (uint32_t)lane;
            int cnt = 0, nstep = 0, stepb = 0;
            if (b < n_groups) {
                const PnGroup g = g_cur[b];
                cnt = (n_groups == 1) ? sum : group_cnt[b];
                if (n_groups > 1) group_cnt[b] = 0;
                stepb = g.step_base + g.n_step;
                const uint32_t rays_b = min(group_rays, N_rays - b * group_rays);  // N_b
                const bool over = cnt <= 0 || (uint32_t)stepb >= max_steps || g.n_step == 0;
                nstep = over ? 0 : max(min((int)(rays_b / (uint32_t)cnt), 8), 1);
            }
            const int slots = cnt * nstep;
            int a_inc = cnt, s_inc = slots;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int ua = __shfl_up(a_inc, o), us = __shfl_up(s_inc, o);
                if (lane >= o) { a_inc += ua; s_inc += us; }
            }
            if (b < n_groups) g_next[b] = PnGroup{alive_run + a_inc - cnt, nstep, slot_run + s_inc - slots, stepb};
            if (b == 0) step0 = nstep;
            alive_run += __shfl(a_inc, 63);
            slot_run += __shfl(s_inc, 63);
            int lv = nstep > 0 ? cnt : 0;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) lv += __shfl_xor(lv, o);
            live += lv;
        }
        step0 = __shfl(step0, 0);
        if (lane == 0) {
            // rays of groups that ran into max_steps stay listed until the next composite retires them; the frame is over when no group marches on
            const bool done = live <= 0;
            next->n_alive = done ? 0 : sum;
            next->n_step = done ? 1 : max(step0, 1);  // informational with groups (every kernel reads the group records)
            next->step_base = trip->step_base + trip->n_step;
            next->dense = (dense_trips && !done) ? 1 : 0;
            next->n_samples = (dense_trips && !done) ? slot_run : 0;
            next->n_emitted = 0;
        }
    } else if (lane == 0) {
        const int step = trip->step_base + trip->n_step;
        const bool done = (sum <= 0) || ((uint32_t)step >= max_steps);
        next->n_alive = done ? 0 : sum;
        next->n_step = done ? 1 : max(min((int)(N_rays / (uint32_t)sum), 8), 1);
        next->step_base = step;
        // dense trip (see trip_is_dense): the list is the identity over all slots; n_emitted = -1 marks a list trip
        // every trip after the first is dense: its rays are the ones that found a sample before (n_step == 1 then means more than half of
        // all rays are still alive — they will mostly fill their single slot too)
        const bool dense = dense_trips && !done;
        next->dense = dense ? 1 : 0;
        next->n_samples = dense ? sum * next->n_step : 0;
        next->n_emitted = 0;
    }
}

// Block c moves the survivors of chunk c to out[prefix(c) ...], keeping order (== rays_alive[rays_alive >= 0]).
// Block 0 also publishes the total and, in frame-driver mode, the next trip's record (renderer.py:839-846,891).
// Ray groups (g_next != nullptr, see PnGroup): chunk 0's first wave also writes the next trip's group records from this trip's records and the
// per-group survivor counts of k_composite — N_b // n_alive_b per group, exclusive sums for the first alive position and the first sample slot.
__global__ void __launch_bounds__(256) k_compact(const int* __restrict__ in, uint32_t n_arg, const int* __restrict__ chunk_counts,
                                                 int* __restrict__ out, int* n_out, PnTrip* trip, PnTrip* next, uint32_t N_rays,
                                                 uint32_t max_steps, int dense_trips, int* seg_counters, int* tail_diag,
                                                 const PnGroup* __restrict__ g_cur, PnGroup* __restrict__ g_next, int* group_cnt, uint32_t group_rays,
                                                 uint32_t n_groups) {
    __shared__ int red[4];
    __shared__ int woff[4];
    const uint32_t n = trip ? (uint32_t)trip->n_alive : n_arg;
    const uint32_t n_chunks = (n + 255) / 256;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    // bounded grid (see k_march): chunks are dealt round-robin; chunk 0 always runs once (it publishes the totals even when n == 0)
    for (uint32_t c = blockIdx.x; c == 0 || c * 256 < n; c += gridDim.x) {
        // prefix over earlier chunks (and, for chunk 0, the grand total)
        const uint32_t upto = (c == 0) ? n_chunks : c;
        int part = 0;
        for (uint32_t k = threadIdx.x; k < upto; k += 256) part += chunk_counts[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
        if (lane == 0) red[wid] = part;
        __syncthreads();
        const int sum = red[0] + red[1] + red[2] + red[3];
        const int offset = (c == 0) ? 0 : sum;
        if (c == 0 && wid == 0) {
            if (n_out && lane == 0) *n_out = sum;
            trip_epilogue(lane, sum, trip, next, N_rays, max_steps, dense_trips, seg_counters, tail_diag, g_cur, g_next, group_cnt, group_rays, n_groups);
        }
        const uint32_t i = c * 256 + threadIdx.x;
        const int v = (i < n) ? in[i] : -1;
        const bool keep = v >= 0;
        const unsigned long long m = __ballot(keep);
        const int rank = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) woff[wid] = __popcll(m);
        __syncthreads();
        int wbase = 0;
        for (int w = 0; w < wid; w++) wbase += woff[w];
        if (keep) out[offset + wbase + rank] = v;
        __syncthreads();  // red / woff are reused by the next chunk
    }
}

// ---- composite + stable compaction + end-of-trip bookkeeping in ONE launch (frame driver of the deformed render).
// kernel_composite_rays (raymarching.cu:827-923) followed by rays_alive = rays_alive[rays_alive >= 0] (renderer.py:887) is a scan: where a
// survivor goes depends on how many rays before it survive.  Two launches did that through per-chunk counts in memory (k_composite, k_compact);
// here workgroup b takes the chunks b, b + grid, ... of 256 * R consecutive alive positions, composites them, publishes each chunk's survivor
// count as (trip tag << 16 | count) and then sums the words of ALL chunks before its own, polling those that do not carry this trip's tag yet.
// No chain: a chunk waits for the composites of earlier chunks, never for their sums, so the launch lasts one composite plus one gather of at
// most n_chunks words.  (A decoupled look-back over 64 descriptors at a time was tried first: with every chunk of the trip in flight at once the
// prefixes have nothing to propagate from — 10 dependent steps on trip 0 — and 1 250 returning ticket / completion atomics on one address at
// 11.4 ns each: 174 us against 25 for the two launches.)  Progress: a chunk depends on lower-numbered chunks only and every workgroup takes its
// chunks in ascending order, so the launch finishes whenever ALL its workgroups can be resident at the same time — which is why the grid is bounded
// (PN_CC_GRID, 512 workgroups of 4 waves: three render lanes' composites together stay below the 8 192 wave slots of the part).  Rounds 2-3 launched
// one workgroup per chunk (2 500 on a frame's first trip) on the assumption that workgroups start in index order; the eight XCDs dispatch their shares
// independently, and two first-trip composites of different lanes could each fill an XCD with pollers waiting for a chunk whose workgroup had no slot
// on the other one: a deadlock, seen (as the poll guard's flag 16) in bench.py --config stress.  A workgroup's later chunks add only the words
// behind its previous chunk to the prefix it already has: 512 words per chunk instead of all before it.
// The tag makes last trip's words read as "not written yet"; the words are cleared once per frame (k_frame_prologue).  The workgroup of the
// trip's LAST chunk has the grand total and runs trip_epilogue: every earlier chunk has published its count, i.e. finished its composites and
// its per-group survivor atomics.  R = alive positions per thread (1; PN_CC_R0 = 2 / 4 for the frame's first trip are kept for experiments:
// fewer chunks make the quadratic gather smaller, but the strided accesses of the composite cost more than that saves).
template <int R>
__global__ void __launch_bounds__(256) k_composite_compact(float T_thresh, const int* __restrict__ cur, int* __restrict__ nxt, float* rays_t,
                                                           const float* __restrict__ sigmas, const float* __restrict__ rgbs, const float* __restrict__ deltas,
                                                           float* weights_sum, float* depth, float* image, PnTrip* trip, PnTrip* next,
                                                           unsigned* words, uint32_t tag, uint32_t N_rays, uint32_t max_steps, int dense_trips,
                                                           int* seg_counters, int* tail_diag, const PnGroup* __restrict__ g_cur, PnGroup* __restrict__ g_next,
                                                           int* group_cnt, uint32_t group_rays, uint32_t n_groups, int* err_flag, uint32_t poll_cap) {
    __shared__ int s_wcnt[4], s_part[4];
    const uint32_t n_alive = (uint32_t)trip->n_alive, n_step_trip = (uint32_t)trip->n_step;
    const uint32_t CH = 256u * R;
    const uint32_t n_chunks = max((n_alive + CH - 1) / CH, 1u);  // chunk 0 always runs: somebody has to write the next trip's record
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    bool have_prev = false;
    uint32_t c_prev = 0;
    int excl_prev = 0, count_prev = 0;
    for (uint32_t c = blockIdx.x; c < n_chunks; c += gridDim.x) {
        // ---- composite: R consecutive alive positions per thread
        int keep[R];
        int mine = 0;
#pragma unroll
        for (int r = 0; r < R; r++) {
            const uint32_t n = c * CH + threadIdx.x * R + r;
            bool alive = false;
            int grp = -1, index = -1;
            if (n < n_alive) {
                index = cur[n];
                uint32_t n_step = n_step_trip, slot0;
                ray_slots(g_cur, group_rays, index, n, n_step, slot0);
                if (g_cur) grp = (int)((uint32_t)index / group_rays);
                // n_step == 0: the ray's group has reached max_steps — the batch's loop is over (renderer.py:836), the ray is dropped
                if (n_step != 0) alive = composite_one(index, slot0, n_step, T_thresh, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image);
            }
            keep[r] = alive ? index : -1;
            mine += alive ? 1 : 0;
            if (n_groups > 1) {  // survivors per group: one atomic per run of equal group ids (the positions of the 64 lanes are R apart, still sorted)
                const unsigned long long am = __ballot(alive);
                const int prev = __shfl_up(grp, 1);
                const bool head = lane == 0 || grp != prev;
                const unsigned long long hm = __ballot(head);
                if (head && grp >= 0) {
                    const unsigned long long above = lane == 63 ? 0ull : hm & ~((2ull << lane) - 1ull);
                    const unsigned long long upto = above ? ((1ull << (__ffsll((long long)above) - 1)) - 1ull) : ~0ull;
                    const int cc = (int)__popcll(am & upto & ~((1ull << lane) - 1ull));
                    // returning form: the value has to be back before this chunk's count is published below (the epilogue reads the counters once
                    // every count is out); a release fence would do the same by writing this XCD's whole L2 back
                    if (cc) { const int old = atomicAdd(group_cnt + grp, cc); asm volatile("" ::"v"(old)); }
                }
            }
        }
        // ---- this chunk's survivor count; exclusive prefix of the thread inside the chunk
        int inc = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int u = __shfl_up(inc, o);
            if (lane >= o) inc += u;
        }
        if (lane == 63) s_wcnt[wid] = inc;
        __syncthreads();
        const int count = s_wcnt[0] + s_wcnt[1] + s_wcnt[2] + s_wcnt[3];
        int tbase = inc - mine;
        for (int w = 0; w < wid; w++) tbase += s_wcnt[w];
        // relaxed, device scope: the word IS the message (the XCDs' L2s are not coherent with each other: release / acquire at device scope
        // write back and invalidate whole caches — with them this kernel took 185 us on trip 0)
        if (threadIdx.x == 0) __hip_atomic_store(words + c, (tag << 16) | (unsigned)count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // ---- survivors of all chunks before this one: the words of the chunks behind this workgroup's previous chunk, on top of what it had there
        int part = 0;
        for (uint32_t k = (have_prev ? c_prev + 1 : 0u) + threadIdx.x; k < c; k += 256) {
            unsigned w;
            uint32_t polls = 0;
            do {
                w = __hip_atomic_load(words + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                // never seen: a word that stays unwritten would mean the dispatcher started this workgroup before a lower-numbered one that has
                // no slot yet.  Rather than hang the GPU, give up after ~a second, flag the frame (err bit 16) and carry on with garbage.
                if (++polls > poll_cap) { if (err_flag) atomicOr(err_flag, 16); w = tag << 16; }
            } while ((w >> 16) != tag);
            part += (int)(w & 0xFFFFu);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
        if (lane == 0) s_part[wid] = part;
        __syncthreads();
        const int excl = (have_prev ? excl_prev + count_prev : 0) + s_part[0] + s_part[1] + s_part[2] + s_part[3];
        have_prev = true; c_prev = c; excl_prev = excl; count_prev = count;
        // ---- survivors in order
        int w0 = excl + tbase;
#pragma unroll
        for (int r = 0; r < R; r++)
            if (keep[r] >= 0) nxt[w0++] = keep[r];
        if (c == n_chunks - 1 && wid == 0)
            trip_epilogue(lane, excl + count, trip, next, N_rays, max_steps, dense_trips, seg_counters, tail_diag, g_cur, g_next, group_cnt, group_rays, n_groups);
        __syncthreads();  // s_wcnt / s_part are rewritten by the next chunk
    }
}

extern "C" uint32_t pn_compact_scratch_ints(uint32_t n) { return pn_div_up(n, 256) + 1; }

extern "C" int pn_compact_rays(const int* rays_alive, uint32_t n, int* out, int* n_out, int* scratch, void* stream) {
    PN_REQUIRE(n_out);
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) { PN_HIP_CHECK(hipMemsetAsync(n_out, 0, sizeof(int), st)); return PN_OK; }  // empty tensors have null data pointers
    PN_REQUIRE(rays_alive && out && scratch);
    const uint32_t chunks = pn_div_up(n, 256);
    k_chunk_count<<<chunks, 256, 0, st>>>(rays_alive, n, scratch);
    k_compact<<<chunks, 256, 0, st>>>(rays_alive, n, scratch, out, n_out, nullptr, nullptr, 0, 0, 0, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0);
    PN_LAUNCH_CHECK();
    return PN_OK;
}

// ------------------------------------------------------------------------------------------------ whole frame

#define PN_MAX_TRIPS 1100
#define PN_MIN_RAY_BATCH 64  // smallest pn_render_opts::ray_batch (sizes the group records of a workspace)
#define PN_TRIP_BATCH 8
#define PN_TRIP_MARGIN 2  // trips a captured render carries beyond what its sizing frame needed (harness: measured + 2)
#define PN_TIMED_TRIPS 64

#include "pn_trips_fused.h"

struct pn_frame {
    uint32_t max_rays, max_vtx, max_cells;
    float *nears, *fars, *rays_t, *xyzs, *dirs, *deltas, *sigmas, *rgbs;
    float* acc_image;  // [max_rays,3] colour accumulated by composite; the epilogue writes image = acc + (1 - weights_sum) * bg, so a frame can be
                       // continued with more trips and finished again (pn_render_continue)
    int *alive_a, *alive_b, *list, *chunk_counts;
    TailEntry* tail;    // [PN_SEGS x seg_cap] rays handed from k_march to k_march_tail
    int* list_seg;      // [PN_SEGS x seg_cap] segmented sample list of a list trip (k_list_pack -> list)
    int* active_seg;    // [PN_SEGS x seg_cap] trip 0: the slots k_march_skip left something to march for
    uint32_t seg_cap;
    uint32_t* cell_bits;  // [2][(max_cells + 31) / 32] bit c: search cell c has candidates / is within one cell of such a cell (cleared by
                          // k_frame_tables, set by k_frame_lists)
    uint32_t* grid_regions;  // [PN_GRID_REGION_WORDS] --cut frames: the region map of the skip pre-pass (MarchIO::grid_regions; k_frame_prologue)
    float* fars_eff;      // [max_rays] the rays' ends shortened to where they can still find candidates (k_march_skip)
    int* seg_counters;  // [6][PN_SEGS] counters, one per 128 B: tail | sample | emitted | tail cursor | tail back (cleared by each trip's compaction) | active (k_frame_rays)
    int* tail_counts;   // [PN_MAX_TRIPS + 2] diagnostics: rays each trip handed to the tail pass
    int *pig_cnt, *pig_bgn, *pig_idx, *pig_cursor;
    MarchSide side;  // candidate lists + packed IP records of the cooperative march
    PnTrip* trips;  // [PN_MAX_TRIPS + 2]
    PnGroup* groups;     // [2][max_groups] ray-group records of the current / next trip (trip parity), see PnGroup
    int* group_cnt;      // [max_groups] survivors per group (composite -> trip_epilogue, which clears them)
    uint32_t max_groups;
    PnFrameDev* dev;
    float* cut_bounds;
    PnTrip* trips_pinned;  // host-pinned mirror
    PnFrameDev* dev_pinned;
    float cut_bounds_host[6];
    int cut_bounds_valid;
    int last_trips;  // trips enqueued by the last render (incl. continuations)
    uint32_t last_N;
    uint32_t last_group_rays;  // ray_batch of the last render (a continuation must use the same)
    int tables_n_vtx;  // IP count the workspace's tables were built for (0: none); pn_render_opts::reuse_tables
    unsigned long long* march_counters;  // device [4], see MarchParams::stats
    int march_counters_on;
    hipEvent_t ev[PN_TIMED_TRIPS][3];    // measurement mode: before march / after march / after network, per trip
    int timed_trips;
    unsigned long long* stamps;          // device [PN_TIMED_TRIPS][3]: the same three points as 100 MHz wall-clock stamps written by one-lane kernels
    int stamped;                         // — the form that also works inside a captured graph (HIP events recorded in a graph cannot be timed)
    // the fused later trips (pn_trips_fused.h)
    int* fused_ctl;                      // [PN_FUSED_CTL_INTS] hand-out cursors, per-trip counters, workgroups done: zero between launches
    uint32_t fused_blocks;               // workgroups of a fused launch (one per CU); xyzs / dirs / deltas / sigmas / rgbs hold 64 slots per wave of it
    unsigned long long* fused_clocks;    // device [8] phase clocks of the fused launches (march_counters_on & 4)
    int fused_first;                     // first trip the last render ran fused (-1: none): where its time stamps sit
    float* t_resume;                     // [max_rays] per alive slot of a frame's first trip: where k_march_skip left the ray
    int* blist;                          // whole-frame fused launch: [2 x blist_cap] ray ids of the first trip's shares / of the rays that outlive it
    int4* strag;                         // [blist_cap] its rays still searching after the one-lane rounds
    uint32_t blist_cap;
    int skip_done;                       // the last render on this workspace ran k_march_skip (a continuation from trip 0 must not run it again)
    int head_marched;                    // ... and the first trip's march launches (pn_render_opts.fused_fold): a continuation from trip 0 goes on behind them
    int fused_mode;                      // form of the last fused launch that was enqueued: 0 later trips, 1 whole frame, 2 first trip folded in (pn_trips_fused.h)
};

// image = acc + (1 - weights_sum) * bg ; depth = clamp(depth - nears, 0) / (fars - nears) (renderer.py:896-899)
// The first wave of the launch also closes the frame's books: trips_run (+ the trips a fused launch ran, whose number only the device knows), the summary
// of the trip records (what pn_render_status reports) and the rays a fixed-trip render left alive.
__global__ void __launch_bounds__(256) k_frame_finish(uint32_t N, float bg, const float* __restrict__ nears, const float* __restrict__ fars,
                                                      const float* __restrict__ weights_sum, const float* __restrict__ depth_0,
                                                      const float* __restrict__ acc, float* __restrict__ image, float* __restrict__ depth,
                                                      const PnTrip* __restrict__ trips, PnFrameDev* dev, int trips_run, int add_fused) {
    const uint32_t i = threadIdx.x + blockIdx.x * blockDim.x;
    if (i < 64) {
        const int lane = (int)i;
        const int t_final = min(trips_run + (add_fused ? dev->fused_trips : 0), PN_MAX_TRIPS);
        int n_trips = 0;
        long long n_samples = 0;
        for (int k = lane; k < t_final; k += 64) {
            const PnTrip* r = trips + k;
            if (r->n_alive > 0) n_trips++;
            n_samples += r->dense ? r->n_emitted : r->n_samples;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { n_trips += __shfl_xor(n_trips, o); n_samples += __shfl_xor(n_samples, o); }
        if (lane == 0) {
            const int left = trips[t_final].n_alive;
            dev->trips_run = t_final;
            dev->fused_trips = 0;
            dev->stat_trips = n_trips;
            dev->stat_samples = n_samples;
            dev->alive_at_exit = left;
            if (left > 0) atomicAdd(&dev->unfinished, left);
        }
    }
    if (i >= N) return;
    const float k = (1 - weights_sum[i]) * bg;
    image[i * 3] = acc[i * 3] + k;
    image[i * 3 + 1] = acc[i * 3 + 1] + k;
    image[i * 3 + 2] = acc[i * 3 + 2] + k;
    depth[i] = fmaxf(depth_0[i] - nears[i], 0.0f) / (fars[i] - nears[i]);
}

// measurement: a stream-ordered time stamp (constant 100 MHz clock) as an ordinary kernel node, so that it can live inside a captured graph
__global__ void k_stamp(unsigned long long* slot) { *slot = __builtin_amdgcn_s_memrealtime(); }

// ---- fused frame prologue (3 launches instead of 13; every one of them was a few-microsecond kernel with a launch gap)
// (1) k_frame_tables, ONE workgroup of 1024 threads: IP bounding box +-1e-3 and spatial-hash resolution (nerf/renderer.py:782-791), the spatial
//     hash itself (count -> scan -> cursor fill -> per-cell sort, = k_pig_*) and the per-cell candidate-list offsets (k_nb_count
//     + scan).  The phases talk through global memory (L2) with relaxed agent-scope atomic loads where a value was produced by
//     an atomic or by another thread of the block, and __syncthreads() in between.
//     LARGE = false: everything in this one workgroup, per-cell counters in LDS (two 16-bit counters per word: up to ~290 k cells minus the
//     staged index table fit the 160 KB).  LARGE = true: the grid is too large for that (bound 2 with --cut: the spatial hash spans +-bound,
//     67^3 = 300 k cells at hgs 0.06) — this kernel only does the bounding box / resolution part and the tables are built by the
//     multi-workgroup kernels of get_pnts_in_grids (k_pig_*) + k_nb_count + a second scan; same tables, bit for bit.
template <bool LARGE>
__global__ void __launch_bounds__(1024) k_frame_tables(const float* __restrict__ p_def, int n_vtx, int cut, float bound, float hgs, int max_cells,
                                                       PnFrameDev* dev, int* pig_cnt, int* pig_bgn, int* pig_idx, int* pig_cursor, uint32_t* cell_bits) {
    extern __shared__ unsigned cnt2[];  // per-cell point counts, two 16-bit counters per word (a cell never holds 65 536 IPs)
    for (int w = threadIdx.x; w < 2 * ((max_cells + 31) / 32); w += blockDim.x) cell_bits[w] = 0u;  // both maps; set by k_frame_lists
    __shared__ float smin[3][16], smax[3][16];
    __shared__ float sh_min[3];
    __shared__ int sh_res[4];
    __shared__ int wsum[16];
    __shared__ int carry_s;
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int i = threadIdx.x; i < n_vtx; i += blockDim.x)
#pragma unroll
        for (int c = 0; c < 3; c++) { const float v = p_def[i * 3 + c]; mn[c] = fminf(mn[c], v); mx[c] = fmaxf(mx[c], v); }
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < 3; c++) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { mn[c] = fminf(mn[c], __shfl_xor(mn[c], o)); mx[c] = fmaxf(mx[c], __shfl_xor(mx[c], o)); }
        if (lane == 0) { smin[c][wid] = mn[c]; smax[c][wid] = mx[c]; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {  // renderer.py:782-791
        int ncell = 1;
        for (int c = 0; c < 3; c++) {
            float a = smin[c][0], b = smax[c][0];
            for (int w = 1; w < 16; w++) { a = fminf(a, smin[c][w]); b = fmaxf(b, smax[c][w]); }
            if (cut) { a = -bound; b = bound; }
            const float lo = a - 1e-3f, hi = b + 1e-3f;
            dev->aabb[c] = lo;
            dev->aabb[3 + c] = hi;
            sh_min[c] = lo;
            const int r = (int)ceilf((hi - lo) / hgs);
            dev->resolution[c] = r;
            sh_res[c] = r;
            ncell *= r;

        }
        int err = 0;
        if (ncell > max_cells || ncell <= 0) { err = 4; ncell = 0; }
        dev->resolution[3] = ncell;
        dev->err = err;
        dev->nb_alloc = 0;
        sh_res[3] = ncell;
        carry_s = 0;
    }
    __syncthreads();
    const int n_grid_all = sh_res[3], r0 = sh_res[0], r1 = sh_res[1], r2 = sh_res[2];
    const float b0 = sh_min[0], b1 = sh_min[1], b2 = sh_min[2];
    if (n_grid_all == 0) return;
    // the cells that hold integration points, exactly as p2g files them (a point outside the grid whose flat index still lies in [0, n_grid) is filed under that
    // index, as in the reference): their extent per axis -> PnFrameDev::ip_lo / ip_hi (k_frame_prologue builds lists only near them)
    __shared__ int s_lo[3], s_hi[3];
    if (threadIdx.x < 3) { s_lo[threadIdx.x] = 0x7fffffff; s_hi[threadIdx.x] = -1; }
    __syncthreads();
    for (int p = threadIdx.x; p < n_vtx; p += blockDim.x) {
        const int q0 = (int)floorf((p_def[p * 3] - b0) / hgs), q1 = (int)floorf((p_def[p * 3 + 1] - b1) / hgs), q2 = (int)floorf((p_def[p * 3 + 2] - b2) / hgs);
        const int gid = q2 * r1 * r0 + q1 * r0 + q0;
        if (gid < 0 || gid >= n_grid_all) continue;
        const int c0 = gid % r0, c1 = (gid / r0) % r1, c2 = gid / (r0 * r1);
        atomicMin(&s_lo[0], c0); atomicMax(&s_hi[0], c0);
        atomicMin(&s_lo[1], c1); atomicMax(&s_hi[1], c1);
        atomicMin(&s_lo[2], c2); atomicMax(&s_hi[2], c2);
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const bool any = s_hi[threadIdx.x] >= 0;
        dev->ip_lo[threadIdx.x] = any ? s_lo[threadIdx.x] : 0;
        dev->ip_hi[threadIdx.x] = any ? s_hi[threadIdx.x] : -1;
    }
    if (LARGE) return;  // the tables themselves are built by the multi-workgroup kernels (pn_frame_prologue)
    const int n_grid = n_grid_all;
    for (int g = threadIdx.x; g < (n_grid + 1) / 2; g += blockDim.x) cnt2[g] = 0u;
    __syncthreads();
    auto cell_of = [&](int p) {  // p2g, nerf/utils.py:389-407
        const int g0 = (int)floorf((p_def[p * 3] - b0) / hgs);
        const int g1 = (int)floorf((p_def[p * 3 + 1] - b1) / hgs);
        const int g2 = (int)floorf((p_def[p * 3 + 2] - b2) / hgs);
        const int gid = g2 * r1 * r0 + g1 * r0 + g0;
        return (gid < 0 || gid >= n_grid) ? -1 : gid;
    };
    auto count_of = [&](int g) { return (int)((cnt2[g >> 1] >> (16 * (g & 1))) & 0xFFFFu); };
    for (int p = threadIdx.x; p < n_vtx; p += blockDim.x) {
        const int gid = cell_of(p);
        if (gid >= 0) atomicAdd(&cnt2[gid >> 1], 1u << (16 * (gid & 1)));
        else atomicOr(&dev->err, 2);
    }
    __syncthreads();
    // exclusive scan of the counts -> pig_cnt / pig_bgn / pig_cursor.  (Rounds 1-2 also summed every cell's 27-neighbourhood here and scanned
    // that for the candidate-list offsets: 27 LDS reads + the neighbour arithmetic per cell on ONE compute unit were 50 of this kernel's 82 us.
    // The lists now get their space from a bump counter in k_frame_prologue, which runs on the whole chip.)
    {
        int* out_cnt = pig_cnt;
        int* out_bgn = pig_bgn;
        int* out_cur = pig_cursor;
        for (int base = 0; base < n_grid; base += 4096) {
            const int i0 = base + threadIdx.x * 4;
            int v[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int c = i0 + k;
                int val = 0;
                if (c < n_grid) {
                    val = count_of(c);
                    out_cnt[c] = val;
                }
                v[k] = val;
            }
            const int tsum = v[0] + v[1] + v[2] + v[3];
            int inc = tsum;  // inclusive wave scan
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int u = __shfl_up(inc, o);
                if (lane >= o) inc += u;
            }
            if (lane == 63) wsum[wid] = inc;
            __syncthreads();
            int woff = 0;
            for (int w = 0; w < wid; w++) woff += wsum[w];
            int total = 0;
            for (int w = 0; w < 16; w++) total += wsum[w];
            int run = carry_s + woff + inc - tsum;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (i0 + k < n_grid) { out_bgn[i0 + k] = run; out_cur[i0 + k] = run; }
                run += v[k];
            }
            __syncthreads();
            if (threadIdx.x == 0) carry_s += total;
            __syncthreads();
        }
        if (threadIdx.x == 0) carry_s = 0;
        __syncthreads();
    }
    // cursor fill (get_pig_idx, nerf/utils.py:427-443): slots claimed through the per-cell cursor, entries staged in LDS ...
    int* lidx = reinterpret_cast<int*>(cnt2 + (max_cells + 1) / 2);
    for (int p = threadIdx.x; p < n_vtx; p += blockDim.x) {
        const int gid = cell_of(p);
        if (gid >= 0) lidx[atomicAdd(pig_cursor + gid, 1)] = p;
    }
    __syncthreads();
    // ... then ascending point id inside each cell (k_pig_sort): the table does not depend on the order of the atomics
    for (int g = threadIdx.x; g < n_grid; g += blockDim.x) {
        const int c = count_of(g);
        if (c < 2) continue;
        int* a = lidx + __hip_atomic_load(pig_bgn + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int i = 1; i < c; i++) {
            const int v = a[i];
            int j = i - 1;
            while (j >= 0 && a[j] > v) { a[j + 1] = a[j]; j--; }
            a[j + 1] = v;
        }
    }
    __syncthreads();
    for (int p = threadIdx.x; p < n_vtx; p += blockDim.x) pig_idx[p] = lidx[p];
}

// (2) k_frame_prologue, ONE launch for everything else the frame needs before its first trip, three independent block ranges:
//     [0, list_blocks)  candidate lists, 8 lanes per cell and 32 cells per workgroup round: the 27 neighbours' counts (lane j holds visiting
//                       positions j, j+8, j+16, j+24), their running sum inside the 8-lane group (= where each neighbour's entries go), list space
//                       by ONE returning atomic per workgroup round on dev->nb_alloc (the order of the lists in memory means nothing; an
//                       atomic per cell on one address would serialise, see PN_SEGS), the entries, the (begin, end) record, and the two cell maps:
//                       "has candidates" and — scattered to the 27 neighbours of every such cell — "within one cell of a cell with candidates".
//                       The maps are a few cache lines (chair: 10 k cells = 10 lines) and atomics on one LINE queue like atomics on one address
//                       (measured: 160 k atomicOr straight to global memory made this kernel 125 us), so every workgroup collects its bits in
//                       LDS (lds_words > 0) and ORs only its non-zero words into the global maps;
//     [.., + pack_blocks)  the packed IP records (k_pack_ip);
//     the rest             k_near_far + the per-ray initialisation: zeroed accumulators (renderer.py:807-809), rays_alive = arange(N) (:828),
//                          rays_t = nears (:829), zeroed trip records / counters, trip 0 = (N rays, n_step 1).  Needs only the bounding box.
struct FramePrologue {
    // lists
    int n_grid_max; const int* n_grid_dev; const int* res; const int* pig_cnt; const int* pig_bgn; const int* pig_idx; const float* p_def; int swap;
    int2* nb_rng; float4* nb; int nb_capacity; int list_blocks; uint32_t* cell_bits; int lds_words;
    // records
    int pack_blocks; int n_vtx; const float* p_ori; const float* F_IP; const float* dF_IP; float* rec;
    // rays
    const float* rays_o; const float* rays_d; PnFrameDev* dev; uint32_t N; float min_near; float* nears; float* fars; float* rays_t; PnTrip* trips;
    int* tail_counts; int* seg_counters; int n_trip_records; int* alive; float* weights_sum; float* depth_0; float* image; PnGroup* groups;
    int* group_cnt; uint32_t group_rays; uint32_t n_groups; int* chunk_words;
    uint32_t tile_w, tile_lw;  // tile_w > 0: alive list starts in 16 x 4 pixel tile order (pn_render_opts.ray_tile_w, validated by the host)
    // early_finish: the frame's epilogue is left to the fused launch (pn_trips_fused.h: finalize) — every ray gets the pixel of a ray without samples here
    int early_finish; float bg; float* image_out; float* depth_out;
    // --cut frames: the region map (MarchIO::grid_regions): gr_blocks workgroups, a lane per region of the (H / 8)^3 grid
    const uint8_t* grid; uint32_t* grid_regions; int gr_blocks; int gr_R; int gr_C; uint32_t gr_H; float gr_bound; const float* cut_bounds;
};
#define PN_GRID_REGION_WORDS 1024  // 32 768 regions: H <= 256

__device__ __forceinline__ void frame_lists_block(const FramePrologue& a) {
    extern __shared__ uint32_t lds_bits[];  // [2][lds_words] when lds_words > 0
    __shared__ int wtot[4];
    __shared__ int blk_base;
    const int n_grid = min(*a.n_grid_dev, a.n_grid_max);
    const int r0 = a.res[0], r1 = a.res[1], r2 = a.res[2];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, sub = threadIdx.x & 7;
    const int words = (a.n_grid_max + 31) / 32;
    const int per_round = a.list_blocks * 32;
    const int ilo0 = a.dev->ip_lo[0], ilo1 = a.dev->ip_lo[1], ilo2 = a.dev->ip_lo[2], ihi0 = a.dev->ip_hi[0], ihi1 = a.dev->ip_hi[1], ihi2 = a.dev->ip_hi[2];
    const bool in_lds = a.lds_words > 0;
    if (in_lds) {
        for (int w = threadIdx.x; w < 2 * a.lds_words; w += blockDim.x) lds_bits[w] = 0u;
        __syncthreads();
    }
    for (int c0 = 0; c0 < n_grid; c0 += per_round) {  // uniform trip count: the round's workgroup-wide sum needs every thread
        const int c = c0 + (int)blockIdx.x * 32 + ((int)threadIdx.x >> 3);
        const bool valid = c < n_grid;
        int g0 = 0, g1 = 0, g2 = 0;
        if (valid) nb_cell_coords(c, r0, r1, g0, g1, g2);
        // a cell more than one cell away from every integration point has an empty list: no neighbour to look at (PnFrameDev::ip_lo / ip_hi)
        const bool near_ips = valid && g0 >= ilo0 - 1 && g0 <= ihi0 + 1 && g1 >= ilo1 - 1 && g1 <= ihi1 + 1 && g2 >= ilo2 - 1 && g2 <= ihi2 + 1;
        int cell[4], cnt[4], before[4];
        int total = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {  // visiting position q = 0 is the cell itself, q = 1..26 its neighbours q - 1
            const int q = sub + 8 * k;
            cell[k] = (near_ips && q < 27) ? ((q == 0) ? c : nb_neighbour(q - 1, a.swap, g0, g1, g2, r0, r1, r2)) : -1;
            cnt[k] = cell[k] >= 0 ? a.pig_cnt[cell[k]] : 0;
            int inc = cnt[k];  // running sum over the 8 lanes of the group
#pragma unroll
            for (int o = 1; o < 8; o <<= 1) {
                const int u = __shfl_up(inc, o, 8);
                if (sub >= o) inc += u;
            }
            before[k] = total + inc - cnt[k];
            total += __shfl(inc, 7, 8);
        }
        // list space: exclusive sum of the round's 32 totals + one bump of the frame's counter
        const int mine = (sub == 0) ? total : 0;
        int inc = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int u = __shfl_up(inc, o);
            if (lane >= o) inc += u;
        }
        if (lane == 63) wtot[wid] = inc;
        const int in_wave = __shfl(inc - mine, lane & ~7);  // the group's first lane holds the cell's exclusive offset
        __syncthreads();
        if (threadIdx.x == 0) {
            const int sum = wtot[0] + wtot[1] + wtot[2] + wtot[3];
            blk_base = sum ? atomicAdd(&a.dev->nb_alloc, sum) : 0;
        }
        __syncthreads();
        int w0 = blk_base + in_wave;
        for (int w = 0; w < wid; w++) w0 += wtot[w];
        __syncthreads();  // wtot / blk_base are rewritten by the next round
        if (!valid) continue;
        const bool fits = w0 + total <= a.nb_capacity;
        if (sub == 0) {
            a.nb_rng[c] = (total > 0 && fits) ? make_int2(w0, w0 + total) : make_int2(0, 0);
            if (total > 0 && !fits) atomicOr(&a.dev->err, 8);
        }
        if (total == 0 || !fits) continue;
        if (sub == 0) atomicOr((in_lds ? lds_bits : a.cell_bits) + (c >> 5), 1u << (c & 31));
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (cell[k] < 0) continue;
            // second map: `cell[k]` lies within one cell of a cell with candidates
            atomicOr((in_lds ? lds_bits + a.lds_words : a.cell_bits + words) + (cell[k] >> 5), 1u << (cell[k] & 31));
            const int n = cnt[k], b = a.pig_bgn[cell[k]];
            for (int i = 0; i < n; i++) {
                const int ip = a.pig_idx[b + i];
                a.nb[w0 + before[k] + i] = make_float4(a.p_def[ip * 3], a.p_def[ip * 3 + 1], a.p_def[ip * 3 + 2], __int_as_float(ip));
            }
        }
    }
    if (in_lds) {
        __syncthreads();
        for (int w = threadIdx.x; w < 2 * a.lds_words; w += blockDim.x) {
            const uint32_t v = lds_bits[w];
            if (v) atomicOr(a.cell_bits + (w < a.lds_words ? w : words + (w - a.lds_words)), v);
        }
    }
}

__device__ __forceinline__ void frame_rays_block(const FramePrologue& a, uint32_t block) {
    const uint32_t n = threadIdx.x + block * blockDim.x;
    if (block == 0 && a.groups) {  // trip 0 of every group: all its rays, one sample each (max(min(N_b // N_b, 8), 1))
        for (uint32_t b = threadIdx.x; b < a.n_groups; b += blockDim.x) {
            a.groups[b] = PnGroup{(int)(b * a.group_rays), 1, (int)(b * a.group_rays), 0};
            a.group_cnt[b] = 0;
        }
    }
    // the trip records (1102 x 256 B) are cleared four to a workgroup, one dword per lane (one workgroup clearing all of them was this launch's
    // critical path); trip 0 (n_step == 1) is a list trip over all N rays
    {
        const int t = (int)block * 4 + (int)(threadIdx.x >> 6), w = threadIdx.x & 63;
        if (t < a.n_trip_records) {
            int v = 0;
            if (t == 0 && w == 0) v = (a.dev->err & 7) ? 0 : (int)a.N;  // flags of k_frame_tables stop the frame
            if (t == 0 && w == 1) v = 1;                                // n_step = max(min(N // N, 8), 1)
            reinterpret_cast<int*>(a.trips + t)[w] = v;
            if (w == 0) a.tail_counts[t] = 0;
        }
        // a frame with fewer rays than that: the last workgroup clears what is left
        if (block + 1 == gridDim.x - (uint32_t)(a.list_blocks + a.pack_blocks)) {
            for (int t2 = ((int)block + 1) * 4 + (int)(threadIdx.x >> 6); t2 < a.n_trip_records; t2 += 4) {
                reinterpret_cast<int*>(a.trips + t2)[w] = 0;
                if (w == 0) a.tail_counts[t2] = 0;
            }
        }
    }
    if (block == 0)
        for (int t = threadIdx.x; t < 6 * PN_SEGS; t += blockDim.x) a.seg_counters[t * PN_SEG_STRIDE] = 0;
    if (block == 0 && threadIdx.x == 0) {
        // Every frame starts with no fused trips on its books: a frame finished INSIDE a fused launch leaves its count behind (no k_frame_finish ran to
        // clear it), and a later frame on this pn_frame whose fused launch steps aside would otherwise read it as "ran to the end" (round-4 advisor).
        a.dev->fused_trips = 0;
        if (a.early_finish) {  // the frame's books until the fused launch closes them (if it steps aside: an unfinished frame at trip 0)
            a.dev->trips_run = 0; a.dev->stat_trips = 0; a.dev->stat_samples = 0; a.dev->alive_at_exit = (int)a.N;
        }
    }
    if (threadIdx.x == 0) a.chunk_words[block] = 0;  // one (tag, count) word per 256 rays (+ the spare ones by the last workgroup), see k_composite_compact
    if (threadIdx.x < 2 && block + 1 == gridDim.x - (uint32_t)(a.list_blocks + a.pack_blocks)) a.chunk_words[block + 1 + threadIdx.x] = 0;
    if (n >= a.N) return;
    const float* aabb = a.dev->aabb;
    const float ox = a.rays_o[n * 3], oy = a.rays_o[n * 3 + 1], oz = a.rays_o[n * 3 + 2];
    const float dx = a.rays_d[n * 3], dy = a.rays_d[n * 3 + 1], dz = a.rays_d[n * 3 + 2];
    const float rdx = 1 / dx, rdy = 1 / dy, rdz = 1 / dz;
    float near = (aabb[0] - ox) * rdx, far = (aabb[3] - ox) * rdx;
    if (near > far) { float c = near; near = far; far = c; }
    float near_y = (aabb[1] - oy) * rdy, far_y = (aabb[4] - oy) * rdy;
    if (near_y > far_y) { float c = near_y; near_y = far_y; far_y = c; }
    bool miss = (near > far_y || near_y > far);
    if (!miss) {
        if (near_y > near) near = near_y;
        if (far_y < far) far = far_y;
        float near_z = (aabb[2] - oz) * rdz, far_z = (aabb[5] - oz) * rdz;
        if (near_z > far_z) { float c = near_z; near_z = far_z; far_z = c; }
        miss = (near > far_z || near_z > far);
        if (!miss) {
            if (near_z > near) near = near_z;
            if (far_z < far) far = far_z;
            if (near < a.min_near) near = a.min_near;
        }
    }
    if (miss) near = far = FLT_MAX;
    a.nears[n] = near;
    a.fars[n] = far;
    a.rays_t[n] = near;  // rays_t = nears.clone() (renderer.py:829)
    // rays_alive = arange(N) (renderer.py:828) — or, for a whole image, the same set in 16 x 4 pixel tiles: slot n = pixel (n & 15, (n >> 4) & 3) of
    // tile n / 64 (tiles row-major).  A wave's 64 slots are then a tile, and every later alive list (stable compaction) keeps that order
    uint32_t ray = n;
    if (a.tile_w) {
        const uint32_t lw = a.tile_lw, tile = n >> 6, in = n & 63u, tiles_x = a.tile_w >> lw;  // tile of (1 << lw) x (64 >> lw) pixels, lw = 4
        ray = ((tile / tiles_x) * (64u >> lw) + (in >> lw)) * a.tile_w + ((tile % tiles_x) << lw) + (in & ((1u << lw) - 1u));
    }
    a.alive[n] = (int)ray;
    a.weights_sum[n] = 0.f;
    a.depth_0[n] = 0.f;
    a.image[n * 3] = 0.f; a.image[n * 3 + 1] = 0.f; a.image[n * 3 + 2] = 0.f;
    if (a.early_finish) {  // k_frame_finish's expressions for weights_sum = depth_0 = acc = 0 (renderer.py:896-899)
        const float k = (1 - 0.f) * a.bg;
        a.image_out[n * 3] = 0.f + k; a.image_out[n * 3 + 1] = 0.f + k; a.image_out[n * 3 + 2] = 0.f + k;
        a.depth_out[n] = fmaxf(0.f - near, 0.0f) / (far - near);
    }
}

__global__ void __launch_bounds__(256) k_frame_prologue(FramePrologue a) {
    const int b = (int)blockIdx.x;
    if (b < a.list_blocks) { frame_lists_block(a); return; }
    if (b < a.list_blocks + a.pack_blocks) {  // k_pack_ip
        const int t = threadIdx.x + (b - a.list_blocks) * 256;
        const int ip = t / PN_REC_FLOATS, j = t % PN_REC_FLOATS;
        if (ip < a.n_vtx) a.rec[t] = pnm2::pack_ip_float(j, ip, a.p_ori, a.p_def, a.F_IP, a.dF_IP);
        return;
    }
    if (b < a.list_blocks + a.pack_blocks + a.gr_blocks) {  // region map of the density bitfield (pn_march_window.h: region_dda)
        const int R = a.gr_R, n_reg = R * R * R;
        const int r = threadIdx.x + (b - a.list_blocks - a.pack_blocks) * 256;
        if (r < n_reg) {
            bool any = false;
            const int b0 = r % R, b1 = (r / R) % R, b2 = r / (R * R);
            // a region is V = (H / R)^3 voxels = V / 64 consecutive 8-byte words of a level's bitfield in morton order (R = H / 8: a 64-byte line; R = H / 4: one word)
            const uint32_t vox_side = a.gr_H / (uint32_t)R, words_per_region = (vox_side * vox_side * vox_side) >> 6;
            const uint32_t words_per_level = (a.gr_H * a.gr_H * a.gr_H) >> 6;
            const uint2* g2 = reinterpret_cast<const uint2*>(a.grid);
            uint32_t acc_l[3] = {0u, 0u, 0u};   // occupancy of the region on level l (gr_C <= 3)
            for (int l = 0; l < a.gr_C; l++) {
                // on level l (R blocks over +-2^l) the region is the aligned cube of 2^j blocks per axis at R / 2 + (b - R / 2) 2^j, j = C - 1 - l — contiguous
                // words in morton order — or lies outside the level's volume, where no point can be tested on it
                const int j = a.gr_C - 1 - l, side = 1 << j;
                const int c0 = R / 2 + (b0 - R / 2) * side, c1 = R / 2 + (b1 - R / 2) * side, c2 = R / 2 + (b2 - R / 2) * side;
                if (c0 < 0 || c1 < 0 || c2 < 0 || c0 + side > R || c1 + side > R || c2 + side > R) continue;
                const uint32_t first = (uint32_t)l * words_per_level + pnm2::morton3D((uint32_t)c0, (uint32_t)c1, (uint32_t)c2) * words_per_region;
                const uint32_t n_words = words_per_region << (3 * j);
                for (uint32_t q = 0; q < n_words; q++) {
                    const uint2 v = g2[(size_t)first + q];
                    acc_l[l] |= v.x | v.y;
                }
            }
            // map L serves the rays whose mip level cannot fall below L any more (level >= mip_from_dt(dt), dt grows with t): occupied on a level >= L
            for (int L = a.gr_C - 2; L >= 0; L--) acc_l[L] |= acc_l[L + 1];
            // ... or it meets the cut box: x in (cb0, cb1), y > cb2, z in (cb4, cb5) — a superset of the reference's test (raymarching.cu:1210 compares x with
            // cut_bounds[3] where y is meant), widened by a hundredth of a region
            const float w = 2.0f * a.gr_bound / (float)R, eps = 0.01f * w;
            const float x0 = -a.gr_bound + (float)b0 * w, y0 = -a.gr_bound + (float)b1 * w, z0 = -a.gr_bound + (float)b2 * w;
            const float* cb = a.cut_bounds;
            if (x0 + w > cb[0] - eps && x0 < cb[1] + eps && y0 + w > cb[2] - eps && z0 + w > cb[4] - eps && z0 < cb[5] + eps) any = true;
            for (int L = 0; L < a.gr_C; L++) {   // one map per minimum level, behind each other
                const unsigned long long m = __ballot(any || acc_l[L] != 0u);
                if ((threadIdx.x & 63) == 0) {   // (R^3 is a multiple of 64: whole words only, whole waves inside n_reg)
                    a.grid_regions[(size_t)L * (n_reg >> 5) + (r >> 5)] = (uint32_t)m;
                    a.grid_regions[(size_t)L * (n_reg >> 5) + (r >> 5) + 1] = (uint32_t)(m >> 32);
                }
            }
        }
        return;
    }
    frame_rays_block(a, (uint32_t)(b - a.list_blocks - a.pack_blocks - a.gr_blocks));
}

extern "C" int pn_frame_create(pn_frame** out, uint32_t max_rays, uint32_t max_vtx, uint32_t max_grid_cells) {
    PN_REQUIRE(out && max_rays > 0 && max_vtx > 0 && max_grid_cells > 0);
    pn_frame* f = new pn_frame();
    memset(f, 0, sizeof(*f));
    f->max_rays = max_rays; f->max_vtx = max_vtx; f->max_cells = max_grid_cells;
    const size_t N = max_rays;
#define PN_ALLOC(ptr, bytes) PN_HIP_CHECK(hipMalloc((void**)&(ptr), (bytes)))
    PN_ALLOC(f->nears, N * 4); PN_ALLOC(f->fars, N * 4); PN_ALLOC(f->rays_t, N * 4);
    // sample slots: one per ray for the per-trip launches, 64 per wave of a fused launch (pn_trips_fused.h: one workgroup per CU) + one per position of a
    // whole-frame launch's first trip — the larger of the two
    {
        int dev_id = 0, cus = 0;
        PN_HIP_CHECK(hipGetDevice(&dev_id));
        PN_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev_id));
        f->fused_blocks = (uint32_t)std::min(std::max(cus, 1), 1024);
    }
    f->blist_cap = (uint32_t)(N / 8 + 64 + (size_t)(f->fused_blocks + 1) * 64);  // n_active <= N / 8, + one partial chunk per workgroup (pn_trips_fused.h)
    const size_t NS = N + (size_t)f->fused_blocks * PN_FUSED_WAVES * 64 + f->blist_cap;  // (per-ray slots | the fused launch's waves' slots | its first-trip positions)
    PN_ALLOC(f->xyzs, NS * 12); PN_ALLOC(f->dirs, NS * 12); PN_ALLOC(f->deltas, NS * 8); PN_ALLOC(f->sigmas, NS * 4); PN_ALLOC(f->rgbs, NS * 12);
    PN_ALLOC(f->fused_ctl, (size_t)PN_FUSED_CTL_INTS * 4);
    PN_HIP_CHECK(hipMemset(f->fused_ctl, 0, (size_t)PN_FUSED_CTL_INTS * 4));
    PN_ALLOC(f->fused_clocks, 16 * sizeof(unsigned long long));
    PN_HIP_CHECK(hipMemset(f->fused_clocks, 0, 16 * sizeof(unsigned long long)));
    f->fused_first = -1;
    PN_ALLOC(f->t_resume, N * 4);
    PN_ALLOC(f->blist, (size_t)f->blist_cap * 8);
    PN_ALLOC(f->strag, (size_t)f->blist_cap * 16);
    PN_ALLOC(f->acc_image, N * 12);
    PN_ALLOC(f->alive_a, N * 4); PN_ALLOC(f->alive_b, N * 4); PN_ALLOC(f->list, N * 4); PN_ALLOC(f->chunk_counts, (N / 256 + 4) * 4);
    PN_ALLOC(f->pig_cnt, (size_t)max_grid_cells * 4); PN_ALLOC(f->pig_bgn, (size_t)max_grid_cells * 4);
    PN_ALLOC(f->pig_cursor, (size_t)max_grid_cells * 4); PN_ALLOC(f->pig_idx, (size_t)max_vtx * 4);
    PN_ALLOC(f->side.nb_rng, ((size_t)max_grid_cells + 1) * sizeof(int2));
    f->side.nb_capacity = 27 * (int)max_vtx;
    PN_ALLOC(f->side.nb, (size_t)f->side.nb_capacity * sizeof(float4)); PN_ALLOC(f->side.rec, (size_t)max_vtx * PN_REC_FLOATS * 4);
    f->seg_cap = seg_cap_for(max_rays);
    PN_ALLOC(f->tail, (size_t)PN_SEGS * f->seg_cap * sizeof(TailEntry)); PN_ALLOC(f->tail_counts, sizeof(int) * (PN_MAX_TRIPS + 2));
    PN_ALLOC(f->list_seg, (size_t)PN_SEGS * f->seg_cap * 4); PN_ALLOC(f->active_seg, (size_t)PN_SEGS * f->seg_cap * 4);
    PN_ALLOC(f->seg_counters, (size_t)6 * PN_SEGS * PN_SEG_STRIDE * 4);
    PN_ALLOC(f->cell_bits, 2 * (((size_t)max_grid_cells + 31) / 32) * 4);
    PN_ALLOC(f->fars_eff, N * 4);
    PN_ALLOC(f->grid_regions, (size_t)PN_GRID_REGION_WORDS * 4);
    PN_ALLOC(f->trips, sizeof(PnTrip) * (PN_MAX_TRIPS + 2)); PN_ALLOC(f->dev, sizeof(PnFrameDev)); PN_ALLOC(f->cut_bounds, 6 * 4);
    f->max_groups = pn_div_up(max_rays, PN_MIN_RAY_BATCH) + 1;
    PN_ALLOC(f->groups, sizeof(PnGroup) * 2 * f->max_groups); PN_ALLOC(f->group_cnt, sizeof(int) * f->max_groups);
#undef PN_ALLOC
    PN_HIP_CHECK(hipMalloc((void**)&f->march_counters, 16 * sizeof(unsigned long long)));  // [4..15]: debug phase clocks (PN_DBG_PHASES builds)
    PN_HIP_CHECK(hipMalloc((void**)&f->stamps, sizeof(unsigned long long) * PN_TIMED_TRIPS * 3));
    PN_HIP_CHECK(hipHostMalloc((void**)&f->trips_pinned, sizeof(PnTrip) * (PN_MAX_TRIPS + 2)));
    PN_HIP_CHECK(hipHostMalloc((void**)&f->dev_pinned, sizeof(PnFrameDev)));
    PN_HIP_CHECK(hipMemset(f->dev, 0, sizeof(PnFrameDev)));
    memset(f->dev_pinned, 0, sizeof(PnFrameDev));
    *out = f;
    return PN_OK;
}

extern "C" void pn_frame_destroy(pn_frame* f) {
    if (!f) return;
    void* ptrs[] = {f->acc_image, f->nears, f->fars, f->rays_t, f->xyzs, f->dirs, f->deltas, f->sigmas, f->rgbs, f->alive_a, f->alive_b, f->list,
                    f->chunk_counts, f->pig_cnt, f->pig_bgn, f->pig_cursor, f->pig_idx, f->trips, f->dev, f->cut_bounds,
                    f->side.nb_rng, f->side.nb, f->side.rec, f->march_counters, f->tail, f->tail_counts, f->stamps,
                    f->list_seg, f->active_seg, f->seg_counters, f->cell_bits, f->fars_eff, f->grid_regions, f->groups, f->group_cnt, f->fused_ctl, f->fused_clocks, f->t_resume, f->blist, f->strag};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    for (int t = 0; t < PN_TIMED_TRIPS; t++)
        for (int e = 0; e < 3; e++) if (f->ev[t][e]) (void)hipEventDestroy(f->ev[t][e]);
    if (f->trips_pinned) (void)hipHostFree(f->trips_pinned);
    if (f->dev_pinned) (void)hipHostFree(f->dev_pinned);
    delete f;
}

static void frame_stats(pn_frame* f, int64_t* stats_host) {
    // the frame's own summary (k_frame_finish), as the render recorded it — right after graph replays too
    stats_host[0] = f->dev_pinned->stat_trips;
    stats_host[1] = f->dev_pinned->stat_samples;
    stats_host[2] = f->dev_pinned->err;
    stats_host[3] = f->dev_pinned->alive_at_exit;
    stats_host[4] = f->dev_pinned->unfinished;
}

// async_trips == 0: blocking form (trips are enqueued in batches until a readback shows no ray alive).
// async_trips  > 0: exactly that many trips are enqueued, then the epilogue and an async copy of the trip records to pinned
//                   host memory; nothing blocks the host and every call is legal inside a HIP-graph stream capture.
// aabb_static != nullptr: the undeformed render (NeRFRenderer.run_cuda, eval branch, renderer.py:267-387): no IP state, near / far from the
// given box, kernel_march_rays instead of the bending march; everything else (trip records, network, composite, compaction, epilogue)
// is the same driver.
static int render_impl(pn_frame* f, const pn_net* net, const pn_render_opts* o, const float* rays_o, const float* rays_d, uint32_t N,
                       const float* p_def, const float* p_ori, const float* F_IP, const float* dF_IP, int n_vtx, const uint8_t* bitfield, float* image,
                       float* depth, float* depth_0, float* weights_sum, int64_t* stats_host, int async_trips, void* stream,
                       const float* aabb_static = nullptr, int mode = 0 /* 0: whole frame, 1: continue the deformed frame on f, 2: continue the static one */) {
    const bool is_static = aabb_static != nullptr || mode == 2;
    const bool resume = mode != 0;
    PN_REQUIRE(f && net && o && rays_o && rays_d && bitfield && image && depth && depth_0 && weights_sum);
    PN_REQUIRE(is_static || resume || (p_def && p_ori && F_IP && dF_IP && n_vtx > 0 && (uint32_t)n_vtx <= f->max_vtx));
    PN_REQUIRE(!resume || N == f->last_N);
    PN_REQUIRE(N > 0 && N <= f->max_rays);
    PN_REQUIRE(o->num_seek_IP >= 1 && o->num_seek_IP <= 3 && o->cascade >= 1 && o->cascade <= 8 && o->max_steps <= PN_MAX_TRIPS - PN_TRIP_BATCH);
    PN_REQUIRE(async_trips >= 0 && async_trips <= PN_MAX_TRIPS);
    PN_REQUIRE(!o->fp16 || net->emb_half);  // pn_net_enable_half before an fp16 render
    hipStream_t st = (hipStream_t)stream;
    const uint32_t nblk = pn_div_up(N, 256);
    // per-trip launches use bounded grids with round-robin chunk loops (the alive count lives on the device): 32 march blocks and
    // 4 composite/compact blocks per CU (measured: 8192 march blocks is ~2 % faster than one block per 32 rays, 2048 is 6 % slower).
    // PN_MARCH_GRID / PN_TRIP_GRID override for experiments.
    static const uint32_t march_grid_cfg = pn_env_u32("PN_MARCH_GRID", 8192), trip_grid_cfg = pn_env_u32("PN_TRIP_GRID", 1024);
    // trips after the first find at most N / 8 rays in the typical frame (n_step = 8) = N / 256 chunks of 32: a grid of that size (the chunk loop takes
    // care of frames with more) instead of 8192 mostly empty workgroups per launch — what an empty captured trip costs is dispatch
    static const uint32_t march_grid_later_cfg = pn_env_u32("PN_MARCH_GRID_LATER", 0);
    const uint32_t march_grid = march_grid_cfg, trip_grid = std::min(nblk, trip_grid_cfg);
    const uint32_t march_grid_later = march_grid_later_cfg ? march_grid_later_cfg : std::max(std::min(pn_div_up(N, 256), march_grid_cfg), (uint32_t)PN_SEGS);
    // PN_SKIP_LATE_START=1 (experiment, off by default): the skip pre-pass starts its hop chain at the last lattice element before the candidates'
    // neighbourhood (pn_march_window.h).  Bit-identical in every march / frame test, but it only takes k_march_skip from 75 to 68 us on the chair (its
    // bounding box lies almost entirely within two cells of the object: the leading walks are short already) — not worth a second code path by default.
    // k_march_skip: DDA start + hop budget (pn_march_window.h: skip_empty_cells); PN_SKIP_DDA=0 walks hop by hop like rounds 1-2 (same results bit for bit)
    static const int dda_env = [] { const char* v = getenv("PN_SKIP_DDA"); return (v && v[0] == '0') ? 0 : 1; }();
    const int dda_start = g_skip_dda_override >= 0 ? g_skip_dda_override : dda_env;
    static const uint32_t skip_hop_budget = pn_env_u32("PN_SKIP_HOPS", 8);
    // a frame's first trip with ONE lane per ray in pass 1 (k_march<.., 1>) for this many rounds = visited points before a ray goes to the windows
    static const uint32_t lpr_env = pn_env_u32("PN_MARCH_LPR", 0);
    const uint32_t lpr_rounds = lpr_env ? lpr_env : (o->throughput > 0 ? (uint32_t)o->throughput : 0u);  // (PN_MARCH_LPR: experiments)
    static const uint32_t lpr_trips_env = pn_env_u32("PN_MARCH_LPR_TRIPS", 0);  // experiments
    const uint32_t lpr_trips = lpr_trips_env ? lpr_trips_env : (uint32_t)std::max(o->throughput_trips, 1);  // leading trips in that form
    static const bool split_compact = pn_env_u32("PN_SPLIT_COMPACT", 0) != 0;  // experiments: composite and compaction as two launches (rounds 1-2)
    static const uint32_t tail_grid_cfg = pn_env_u32("PN_TAIL_GRID", 1024);  // x4 waves, one unfinished ray per wave at a time
    const uint32_t tail_grid = std::max(std::min(pn_div_up(N, 4), tail_grid_cfg), (uint32_t)PN_SEGS / 4);  // every tail segment needs a wave
    // the skip pre-pass keeps the cells' emptiness bits in LDS when they fit (48 KB = 393 k cells)
    const size_t bit_words = (f->max_cells + 31) / 32;
    const bool short_rays = !is_static && !o->cut && bit_words * 8 <= 48 * 1024;  // both maps in LDS: rays end where their candidates end
    const int skip_bits_words = (short_rays || bit_words * 4 <= 48 * 1024) ? (int)bit_words : 0;
    // --cut: the region map for the skip pre-pass (MarchIO::grid_regions; pn_march_window.h: region_dda) where its assumptions hold: the top cascade level
    // spans exactly +-bound (bound == 2^(C - 1)), regions are whole 64-byte lines of the bitfield and nest on every level, the map fits
    static const bool grid_regions_off = pn_env_u32("PN_GRID_REGIONS_OFF", 0) != 0;   // A/B runs: same frames, bit for bit
    // regions of 8^3 voxels (a 64-byte line of the bitfield).  PN_REGION_SIDE=4: 4^3-voxel regions (one 8-byte word) — measured on the trex option set alone:
    // 232 against 177 us (hop by hop 285): the restart zone in front of an interesting region is longer than a 4-voxel region, so fewer runs qualify
    static const uint32_t reg_side_env = pn_env_u32("PN_REGION_SIDE", 0);
    const uint32_t reg_side = reg_side_env == 4 ? 4u : 8u;
    const uint32_t reg_R = o->grid_size / reg_side;
    const bool reg_ok = !is_static && o->cut && !grid_regions_off && dda_start && bitfield && o->grid_size % 32 == 0 && o->cascade >= 1 && o->cascade <= 3 &&
                        o->bound == (float)(1u << (o->cascade - 1)) && (reg_R / 2) % (1u << (o->cascade - 1)) == 0 &&
                        (uint64_t)reg_R * reg_R * reg_R / 32 * o->cascade <= PN_GRID_REGION_WORDS && ((uintptr_t)bitfield & 15) == 0;
    const int grid_region_words_1 = reg_ok ? (int)((uint64_t)reg_R * reg_R * reg_R / 32) : 0;   // one map
    const int grid_region_words = grid_region_words_1 * (int)o->cascade;                          // one per minimum mip level (pn_march_window.h)
    const size_t skip_lds = (size_t)skip_bits_words * 4 * (short_rays ? 2 : 1) + (size_t)grid_region_words * 4;

    if (!f->cut_bounds_valid || memcmp(f->cut_bounds_host, o->cut_bounds, sizeof(f->cut_bounds_host)) != 0) {  // uploaded only when it changes
        memcpy(f->cut_bounds_host, o->cut_bounds, sizeof(f->cut_bounds_host));
        PN_HIP_CHECK(hipMemcpyAsync(f->cut_bounds, f->cut_bounds_host, 6 * sizeof(float), hipMemcpyHostToDevice, st));
        f->cut_bounds_valid = 1;
    }

    // ray groups (ray_batch > 0): per-batch trip schedules inside the same launches; not for the static render (its trip kernel keeps one schedule)
    PN_REQUIRE(o->ray_batch == 0 || (o->ray_batch >= PN_MIN_RAY_BATCH && !is_static));
    const uint32_t group_rays = o->ray_batch > 0 ? (uint32_t)o->ray_batch : 0u;
    const uint32_t n_groups = group_rays ? pn_div_up(N, group_rays) : 0u;
    PN_REQUIRE(n_groups <= f->max_groups);
    PN_REQUIRE(!resume || group_rays == f->last_group_rays);

    // forms of the fused launch that take the frame from its first trip on (see the trip loop below) also write its epilogue: decided here, before the prologue
    static const bool fused_env0 = [] { const char* v = getenv("PN_FUSED"); return !(v && v[0] == '0'); }();
    static const int whole_env0 = [] { const char* v = getenv("PN_FUSED_WHOLE"); return !v ? -1 : (v[0] == '0' ? 0 : 1); }();
    static const int fold_env0 = [] { const char* v = getenv("PN_FUSED_FOLD"); return !v ? -1 : (v[0] == '0' ? 0 : 1); }();
    // Measured on the chair, alternating runs on one box (profiles/r04_early_finish_ab.txt): with the whole-frame launch (two lanes) 1 719 against 1 601
    // steps/s; with the folded first trip (three lanes) 1 983 / 1 950 against 2 002 / 1 981 — there the dying rays' extra loads inside the launch cost what
    // the launch off the chain saves.  So: the whole-frame form only.  PN_EARLY_FINISH=0: never; 2: with the folded first trip too (A/B)
    static const uint32_t early_env = pn_env_u32("PN_EARLY_FINISH", 1);
    const bool fused_ok0 = fused_env0 && !is_static && !group_rays && o->fused_from >= 0 && o->max_steps <= 8u * PN_FUSED_MAX_TRIPS;
    const bool want_whole = fused_ok0 && (whole_env0 < 0 ? o->fused_whole != 0 : whole_env0 != 0) && o->fused_from == 0 && !resume;
    const bool want_fold = fused_ok0 && !want_whole && (fold_env0 < 0 ? o->fused_fold != 0 : fold_env0 != 0) && o->fused_from <= 1 && !resume;
    const bool early_finish = early_env != 0 && (want_whole || (want_fold && early_env >= 2));
    const float* bbmin = f->dev->aabb;  // device addresses of struct members
    const float* bbmax = f->dev->aabb + 3;
    const int* res = f->dev->resolution;
    const int* n_grid_dev = f->dev->resolution + 3;
    int* err = &f->dev->err;
    int rc = PN_OK;
    const int swap = (o->num_seek_IP == 1) ? 1 : 0;
    if (!resume) {  // ---- prologue: tables, lists, ray state (a continuation finds all of it in the workspace)
    // two 16-bit cell counters per LDS word + the staged point-index table
    const size_t tables_lds = ((size_t)f->max_cells + 1) / 2 * sizeof(unsigned) + (size_t)f->max_vtx * sizeof(int);
    const bool large = tables_lds > 150 * 1024;  // grid too large for the one-workgroup LDS build
    // (Round 6 built the one-workgroup LDS build over the BOX of cells that hold integration points for such grids — one launch instead of eight — and measured it
    // on the trex option set while the body fits the box: 1 739 / 1 737 / 1 736 steps/s for the eight launches against 1 747 / 1 741 / 1 724, no gain: the pipeline is not bound by that
    // chain; and a body that spreads beyond the box's 64 000 cells, as bench.py's does within 250 substeps, had to stop with a capacity flag.  Removed.)
    const bool keep_tables = !is_static && o->reuse_tables && f->tables_n_vtx == n_vtx;  // staged batches of one frame: same IP state
    if (is_static) {
        k_set_aabb<<<1, 1, 0, st>>>(f->dev, aabb_static[0], aabb_static[1], aabb_static[2], aabb_static[3], aabb_static[4], aabb_static[5]);
    } else if (keep_tables) {
        // nothing: bounding box, spatial hash, candidate lists and packed IP records of the previous render on this workspace stay
    } else if (!large) {
        // dynamic LDS above 64 KB has to be opted into (gfx950: 160 KB per workgroup); the attribute is per DEVICE, so the cache is too
        static size_t tables_lds_set[PN_MAX_DEVICES] = {0};
        int dev_id = 0;
        PN_HIP_CHECK(hipGetDevice(&dev_id));
        if (dev_id < 0 || dev_id >= PN_MAX_DEVICES || tables_lds > tables_lds_set[dev_id]) {
            PN_HIP_CHECK(hipFuncSetAttribute((const void*)k_frame_tables<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tables_lds));
            if (dev_id >= 0 && dev_id < PN_MAX_DEVICES) tables_lds_set[dev_id] = tables_lds;
        }
        k_frame_tables<false><<<1, 1024, tables_lds, st>>>(p_def, n_vtx, o->cut, o->bound, o->hash_grid_size, (int)f->max_cells, f->dev, f->pig_cnt,
                                                       f->pig_bgn, f->pig_idx, f->pig_cursor, f->cell_bits);
    } else {
        k_frame_tables<true><<<1, 1024, 0, st>>>(p_def, n_vtx, o->cut, o->bound, o->hash_grid_size, (int)f->max_cells, f->dev, f->pig_cnt, f->pig_bgn,
                                                 f->pig_idx, f->pig_cursor, f->cell_bits);
        rc = pig_build(n_vtx, (int)f->max_cells, n_grid_dev, p_def, bbmin, o->hash_grid_size, res, f->pig_cnt, f->pig_bgn, f->pig_idx, f->pig_cursor, err, st);
        if (rc) return rc;
    }
    FramePrologue fp;
    memset(&fp, 0, sizeof(fp));
    if (!is_static && !keep_tables) {
        f->tables_n_vtx = n_vtx;
        fp.n_grid_max = (int)f->max_cells; fp.n_grid_dev = n_grid_dev; fp.res = res; fp.pig_cnt = f->pig_cnt; fp.pig_bgn = f->pig_bgn; fp.pig_idx = f->pig_idx;
        fp.p_def = p_def; fp.swap = swap; fp.nb_rng = f->side.nb_rng; fp.nb = f->side.nb; fp.nb_capacity = f->side.nb_capacity; fp.cell_bits = f->cell_bits;
        fp.list_blocks = (int)std::min(pn_div_up((uint64_t)f->max_cells, 32), 2048u);
        fp.pack_blocks = (int)pn_div_up((uint64_t)n_vtx * PN_REC_FLOATS, 256);
        fp.n_vtx = n_vtx; fp.p_ori = p_ori; fp.F_IP = F_IP; fp.dF_IP = dF_IP; fp.rec = f->side.rec;
    }
    fp.rays_o = rays_o; fp.rays_d = rays_d; fp.dev = f->dev; fp.N = N; fp.min_near = o->min_near; fp.nears = f->nears; fp.fars = f->fars; fp.rays_t = f->rays_t;
    fp.trips = f->trips; fp.tail_counts = f->tail_counts; fp.seg_counters = f->seg_counters; fp.n_trip_records = PN_MAX_TRIPS + 2; fp.alive = f->alive_a;
    fp.weights_sum = weights_sum; fp.depth_0 = depth_0; fp.image = f->acc_image; fp.groups = group_rays ? f->groups : nullptr; fp.group_cnt = f->group_cnt;
    fp.group_rays = group_rays; fp.n_groups = n_groups; fp.chunk_words = f->chunk_counts;
    fp.early_finish = early_finish ? 1 : 0; fp.bg = o->bg_color; fp.image_out = image; fp.depth_out = depth;
    {   // pn_render_opts.ray_tile_w; PN_RAY_TILE_OFF=1 keeps the row-major order (A/B runs: same frames, bit for bit)
        static const bool tile_off = pn_env_u32("PN_RAY_TILE_OFF", 0) != 0;
        static const uint32_t lw = std::min(pn_env_u32("PN_RAY_TILE_LOG2W", 4), 5u);  // experiments: 8 x 8 (3), 32 x 2 (5) pixel tiles
        const uint32_t tw = o->ray_tile_w > 0 ? (uint32_t)o->ray_tile_w : 0u;
        fp.tile_lw = lw;
        fp.tile_w = (!tile_off && !is_static && !group_rays && tw && tw % (1u << lw) == 0 && N % ((64u >> lw) * tw) == 0) ? tw : 0u;
    }
    // both cell maps of a workgroup in LDS while it builds its lists (up to 64 KB = 262 k cells; beyond that straight to global memory, where
    // the maps then span enough cache lines for the atomics not to queue)
    fp.lds_words = (fp.list_blocks > 0 && bit_words * 8 <= 64 * 1024) ? (int)bit_words : 0;
    if (grid_region_words > 0) {
        fp.grid = bitfield; fp.grid_regions = f->grid_regions; fp.gr_R = (int)reg_R; fp.gr_C = (int)o->cascade; fp.gr_H = o->grid_size;
        fp.gr_bound = o->bound; fp.cut_bounds = f->cut_bounds; fp.gr_blocks = (int)pn_div_up((uint64_t)grid_region_words_1 * 32, 256);
    }
    k_frame_prologue<<<(uint32_t)(fp.list_blocks + fp.pack_blocks + fp.gr_blocks) + nblk, 256, (size_t)fp.lds_words * 8, st>>>(fp);
    PN_LAUNCH_CHECK();
    }
    pnm2::March2Tables tb{f->side.nb_rng, f->side.nb, (const float4*)f->side.rec};

    pnm::MarchParams mp = make_march_params(f->pig_cnt, f->pig_bgn, f->pig_idx, n_vtx, 0, p_def, p_ori, F_IP, dF_IP, o->max_iter_num, bbmin, bbmax,
                                            o->hash_grid_size, res, o->num_seek_IP, o->IP_dx, o->cut, f->cut_bounds, f->rays_t, rays_o, rays_d,
                                            o->bound, o->dt_gamma, o->max_steps, o->cascade, o->grid_size, bitfield, f->fars, err);
    mp.stats = (f->march_counters_on & 1) ? f->march_counters : nullptr;
    // a continuation picks up at the record the last compaction wrote; the trip count comes from the device (through the pinned copy the
    // previous render made at its end), not from host bookkeeping: the previous render may have been a graph replay
    int t = resume ? f->dev_pinned->trips_run : 0;
    PN_REQUIRE(t >= 0 && t <= PN_MAX_TRIPS);
    bool done = false;
    // The trips from `fuse_from` on as ONE launch (pn_trips_fused.h) where that form applies: a deformed frame with one trip schedule, max_steps within
    // the fused kernel's trip table.  pn_render_opts.fused_from: the first trip to run fused (0: 1 — right behind the frame's first trip; a scene whose
    // later trips still have more than N / 8 rays alive — the trex option set's second — names a later one; < 0: never).  PN_FUSED=0 switches it off (A/B).
    static const bool fused_env = [] { const char* v = getenv("PN_FUSED"); return !(v && v[0] == '0'); }();
    static const uint32_t fused_grid_env = pn_env_u32("PN_FUSED_GRID", 0);
    const bool fused_ok = fused_env && !is_static && !group_rays && o->fused_from >= 0 && o->max_steps <= 8u * PN_FUSED_MAX_TRIPS;
    const int fuse_from = fused_ok ? std::max(o->fused_from, 1) : PN_MAX_TRIPS + 1;
    int add_fused = 0;
    f->fused_first = -1;
    // ... and the WHOLE frame behind the skip pre-pass as one launch (pn_render_opts.fused_whole with fused_from == 0; pn_trips_fused.h, WHOLE): the launch checks on the
    // device that at most N / 8 rays have anything to march (then every trip after the first marches 8 samples per ray whatever the first one finds) and
    // does nothing otherwise — the blocking driver then goes on trip by trip as above, a fixed-trip render is left to pn_render_continue.
    // PN_FUSED_WHOLE=0 / 1: never / wherever fused_from == 0 allows it (A/B runs).
    static const uint32_t a_rounds_env = pn_env_u32("PN_FUSED_AROUNDS", 0);
    bool whole_try = want_whole && t == 0;
    // ... or the first trip's NETWORK, COMPOSITE and COMPACTION inside that launch (pn_render_opts.fused_fold with fused_from <= 1; pn_trips_fused.h, FOLD): the
    // march of the first trip stays what it is — skip pre-pass, one lane per ray / windows, tail pass, on every CU — and leaves the trip's segmented sample list;
    // the launch runs network tiles over it, composites, and takes the survivors on.  Four launches fewer on a frame's chain (k_list_pack, k_nerf_forward,
    // k_composite, k_compact).  Applies when at most N / 8 rays found a sample (checked on the device); otherwise as with fused_whole.  PN_FUSED_FOLD=0 / 1: A/B.
    bool fold_try = want_fold && t == 0;
    bool finished_in_launch = false;   // the frame's epilogue was written by the fused launch (early_finish)
    bool skip_done = resume && f->skip_done != 0;
    bool head_marched = resume && f->head_marched != 0 && t == 0;   // the first trip's march has run: its per-trip launches go on behind it
    if (!resume) { f->skip_done = 0; f->head_marched = 0; }
    // measurement mode (march_counters bit 1): point e (0 before the march, 1 behind it, 2 behind the network) of launch group `trip`
    auto time_mark = [&](int trip, int e) -> int {
        if (!(f->march_counters_on & 2) || trip >= PN_TIMED_TRIPS) return PN_OK;
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        PN_HIP_CHECK(hipStreamIsCapturing(st, &cs));
        const bool stamp = cs != hipStreamCaptureStatusNone;   // inside a capture: stamp kernels (events recorded in a graph cannot be timed)
        f->stamped = stamp ? 1 : 0;
        if (stamp) {
            k_stamp<<<1, 1, 0, st>>>(f->stamps + trip * 3 + e);
        } else {
            if (!f->ev[trip][e]) PN_HIP_CHECK(hipEventCreate(&f->ev[trip][e]));
            PN_HIP_CHECK(hipEventRecord(f->ev[trip][e], st));
        }
        if (e == 2) f->timed_trips = std::max(f->timed_trips, trip + 1);
        return PN_OK;
    };
    int* const seg_tail = f->seg_counters;
    int* const seg_samp = seg_tail + PN_SEGS * PN_SEG_STRIDE;
    int* const seg_emit = seg_samp + PN_SEGS * PN_SEG_STRIDE;
    int* const seg_curs = seg_emit + PN_SEGS * PN_SEG_STRIDE;
    int* const seg_back = seg_curs + PN_SEGS * PN_SEG_STRIDE;
    int* const seg_active = seg_back + PN_SEGS * PN_SEG_STRIDE;
    // trip 0 keeps its skip pre-pass state in f->t_resume and lists the slots worth marching in f->active_seg
    auto make_io = [&](int tt) {
        int* cur = (tt & 1) ? f->alive_b : f->alive_a;
        const bool lpr = (uint32_t)tt < lpr_trips && lpr_rounds > 0;
        return MarchIO{0, 0, cur, f->xyzs, f->dirs, f->deltas, nullptr, f->trips + tt, f->list, (tt == 0) ? f->t_resume : nullptr,
                       f->tail, seg_tail, seg_back, seg_curs, (int)f->seg_cap, lpr ? (int)lpr_rounds : (int)march_tail_rounds(tt),
                       (tt == 0) ? f->active_seg : nullptr, (tt == 0) ? seg_active : nullptr,
                       (int)f->seg_cap, f->list_seg, seg_samp, (int)f->seg_cap, seg_emit, f->cell_bits, skip_bits_words,
                       short_rays ? f->cell_bits + bit_words : nullptr, short_rays ? f->fars_eff : nullptr,
                       group_rays ? f->groups + (size_t)(tt & 1) * f->max_groups : nullptr, group_rays, lpr ? 1 : 0, dda_start,
                       (int)skip_hop_budget, grid_region_words > 0 ? f->grid_regions : nullptr, grid_region_words, (int)reg_R};
    };
    while (!done && t < PN_MAX_TRIPS) {
        const bool whole = whole_try;
        whole_try = false;
        const bool fold = !whole && fold_try && t == 0 && !head_marched;
        if (fold) fold_try = false;
        if (fold) {  // the first trip's march, as its per-trip launches would run it (no k_list_pack: the launch reads the segments)
            const MarchIO io0 = make_io(0);
            if ((rc = time_mark(0, 0))) return rc;
            if (!skip_done) { k_march_skip<<<nblk, 256, skip_lds, st>>>(mp, tb, io0); skip_done = true; f->skip_done = 1; }
            pnm::MarchParams mq0 = mp;
            if (short_rays) mq0.fars = f->fars_eff;
            launch_march(o->num_seek_IP, std::max(std::min(pn_div_up(N, 32), march_grid), (uint32_t)PN_SEGS), tail_grid, st, mq0, tb, io0);
            head_marched = true;
            f->head_marched = 1;
            if ((rc = time_mark(0, 1)) || (rc = time_mark(0, 2))) return rc;
        }
        if (whole || fold || t >= fuse_from) {
            FusedArgs fa;
            memset(&fa, 0, sizeof(fa));
            fa.lv = (const PnFusedLevel*)(o->fp16 ? net->fused_levels : net->byte_levels); fa.emb = net->embeddings; fa.emb_h = (const uint32_t*)net->emb_half; fa.emb_bytes = net->n_entries * 4u;
            fa.wimg_g = (const uint4*)(o->fp16 ? net->whalf : (net->x_ok ? net->wx : net->wsplit)); fa.net_bound = net->bound; fa.net_inv2b = 1.0f / (2 * net->bound); fa.density_scale = o->density_scale;
            fa.x_scales = net->x_scales;
            fa.trips = f->trips + t; fa.N_rays = N; fa.max_steps = o->max_steps; fa.T_thresh = o->T_thresh;
            fa.alive = (t & 1) ? f->alive_b : f->alive_a;
            fa.rays_t = f->rays_t; fa.weights_sum = weights_sum; fa.depth = depth_0; fa.image = f->acc_image;
            fa.xyzs = f->xyzs; fa.dirs = f->dirs; fa.deltas = f->deltas; fa.sigmas = f->sigmas; fa.rgbs = f->rgbs;
            fa.ctl = f->fused_ctl; fa.dev = f->dev; fa.tail_diag = f->tail_counts + t;
            fa.clocks = (f->march_counters_on & 4) ? f->fused_clocks : nullptr;
            if (whole) {
                fa.active = f->active_seg; fa.active_counts = seg_active; fa.active_seg_cap = (int)f->seg_cap; fa.t_resume = f->t_resume;
                fa.blist = f->blist; fa.strag = f->strag; fa.blist_cap = f->blist_cap;
                fa.a_rounds = a_rounds_env ? (int)a_rounds_env : 24;
            }
            if (fold) {
                fa.list_seg = f->list_seg; fa.samp_counts = seg_samp; fa.list_seg_cap = (int)f->seg_cap; fa.seg_tail = seg_tail; fa.seg_back = seg_back;
                fa.blist = f->blist; fa.strag = f->strag; fa.blist_cap = f->blist_cap;
            }
            if ((whole || fold) && early_finish) {
                fa.finalize = 1; fa.bg = o->bg_color; fa.nears = f->nears; fa.fars_full = f->fars; fa.image_out = image; fa.depth_out = depth;
            }
            const int tb_idx = fold ? 1 : t;   // launch group the fused launch is timed as (fold: behind the first trip's march)
            pnm::MarchParams mq = mp;
            if (short_rays) mq.fars = f->fars_eff;  // written by trip 0's k_march_skip
            if ((rc = time_mark(tb_idx, 0))) return rc;  // the whole launch is bracketed like a trip's march group (its network share comes from the phase clocks)
            if (whole) {  // the skip pre-pass: per-ray resume points, shortened ends, the active list (one lane per ray)
                const MarchIO io0 = make_io(0);
                k_march_skip<<<nblk, 256, skip_lds, st>>>(mp, tb, io0);
                skip_done = true;
                f->skip_done = 1;
            }
            const uint32_t blocks = fused_grid_env ? std::min(fused_grid_env, f->fused_blocks) : (o->fused_grid > 0 ? std::min((uint32_t)o->fused_grid, f->fused_blocks) : f->fused_blocks);
            rc = launch_trips_fused(o->num_seek_IP, o->max_iter_num > 1, o->fp16 ? 1 : (net->x_ok ? 2 : 0), whole ? 1 : (fold ? 2 : 0), blocks, st, mq, tb, fa);
            if (rc) return rc;
            if ((rc = time_mark(tb_idx, 1)) || (rc = time_mark(tb_idx, 2))) return rc;
            f->fused_first = tb_idx;
            f->fused_mode = whole ? 1 : (fold ? 2 : 0);
            if (async_trips > 0) {  // how many trips it ran only the device knows: k_frame_finish adds them — or the launch itself closed the frame's books
                add_fused = 1;
                finished_in_launch = fa.finalize != 0;   // (if it steps aside the frame is an unfinished one at trip 0: pn_render_continue)
                break;
            }
            PN_HIP_CHECK(hipMemcpyAsync(f->dev_pinned, f->dev, sizeof(PnFrameDev), hipMemcpyDeviceToHost, st));
            PN_HIP_CHECK(hipMemcpyAsync(f->trips_pinned + t, f->trips + t, sizeof(PnTrip), hipMemcpyDeviceToHost, st));
            PN_HIP_CHECK(hipStreamSynchronize(st));
            if (f->dev_pinned->fused_trips > 0) {  // ran until no ray was alive (or max_steps)
                t += f->dev_pinned->fused_trips;
                done = true;
                finished_in_launch = fa.finalize != 0;
                break;
            }
            if (f->trips_pinned[t].n_alive <= 0) { done = true; break; }
            // not applicable at this trip (more than N / 8 rays alive: n_step < 8): one trip of the per-trip launches, then again
            f->fused_first = -1;
            f->fused_mode = 0;
        }
        const int batch = async_trips > 0 ? (fused_ok ? fuse_from - t : async_trips) : (fused_ok ? std::max(fuse_from - t, 1) : PN_TRIP_BATCH);
        for (int k = 0; k < batch; k++, t++) {
            // margin trips: a fixed-trip render (captured graphs) carries PN_TRIP_MARGIN more trips than the frame it was sized on needed, and they
            // find no ray (or a few hundred stragglers).  What they cost is the dispatch of their launches' workgroups, so they get small grids —
            // the chunk loops take care of whatever is alive — and the two-launch composite / compaction (the fused one needs a workgroup per chunk)
            const bool margin = !fused_ok && async_trips > PN_TRIP_MARGIN && k >= async_trips - PN_TRIP_MARGIN && !resume;
            int* cur = (t & 1) ? f->alive_b : f->alive_a;
            int* nxt = (t & 1) ? f->alive_a : f->alive_b;
            // trip 0 (every ray, one sample each) is dominated by rays crossing IP-free cells: a one-lane-per-ray pre-pass
            // fast-forwards them (its per-ray resume point: f->t_resume) and lists the slots that still have work (f->active_seg)
            MarchIO io = make_io(t);
            const bool timed = (f->march_counters_on & 2) && t < PN_TIMED_TRIPS;
            bool stamp = false;
            if (timed) {  // measurement mode: the two heavy launch groups of each trip are bracketed on the launch stream
                hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
                PN_HIP_CHECK(hipStreamIsCapturing(st, &cs));
                stamp = cs != hipStreamCaptureStatusNone;   // inside a capture: stamp kernels (events recorded in a graph cannot be timed)
                f->stamped = stamp ? 1 : 0;
                if (stamp) {
                    k_stamp<<<1, 1, 0, st>>>(f->stamps + t * 3);
                } else {
                    for (int e = 0; e < 3; e++)
                        if (!f->ev[t][e]) PN_HIP_CHECK(hipEventCreate(&f->ev[t][e]));
                    PN_HIP_CHECK(hipEventRecord(f->ev[t][0], st));
                }
            }
            if (is_static) {
                k_march_static_trip<<<trip_grid, 256, 0, st>>>(f->trips + t, cur, f->rays_t, rays_o, rays_d, o->bound, o->dt_gamma, o->max_steps, o->cascade,
                                                               o->grid_size, bitfield, f->fars, f->xyzs, f->dirs, f->deltas, f->list);
            } else {
                if (!(t == 0 && head_marched)) {   // (a first trip whose march has already run — a fused launch that stepped aside — goes on with its sample list)
                if (io.t_resume && !skip_done) { k_march_skip<<<nblk, 256, skip_lds, st>>>(mp, tb, io); skip_done = true; f->skip_done = 1; }
                pnm::MarchParams mq = mp;
                if (short_rays) mq.fars = f->fars_eff;  // written by trip 0's k_march_skip
                launch_march(o->num_seek_IP, t == 0 ? std::max(std::min(pn_div_up(N, 32), march_grid), (uint32_t)PN_SEGS) : (margin ? 2u * PN_SEGS : march_grid_later),
                             margin ? (uint32_t)PN_SEGS : tail_grid, st, mq, tb, io);
                }
                if (t == 0) k_list_pack<<<PN_SEGS, 256, 0, st>>>(f->trips + t, seg_samp, f->list_seg, (int)f->seg_cap, f->list);  // the only list trip
            }
            if (timed && stamp) k_stamp<<<1, 1, 0, st>>>(f->stamps + t * 3 + 1);
            else if (timed) PN_HIP_CHECK(hipEventRecord(f->ev[t][1], st));
            rc = pn_nerf_forward_launch(net, f->xyzs, f->dirs, f->list, &f->trips[t].n_samples, N, o->density_scale, f->sigmas, f->rgbs, o->fp16, st, margin ? 64u : 0u);
            if (rc) return rc;
            if (timed) {
                if (stamp) k_stamp<<<1, 1, 0, st>>>(f->stamps + t * 3 + 2);
                else PN_HIP_CHECK(hipEventRecord(f->ev[t][2], st));
                f->timed_trips = t + 1;
            }
            PnGroup* g_cur = group_rays ? f->groups + (size_t)(t & 1) * f->max_groups : nullptr;
            PnGroup* g_nxt = group_rays ? f->groups + (size_t)((t + 1) & 1) * f->max_groups : nullptr;
            const uint32_t pair_grid = margin ? std::min(trip_grid, 64u) : trip_grid;
            // a frame's first trip has every ray alive (2 500 chunks at 800x800): five rounds of the bounded fused kernel (31 us) cost more than the two
            // launches (25 us), which poll nothing; PN_CC_TRIP0=1 keeps the fused form there
            static const bool cc_trip0 = pn_env_u32("PN_CC_TRIP0", 0) != 0;
            if (!is_static && !split_compact && !margin && (t > 0 || cc_trip0)) {
                // a bounded grid with chunk loops: all of a launch's workgroups can be resident at once, whatever order the XCDs start them in (see the kernel)
#define PN_CC_LAUNCH(R_)                                                                                                                                   \
    k_composite_compact<R_><<<std::min(cc_grid, pn_div_up(N, 256 * R_)), 256, 0, st>>>(o->T_thresh, cur, nxt, f->rays_t, f->sigmas, f->rgbs, f->deltas, weights_sum, depth_0,          \
                                                                     f->acc_image, f->trips + t, f->trips + t + 1, (unsigned*)f->chunk_counts, (uint32_t)t + 1, N, \
                                                                     o->max_steps, 1, f->seg_counters, f->tail_counts + t, g_cur, g_nxt, f->group_cnt, group_rays,  \
                                                                     n_groups, err, cc_poll_cap)
                static const uint32_t cc_grid = pn_env_u32("PN_CC_GRID", 512);
                static const uint32_t cc_poll_cap = 1u << std::min(pn_env_u32("PN_CC_POLL_LOG2", 20), 30u);
                static const uint32_t cc_r0 = pn_env_u32("PN_CC_R0", 1);  // alive positions per thread on a frame's first trip; measured 20.0 / 21.4 / 27.9 us for 1 / 2 / 4
                if (t == 0 && cc_r0 >= 4) PN_CC_LAUNCH(4);
                else if (t == 0 && cc_r0 == 2) PN_CC_LAUNCH(2);
                else PN_CC_LAUNCH(1);
#undef PN_CC_LAUNCH
            } else {
            k_composite<<<pair_grid, 256, 0, st>>>(0, 0, o->T_thresh, cur, f->rays_t, f->sigmas, f->rgbs, f->deltas, weights_sum, depth_0, f->acc_image,
                                                   f->trips + t, f->chunk_counts, g_cur, group_rays, n_groups > 1 ? f->group_cnt : nullptr);
            k_compact<<<pair_grid, 256, 0, st>>>(cur, 0, f->chunk_counts, nxt, nullptr, f->trips + t, f->trips + t + 1, N, o->max_steps, is_static ? 0 : 1,
                                                 is_static ? nullptr : f->seg_counters, f->tail_counts + t, g_cur, g_nxt, f->group_cnt, group_rays, n_groups);
            }
        }
        PN_LAUNCH_CHECK();
        if (async_trips > 0) {
            if (fused_ok) continue;  // the fused launch follows
            break;
        }
        if (fused_ok && t >= fuse_from) continue;  // no read-back: the fused launch finds out by itself whether anything is alive
        // one small readback per batch decides whether more trips are needed (the reference syncs every trip)
        PN_HIP_CHECK(hipMemcpyAsync(f->trips_pinned + t, f->trips + t, sizeof(PnTrip), hipMemcpyDeviceToHost, st));
        PN_HIP_CHECK(hipStreamSynchronize(st));
        done = f->trips_pinned[t].n_alive <= 0;
    }
    if (!finished_in_launch)
        k_frame_finish<<<nblk, 256, 0, st>>>(N, o->bg_color, f->nears, f->fars, weights_sum, depth_0, f->acc_image, image, depth, f->trips, f->dev, t, add_fused);
    PN_LAUNCH_CHECK();
    f->last_trips = t;
    f->last_N = N;
    f->last_group_rays = group_rays;
    // the frame record (trips run, summary of the trip records, flags: written by k_frame_finish) always travels to pinned host memory with one small
    // async copy: pn_render_status / pn_render_continue read it once the caller knows the render has completed
    PN_HIP_CHECK(hipMemcpyAsync(f->dev_pinned, f->dev, sizeof(PnFrameDev), hipMemcpyDeviceToHost, st));
    if (async_trips == 0 && stats_host) {
        PN_HIP_CHECK(hipStreamSynchronize(st));
        frame_stats(f, stats_host);
    }
    return PN_OK;
}

extern "C" int pn_render_deformed(pn_frame* f, const pn_net* net, const pn_render_opts* o, const float* rays_o, const float* rays_d, uint32_t N,
                                  const float* p_def, const float* p_ori, const float* F_IP, const float* dF_IP, int n_vtx,
                                  const uint8_t* bitfield, float* image, float* depth, float* depth_0, float* weights_sum, int64_t* stats_host,
                                  void* stream) {
    return render_impl(f, net, o, rays_o, rays_d, N, p_def, p_ori, F_IP, dF_IP, n_vtx, bitfield, image, depth, depth_0, weights_sum, stats_host, 0, stream);
}

extern "C" int pn_render_deformed_async(pn_frame* f, const pn_net* net, const pn_render_opts* o, const float* rays_o, const float* rays_d,
                                        uint32_t N, const float* p_def, const float* p_ori, const float* F_IP, const float* dF_IP, int n_vtx,
                                        const uint8_t* bitfield, float* image, float* depth, float* depth_0, float* weights_sum, int n_trips,
                                        void* stream) {
    PN_REQUIRE(n_trips > 0);
    return render_impl(f, net, o, rays_o, rays_d, N, p_def, p_ori, F_IP, dF_IP, n_vtx, bitfield, image, depth, depth_0, weights_sum, nullptr, n_trips,
                       stream);
}

extern "C" int pn_render_static(pn_frame* f, const pn_net* net, const pn_render_opts* o, const float* rays_o, const float* rays_d, uint32_t N,
                                const float* aabb_host, const uint8_t* bitfield, float* image, float* depth, float* depth_0, float* weights_sum,
                                int64_t* stats_host, int n_trips, void* stream) {
    PN_REQUIRE(aabb_host && n_trips >= 0);
    return render_impl(f, net, o, rays_o, rays_d, N, nullptr, nullptr, nullptr, nullptr, 0, bitfield, image, depth, depth_0, weights_sum, stats_host, n_trips,
                       stream, aabb_host);
}

extern "C" int pn_render_continue(pn_frame* f, const pn_net* net, const pn_render_opts* o, const float* rays_o, const float* rays_d, uint32_t N,
                                  const uint8_t* bitfield, float* image, float* depth, float* depth_0, float* weights_sum, int64_t* stats_host, int n_trips,
                                  int is_static, void* stream) {
    PN_REQUIRE(n_trips >= 0);
    return render_impl(f, net, o, rays_o, rays_d, N, nullptr, nullptr, nullptr, nullptr, 0, bitfield, image, depth, depth_0, weights_sum, stats_host, n_trips,
                       stream, nullptr, is_static ? 2 : 1);
}

extern "C" int pn_frame_march_counters(pn_frame* f, int enable, uint64_t* counters_host, void* stream) {
    PN_REQUIRE(f);
    hipStream_t st = (hipStream_t)stream;
    if (counters_host) {
        PN_HIP_CHECK(hipMemcpyAsync(counters_host, f->march_counters, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
#if PN_DBG_PHASES || PN_DBG_STATS   // (timing / counting builds only: tools/build_variant.py name -DPN_DBG_STATS=1 prints the debug slots [4..15])
        {
            unsigned long long ph[16];
            PN_HIP_CHECK(hipMemcpy(ph, f->march_counters, sizeof(ph), hipMemcpyDeviceToHost));
            fprintf(stderr, "phases:");
            for (int i = 4; i < 16; i++) fprintf(stderr, " %llu", ph[i]);
            fprintf(stderr, "\n");
        }
#endif
        PN_HIP_CHECK(hipStreamSynchronize(st));
    }
    if ((enable & 1) && !(f->march_counters_on & 1)) PN_HIP_CHECK(hipMemsetAsync(f->march_counters, 0, 16 * sizeof(unsigned long long), st));
    f->march_counters_on = enable & 7;  // bit 0: work counters, bit 1: per-trip event timing, bit 2: phase clocks of the fused launches
    return PN_OK;
}

// Phase clocks of the fused launches on `f` (march_counters bit 2): shader-clock cycles summed over waves for {refill, march window round, 64-lane
// windows, network, composite}, wave-rounds, waves; first_trip_out: the trip the last render started fusing at (-1: it did not).  reset != 0 zeroes them.
extern "C" int pn_frame_fused_clocks(pn_frame* f, uint64_t* clocks_host, int* first_trip_out, int reset, void* stream) {
    PN_REQUIRE(f);
    hipStream_t st = (hipStream_t)stream;
    if (clocks_host) {
        PN_HIP_CHECK(hipMemcpyAsync(clocks_host, f->fused_clocks, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
        PN_HIP_CHECK(hipStreamSynchronize(st));
        clocks_host[15] = f->fused_first >= 0 ? (uint64_t)f->fused_mode : 0u;   // form of the last render's fused launch (host-side knowledge)
    }
    if (first_trip_out) *first_trip_out = f->fused_first;
    if (reset) PN_HIP_CHECK(hipMemsetAsync(f->fused_clocks, 0, 16 * sizeof(unsigned long long), st));
    return PN_OK;
}

extern "C" int pn_frame_trip_times(pn_frame* f, float* march_ms_host, float* network_ms_host, int max_trips, int* n_trips_out, void* stream) {
    PN_REQUIRE(f && march_ms_host && network_ms_host && n_trips_out && max_trips >= 0);
    PN_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    const int n = f->timed_trips < max_trips ? f->timed_trips : max_trips;
    if (f->stamped) {
        static unsigned long long host[PN_TIMED_TRIPS * 3];
        PN_HIP_CHECK(hipMemcpy(host, f->stamps, sizeof(unsigned long long) * (size_t)n * 3, hipMemcpyDeviceToHost));
        for (int t = 0; t < n; t++) {  // s_memrealtime ticks at 100 MHz: 1 tick = 1e-5 ms
            march_ms_host[t] = (float)((double)(host[t * 3 + 1] - host[t * 3]) * 1e-5);
            network_ms_host[t] = (float)((double)(host[t * 3 + 2] - host[t * 3 + 1]) * 1e-5);
        }
    } else {
        for (int t = 0; t < n; t++) {
            PN_HIP_CHECK(hipEventElapsedTime(march_ms_host + t, f->ev[t][0], f->ev[t][1]));
            PN_HIP_CHECK(hipEventElapsedTime(network_ms_host + t, f->ev[t][1], f->ev[t][2]));
        }
    }
    *n_trips_out = n;
    return PN_OK;
}

extern "C" int pn_frame_trip_records(pn_frame* f, int* records_host, int* tail_counts_host, int max_trips, void* stream) {
    PN_REQUIRE(f && records_host && tail_counts_host && max_trips >= 0);
    PN_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    const int n = max_trips < PN_MAX_TRIPS ? max_trips : PN_MAX_TRIPS;
    std::vector<PnTrip> rec((size_t)n);
    PN_HIP_CHECK(hipMemcpy(rec.data(), f->trips, sizeof(PnTrip) * (size_t)n, hipMemcpyDeviceToHost));
    for (int i = 0; i < n; i++) {
        int* r = records_host + 5 * i;
        r[0] = rec[i].n_alive; r[1] = rec[i].n_step; r[2] = rec[i].step_base; r[3] = rec[i].n_samples; r[4] = rec[i].dense ? rec[i].n_emitted : -1;
    }
    PN_HIP_CHECK(hipMemcpy(tail_counts_host, f->tail_counts, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost));
    return PN_OK;
}

__global__ void k_reset_unfinished(PnFrameDev* dev) { dev->unfinished = 0; }

extern "C" int pn_frame_reset_unfinished(pn_frame* f, void* stream) {
    PN_REQUIRE(f);
    k_reset_unfinished<<<1, 1, 0, (hipStream_t)stream>>>(f->dev);
    PN_LAUNCH_CHECK();
    return PN_OK;
}

extern "C" int pn_render_status(pn_frame* f, int64_t* stats_host, int synchronize, void* stream) {
    PN_REQUIRE(f && stats_host);
    if (synchronize) PN_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    frame_stats(f, stats_host);
    return PN_OK;
}
