// Training-side ray ops for gfx950 (off the simulate-and-render hot path).
//
// Reference: raymarching/src/raymarching.cu:314-497 (kernel_march_rays_train), :503-601 (kernel_composite_rays_train_forward),
// :604-700 (kernel_composite_rays_train_backward) — paths relative to /root/reference.  Compiled with -ffp-contract=off like the
// rest of the ray side: sample positions, deltas and ray rows are bit-identical to the oracle's.
//
// march_rays_train is three launches instead of the reference's one: the reference hands out point ranges with
// atomicAdd(counter, num_steps), which makes both the ray-row order and the point layout a race.  Here
//   k_train_count   one lane per ray: first pass of the reference kernel (:357-401) -> rays[n] = (n, -, count)
//   k_train_scan    ONE workgroup of 1024 lanes: exclusive prefix sum of the counts (wave shuffles + 16 wave totals in LDS,
//                   1024-ray tiles with a running carry) -> rays[n,1]; counter += (total, N)
//   k_train_write   one lane per ray: second pass (:420-480) into its own range
// so samples of consecutive rays are consecutive in memory (coalesced composite reads) and every run gives the same bytes.
#include <float.h>

#include "pn_march_math.h"

namespace {
using namespace pnm;

// the loop both passes share; WRITE = false only counts occupied steps
template <bool WRITE>
__device__ __forceinline__ uint32_t train_march_pass(const float* __restrict__ ro, const float* __restrict__ rd, float t0, float far, uint32_t limit,
                                                     float bound, float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H,
                                                     const uint8_t* __restrict__ grid, float* xyzs, float* dirs, float* deltas) {
    const float ox = ro[0], oy = ro[1], oz = ro[2];
    const float dx = rd[0], dy = rd[1], dz = rd[2];
    const float rdx = 1 / dx, rdy = 1 / dy, rdz = 1 / dz;
    const float rH = 1 / (float)H;
    const float H3 = (float)(H * H * H);
    const float dt_min = 2 * 1.73205080757f / max_steps;
    const float dt_max = 2 * 1.73205080757f * (1 << (C - 1)) / H;
    float t = t0, last_t = t0;
    uint32_t step = 0;
    while (t < far && step < limit) {
        const float x = clampf(ox + t * dx, -bound, bound);
        const float y = clampf(oy + t * dy, -bound, bound);
        const float z = clampf(oz + t * dz, -bound, bound);
        const float dt = clampf(t * dt_gamma, dt_min, dt_max);
        const int level = max(mip_from_pos(x, y, z, (float)C), mip_from_dt(dt, (float)H, (float)C));
        const float mip_bound = fminf(scalbnf(1.0f, level), bound);
        const float mip_rbound = 1 / mip_bound;
        // `0.5 * (x * mip_rbound + 1) * H` is a double product in the reference (raymarching.cu:375-377)
        const int nx = (int)clampf((float)(0.5 * (double)(x * mip_rbound + 1) * (double)H), 0.0f, (float)(H - 1));
        const int ny = (int)clampf((float)(0.5 * (double)(y * mip_rbound + 1) * (double)H), 0.0f, (float)(H - 1));
        const int nz = (int)clampf((float)(0.5 * (double)(z * mip_rbound + 1) * (double)H), 0.0f, (float)(H - 1));
        const uint32_t vox = (uint32_t)(level * H3 + (float)morton3D(nx, ny, nz));
        const bool occ = grid[vox / 8] & (1 << (vox % 8));
        if (occ) {
            if (WRITE) {
                xyzs[0] = x; xyzs[1] = y; xyzs[2] = z;
                dirs[0] = dx; dirs[1] = dy; dirs[2] = dz;
            }
            t += dt;
            if (WRITE) {
                deltas[0] = dt;
                deltas[1] = t - last_t;
                last_t = t;
                xyzs += 3; dirs += 3; deltas += 2;
            }
            step++;
        } else {
            const float tx = (((nx + 0.5f + 0.5f * signf(dx)) * rH * 2 - 1) * mip_bound - x) * rdx;
            const float ty = (((ny + 0.5f + 0.5f * signf(dy)) * rH * 2 - 1) * mip_bound - y) * rdy;
            const float tz = (((nz + 0.5f + 0.5f * signf(dz)) * rH * 2 - 1) * mip_bound - z) * rdz;
            const float tt = t + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
            do { t += clampf(t * dt_gamma, dt_min, dt_max); } while (t < tt);
        }
    }
    return step;
}

__device__ __forceinline__ float train_t0(float near, float noise, float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H) {
    const float dt_min = 2 * 1.73205080757f / max_steps;
    const float dt_max = 2 * 1.73205080757f * (1 << (C - 1)) / H;
    return near + clampf(near * dt_gamma, dt_min, dt_max) * noise;  // raymarching.cu:348-351
}

__global__ void __launch_bounds__(128) k_train_count(const float* __restrict__ rays_o, const float* __restrict__ rays_d, const uint8_t* __restrict__ grid,
                                                     float bound, float dt_gamma, uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H,
                                                     const float* __restrict__ nears, const float* __restrict__ fars, int* __restrict__ rays,
                                                     const float* __restrict__ noises) {
    const uint32_t n = threadIdx.x + blockIdx.x * blockDim.x;
    if (n >= N) return;
    const float t0 = train_t0(nears[n], noises ? noises[n] : 0.0f, dt_gamma, max_steps, C, H);
    const uint32_t cnt = train_march_pass<false>(rays_o + (size_t)n * 3, rays_d + (size_t)n * 3, t0, fars[n], max_steps, bound, dt_gamma, max_steps, C, H, grid,
                                                 nullptr, nullptr, nullptr);
    rays[(size_t)n * 3] = (int)n;
    rays[(size_t)n * 3 + 2] = (int)cnt;
}

// ---- wave-per-ray form of the two passes.  4 096 training rays are 64 waves in the lane-per-ray form: one ray's whole march (hundreds of
// dependent iterations) is the launch time, with 1/16 of the SIMDs busy.  Both branches of the reference loop advance t by the same recurrence
// s_{k+1} = s_k + clamp(s_k * dt_gamma, dt_min, dt_max) ("emit" takes one step, "hop to the voxel exit" takes steps until t >= tt), so the
// values a ray can visit are a fixed sequence and the evaluation at s_k is a pure function of s_k (the observation behind the hot path's
// pn_march_window.h).  Per round the 64 lanes evaluate s_0..s_63 of the current window — lane k replays k steps of the recurrence, so every value is
// rounded exactly as in the sequential loop — each lane walks its own hop to the index it would land on, and the wave replays the visit chain
// 0 -> jump[0] -> ... with uniform lane reads.  Samples, deltas and counts are bit-identical to the lane-per-ray kernels (and to the oracle).
struct TrainEval { float x, y, z, dt, t_next; int jump; bool occ, valid; };

__device__ __forceinline__ TrainEval train_eval(float ox, float oy, float oz, float dx, float dy, float dz, float rdx, float rdy, float rdz, float t,
                                                float far, int lane, float bound, float dt_gamma, float dt_min, float dt_max, uint32_t C, uint32_t H,
                                                const uint8_t* __restrict__ grid) {
    TrainEval e;
    e.valid = t < far;
    e.x = clampf(ox + t * dx, -bound, bound);
    e.y = clampf(oy + t * dy, -bound, bound);
    e.z = clampf(oz + t * dz, -bound, bound);
    e.dt = clampf(t * dt_gamma, dt_min, dt_max);
    e.occ = false;
    e.jump = lane + 1;
    e.t_next = t + e.dt;
    if (!e.valid) return e;
    const float rH = 1 / (float)H;
    const float H3 = (float)(H * H * H);
    const int level = max(mip_from_pos(e.x, e.y, e.z, (float)C), mip_from_dt(e.dt, (float)H, (float)C));
    const float mip_bound = fminf(scalbnf(1.0f, level), bound);
    const float mip_rbound = 1 / mip_bound;
    const int nx = (int)clampf((float)(0.5 * (double)(e.x * mip_rbound + 1) * (double)H), 0.0f, (float)(H - 1));
    const int ny = (int)clampf((float)(0.5 * (double)(e.y * mip_rbound + 1) * (double)H), 0.0f, (float)(H - 1));
    const int nz = (int)clampf((float)(0.5 * (double)(e.z * mip_rbound + 1) * (double)H), 0.0f, (float)(H - 1));
    const uint32_t vox = (uint32_t)(level * H3 + (float)morton3D(nx, ny, nz));
    e.occ = grid[vox / 8] & (1 << (vox % 8));
    if (!e.occ) {
        const float tx = (((nx + 0.5f + 0.5f * signf(dx)) * rH * 2 - 1) * mip_bound - e.x) * rdx;
        const float ty = (((ny + 0.5f + 0.5f * signf(dy)) * rH * 2 - 1) * mip_bound - e.y) * rdy;
        const float tz = (((nz + 0.5f + 0.5f * signf(dz)) * rH * 2 - 1) * mip_bound - e.z) * rdz;
        const float tt = t + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
        float u = t;
        int m = 0;
        do { u += clampf(u * dt_gamma, dt_min, dt_max); m++; } while (u < tt);
        e.jump = lane + m;  // may lie beyond the window; t_next is then the next window's first element
        e.t_next = u;
    }
    return e;
}

// One wave marches one ray from t0 until t >= far or `limit` occupied steps.  WRITE = false: returns the count.  WRITE = true: stores the samples.
template <bool WRITE>
__device__ __forceinline__ uint32_t train_march_wave(const float* __restrict__ ro, const float* __restrict__ rd, float t0, float far, uint32_t limit,
                                                     float bound, float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H,
                                                     const uint8_t* __restrict__ grid, float* __restrict__ xyzs, float* __restrict__ dirs,
                                                     float* __restrict__ deltas) {
    const int lane = threadIdx.x & 63;
    const float ox = ro[0], oy = ro[1], oz = ro[2];
    const float dx = rd[0], dy = rd[1], dz = rd[2];
    const float rdx = 1 / dx, rdy = 1 / dy, rdz = 1 / dz;
    const float dt_min = 2 * 1.73205080757f / max_steps;
    const float dt_max = 2 * 1.73205080757f * (1 << (C - 1)) / H;
    float t_win = t0, last_t = t0;  // wave-uniform
    uint32_t step = 0;
    while (t_win < far && step < limit) {
        float t = t_win;  // lane k: k steps of the recurrence from the window's first element
        for (int j = 0; j < lane; j++) t += clampf(t * dt_gamma, dt_min, dt_max);
        const TrainEval e = train_eval(ox, oy, oz, dx, dy, dz, rdx, rdy, rdz, t, far, lane, bound, dt_gamma, dt_min, dt_max, C, H, grid);
        // replay the visit chain (all operands wave-uniform)
        int pos = 0;
        bool done = false;
        bool mine = false;
        uint32_t my_slot = 0;
        float my_last = 0.0f;
        float t_exit = t_win;
        while (pos < 64) {
            const bool valid = __shfl((int)e.valid, pos) != 0;
            if (!valid || step >= limit) { done = true; break; }
            const bool occ = __shfl((int)e.occ, pos) != 0;
            const float tn = __shfl(e.t_next, pos);
            if (occ) {
                if (lane == pos) { mine = true; my_slot = step; my_last = last_t; }
                last_t = tn;
                step++;
            }
            t_exit = tn;
            pos = __shfl(e.jump, pos);
        }
        if (WRITE && mine) {
            float* px = xyzs + (size_t)my_slot * 3;
            float* pd = dirs + (size_t)my_slot * 3;
            float* pl = deltas + (size_t)my_slot * 2;
            px[0] = e.x; px[1] = e.y; px[2] = e.z;
            pd[0] = dx; pd[1] = dy; pd[2] = dz;
            pl[0] = e.dt;
            pl[1] = e.t_next - my_last;
        }
        if (done) break;
        t_win = t_exit;
    }
    return step;
}

__global__ void __launch_bounds__(256) k_train_count_w(const float* __restrict__ rays_o, const float* __restrict__ rays_d, const uint8_t* __restrict__ grid,
                                                       float bound, float dt_gamma, uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H,
                                                       const float* __restrict__ nears, const float* __restrict__ fars, int* __restrict__ rays,
                                                       const float* __restrict__ noises) {
    const uint32_t n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const float t0 = train_t0(nears[n], noises ? noises[n] : 0.0f, dt_gamma, max_steps, C, H);
    const uint32_t cnt = train_march_wave<false>(rays_o + (size_t)n * 3, rays_d + (size_t)n * 3, t0, fars[n], max_steps, bound, dt_gamma, max_steps, C, H, grid,
                                                 nullptr, nullptr, nullptr);
    if ((threadIdx.x & 63) == 0) {
        rays[(size_t)n * 3] = (int)n;
        rays[(size_t)n * 3 + 2] = (int)cnt;
    }
}

__global__ void __launch_bounds__(256) k_train_write_w(const float* __restrict__ rays_o, const float* __restrict__ rays_d, const uint8_t* __restrict__ grid,
                                                       float bound, float dt_gamma, uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, uint32_t M,
                                                       const float* __restrict__ nears, const float* __restrict__ fars, float* __restrict__ xyzs,
                                                       float* __restrict__ dirs, float* __restrict__ deltas, const int* __restrict__ rays,
                                                       const float* __restrict__ noises) {
    const uint32_t n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const uint32_t off = (uint32_t)rays[(size_t)n * 3 + 1], cnt = (uint32_t)rays[(size_t)n * 3 + 2];
    if (cnt == 0 || off + cnt > M) return;  // raymarching.cu:414-415
    const float t0 = train_t0(nears[n], noises ? noises[n] : 0.0f, dt_gamma, max_steps, C, H);
    train_march_wave<true>(rays_o + (size_t)n * 3, rays_d + (size_t)n * 3, t0, fars[n], cnt, bound, dt_gamma, max_steps, C, H, grid, xyzs + (size_t)off * 3,
                           dirs + (size_t)off * 3, deltas + (size_t)off * 2);
}

__global__ void __launch_bounds__(1024) k_train_scan(int* __restrict__ rays, uint32_t N, int* __restrict__ counter) {
    __shared__ uint32_t wave_tot[16];
    __shared__ uint32_t carry_s;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = (uint32_t)counter[0];
    __syncthreads();
    for (uint32_t base = 0; base < N; base += 1024) {
        const uint32_t n = base + threadIdx.x;
        const uint32_t v = n < N ? (uint32_t)rays[(size_t)n * 3 + 2] : 0u;
        uint32_t incl = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t up = __shfl_up(incl, off, 64);
            if (lane >= (uint32_t)off) incl += up;
        }
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        uint32_t before = carry_s;
        for (uint32_t w = 0; w < wave; w++) before += wave_tot[w];
        if (n < N) rays[(size_t)n * 3 + 1] = (int)(before + incl - v);
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = before + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        counter[0] = (int)carry_s;
        counter[1] += (int)N;
    }
}

__global__ void __launch_bounds__(128) k_train_write(const float* __restrict__ rays_o, const float* __restrict__ rays_d, const uint8_t* __restrict__ grid,
                                                     float bound, float dt_gamma, uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, uint32_t M,
                                                     const float* __restrict__ nears, const float* __restrict__ fars, float* __restrict__ xyzs,
                                                     float* __restrict__ dirs, float* __restrict__ deltas, const int* __restrict__ rays,
                                                     const float* __restrict__ noises) {
    const uint32_t n = threadIdx.x + blockIdx.x * blockDim.x;
    if (n >= N) return;
    const uint32_t off = (uint32_t)rays[(size_t)n * 3 + 1], cnt = (uint32_t)rays[(size_t)n * 3 + 2];
    if (cnt == 0 || off + cnt > M) return;  // raymarching.cu:414-415
    const float t0 = train_t0(nears[n], noises ? noises[n] : 0.0f, dt_gamma, max_steps, C, H);
    train_march_pass<true>(rays_o + (size_t)n * 3, rays_d + (size_t)n * 3, t0, fars[n], cnt, bound, dt_gamma, max_steps, C, H, grid, xyzs + (size_t)off * 3,
                           dirs + (size_t)off * 3, deltas + (size_t)off * 2);
}

// kernel_composite_rays_train_forward (raymarching.cu:503-581), one lane per ray row
__global__ void __launch_bounds__(128) k_composite_train_fwd(const float* __restrict__ sigmas, const float* __restrict__ rgbs, const float* __restrict__ deltas,
                                                             const int* __restrict__ rays, uint32_t M, uint32_t N, float T_thresh, float* weights_sum,
                                                             float* depth, float* image) {
    const uint32_t n = threadIdx.x + blockIdx.x * blockDim.x;
    if (n >= N) return;
    const uint32_t index = (uint32_t)rays[(size_t)n * 3], offset = (uint32_t)rays[(size_t)n * 3 + 1], num_steps = (uint32_t)rays[(size_t)n * 3 + 2];
    if (num_steps == 0 || offset + num_steps > M) {
        weights_sum[index] = 0;
        depth[index] = 0;
        image[(size_t)index * 3] = 0; image[(size_t)index * 3 + 1] = 0; image[(size_t)index * 3 + 2] = 0;
        return;
    }
    sigmas += offset; rgbs += (size_t)offset * 3; deltas += (size_t)offset * 2;
    uint32_t step = 0;
    float T = 1.0f, r = 0, g = 0, b = 0, ws = 0, t = 0, d = 0;
    while (step < num_steps) {
        const float alpha = 1.0f - __expf(-sigmas[0] * deltas[0]);
        const float weight = alpha * T;
        r += weight * rgbs[0]; g += weight * rgbs[1]; b += weight * rgbs[2];
        t += deltas[1];
        d += weight * t;
        ws += weight;
        T *= 1.0f - alpha;
        if (T < T_thresh) break;
        sigmas++; rgbs += 3; deltas += 2; step++;
    }
    weights_sum[index] = ws;
    depth[index] = d;
    image[(size_t)index * 3] = r; image[(size_t)index * 3 + 1] = g; image[(size_t)index * 3 + 2] = b;
}

// kernel_composite_rays_train_backward (raymarching.cu:604-686)
__global__ void __launch_bounds__(128) k_composite_train_bwd(const float* __restrict__ grad_weights_sum, const float* __restrict__ grad_image,
                                                             const float* __restrict__ sigmas, const float* __restrict__ rgbs,
                                                             const float* __restrict__ deltas, const int* __restrict__ rays,
                                                             const float* __restrict__ weights_sum, const float* __restrict__ image, uint32_t M, uint32_t N,
                                                             float T_thresh, float* __restrict__ grad_sigmas, float* __restrict__ grad_rgbs) {
    const uint32_t n = threadIdx.x + blockIdx.x * blockDim.x;
    if (n >= N) return;
    const uint32_t index = (uint32_t)rays[(size_t)n * 3], offset = (uint32_t)rays[(size_t)n * 3 + 1], num_steps = (uint32_t)rays[(size_t)n * 3 + 2];
    if (num_steps == 0 || offset + num_steps > M) return;
    const float gws = grad_weights_sum[index];
    const float gi0 = grad_image[(size_t)index * 3], gi1 = grad_image[(size_t)index * 3 + 1], gi2 = grad_image[(size_t)index * 3 + 2];
    const float r_final = image[(size_t)index * 3], g_final = image[(size_t)index * 3 + 1], b_final = image[(size_t)index * 3 + 2];
    const float ws_final = weights_sum[index];
    sigmas += offset; rgbs += (size_t)offset * 3; deltas += (size_t)offset * 2;
    grad_sigmas += offset; grad_rgbs += (size_t)offset * 3;
    uint32_t step = 0;
    float T = 1.0f, r = 0, g = 0, b = 0;
    while (step < num_steps) {
        const float alpha = 1.0f - __expf(-sigmas[0] * deltas[0]);
        const float weight = alpha * T;
        r += weight * rgbs[0]; g += weight * rgbs[1]; b += weight * rgbs[2];
        T *= 1.0f - alpha;
        grad_rgbs[0] = gi0 * weight; grad_rgbs[1] = gi1 * weight; grad_rgbs[2] = gi2 * weight;
        grad_sigmas[0] = deltas[0] * (gi0 * (T * rgbs[0] - (r_final - r)) + gi1 * (T * rgbs[1] - (g_final - g)) + gi2 * (T * rgbs[2] - (b_final - b)) +
                                      gws * (1 - ws_final));
        if (T < T_thresh) break;
        sigmas++; rgbs += 3; deltas += 2; grad_sigmas++; grad_rgbs += 3; step++;
    }
}

// ---- wave-per-ray composite (training batches: 4 096 rays are 64 waves in the lane-per-ray form, each walking ~17 samples one after the
// other with dependent loads).  A wave takes its ray's samples 64 at a time: alpha in parallel (coalesced loads, one __expf per lane),
// transmittance by an inclusive product scan, the early exit as a ballot over T < T_thresh, sums by shuffle trees.  Same formulas as the
// reference kernels; the order of the fp32 products / sums differs from the sequential loop (1e-7 relative), which is why these kernels are
// compared with the oracle to a tolerance like the rest of the composite.
__device__ __forceinline__ float wave_incl_prod(float v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const float u = __shfl_up(v, o); if (lane >= o) v *= u; }
    return v;
}
__device__ __forceinline__ float wave_incl_sum(float v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const float u = __shfl_up(v, o); if (lane >= o) v += u; }
    return v;
}
template <bool BWD>
__global__ void __launch_bounds__(256) k_composite_train_w(const float* __restrict__ grad_weights_sum, const float* __restrict__ grad_image,
                                                           const float* __restrict__ sigmas, const float* __restrict__ rgbs,
                                                           const float* __restrict__ deltas, const int* __restrict__ rays,
                                                           const float* __restrict__ ws_in, const float* __restrict__ image_in, uint32_t M, uint32_t N,
                                                           float T_thresh, float* __restrict__ weights_sum, float* __restrict__ depth,
                                                           float* __restrict__ image, float* __restrict__ grad_sigmas, float* __restrict__ grad_rgbs) {
    const uint32_t n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const int lane = threadIdx.x & 63;
    const uint32_t index = (uint32_t)rays[(size_t)n * 3], offset = (uint32_t)rays[(size_t)n * 3 + 1], num_steps = (uint32_t)rays[(size_t)n * 3 + 2];
    if (num_steps == 0 || offset + num_steps > M) {
        if (!BWD && lane == 0) {
            weights_sum[index] = 0; depth[index] = 0;
            image[(size_t)index * 3] = 0; image[(size_t)index * 3 + 1] = 0; image[(size_t)index * 3 + 2] = 0;
        }
        return;
    }
    float gws = 0, gi0 = 0, gi1 = 0, gi2 = 0, rf = 0, gf = 0, bf = 0, wsf = 0;
    if (BWD) {
        gws = grad_weights_sum[index];
        gi0 = grad_image[(size_t)index * 3]; gi1 = grad_image[(size_t)index * 3 + 1]; gi2 = grad_image[(size_t)index * 3 + 2];
        rf = image_in[(size_t)index * 3]; gf = image_in[(size_t)index * 3 + 1]; bf = image_in[(size_t)index * 3 + 2];
        wsf = ws_in[index];
    }
    float T0 = 1.0f, r = 0, g = 0, b = 0, ws = 0, t0 = 0, d = 0;  // carried across windows (wave-uniform)
    for (uint32_t base = 0; base < num_steps; base += 64) {
        const uint32_t s = base + lane;
        const bool on = s < num_steps;
        const size_t m = (size_t)offset + (on ? s : 0);
        const float sg = on ? sigmas[m] : 0.0f, d0 = on ? deltas[m * 2] : 0.0f, d1 = on ? deltas[m * 2 + 1] : 0.0f;
        const float c0 = on ? rgbs[m * 3] : 0.0f, c1 = on ? rgbs[m * 3 + 1] : 0.0f, c2 = on ? rgbs[m * 3 + 2] : 0.0f;
        const float alpha = on ? 1.0f - __expf(-sg * d0) : 0.0f;
        const float Tafter = T0 * wave_incl_prod(1.0f - alpha, lane);  // transmittance after this sample
        float Tbefore = __shfl_up(Tafter, 1);
        if (lane == 0) Tbefore = T0;
        // the sample at which T drops below the threshold is still accumulated, later ones are not (raymarching.cu:559-560)
        const unsigned long long below = __ballot(on && Tafter < T_thresh);
        const int last = below ? __ffsll((long long)below) - 1 : 63;
        const bool use = on && lane <= last;
        const float w = use ? alpha * Tbefore : 0.0f;
        const float tcum = t0 + wave_incl_sum(on ? d1 : 0.0f, lane);
        const float pr = r + wave_incl_sum(w * c0, lane), pg = g + wave_incl_sum(w * c1, lane), pb = b + wave_incl_sum(w * c2, lane);
        if (BWD) {
            if (use) {
                grad_rgbs[m * 3] = gi0 * w; grad_rgbs[m * 3 + 1] = gi1 * w; grad_rgbs[m * 3 + 2] = gi2 * w;
                grad_sigmas[m] = d0 * (gi0 * (Tafter * c0 - (rf - pr)) + gi1 * (Tafter * c1 - (gf - pg)) + gi2 * (Tafter * c2 - (bf - pb)) + gws * (1 - wsf));
            }
        } else {
            ws += __shfl(wave_incl_sum(w, lane), 63);
            d += __shfl(wave_incl_sum(w * tcum, lane), 63);
        }
        r = __shfl(pr, 63); g = __shfl(pg, 63); b = __shfl(pb, 63);
        t0 = __shfl(tcum, 63);
        T0 = __shfl(Tafter, 63);
        if (below) break;
    }
    if (!BWD && lane == 0) {
        weights_sum[index] = ws; depth[index] = d;
        image[(size_t)index * 3] = r; image[(size_t)index * 3 + 1] = g; image[(size_t)index * 3 + 2] = b;
    }
}

}  // namespace

extern "C" int pn_march_rays_train(const float* rays_o, const float* rays_d, const uint8_t* grid, float bound, float dt_gamma, uint32_t max_steps,
                                   uint32_t N, uint32_t C, uint32_t H, uint32_t M, const float* nears, const float* fars, float* xyzs, float* dirs,
                                   float* deltas, int* rays, int* counter, const float* noises, void* stream) {
    if (N == 0) return PN_OK;
    PN_REQUIRE(rays_o && rays_d && grid && nears && fars && rays && counter && (M == 0 || (xyzs && dirs && deltas)));
    PN_REQUIRE(C >= 1 && C <= 8 && H > 0 && max_steps > 0);
    hipStream_t st = (hipStream_t)stream;
    // wave per ray while that still fits the chip in a few rounds (training batches); lane per ray for whole images (PN_TRAIN_MARCH=lane|wave forces one)
    static const char* form = getenv("PN_TRAIN_MARCH");
    const bool wave = form ? strcmp(form, "wave") == 0 : N <= 131072;
    if (wave) {
        k_train_count_w<<<pn_div_up(N, 4), 256, 0, st>>>(rays_o, rays_d, grid, bound, dt_gamma, max_steps, N, C, H, nears, fars, rays, noises);
        k_train_scan<<<1, 1024, 0, st>>>(rays, N, counter);
        k_train_write_w<<<pn_div_up(N, 4), 256, 0, st>>>(rays_o, rays_d, grid, bound, dt_gamma, max_steps, N, C, H, M, nears, fars, xyzs, dirs, deltas, rays,
                                                         noises);
    } else {
        k_train_count<<<pn_div_up(N, 128), 128, 0, st>>>(rays_o, rays_d, grid, bound, dt_gamma, max_steps, N, C, H, nears, fars, rays, noises);
        k_train_scan<<<1, 1024, 0, st>>>(rays, N, counter);
        k_train_write<<<pn_div_up(N, 128), 128, 0, st>>>(rays_o, rays_d, grid, bound, dt_gamma, max_steps, N, C, H, M, nears, fars, xyzs, dirs, deltas, rays,
                                                         noises);
    }
    PN_LAUNCH_CHECK();
    return PN_OK;
}

extern "C" int pn_composite_rays_train_forward(const float* sigmas, const float* rgbs, const float* deltas, const int* rays, uint32_t M, uint32_t N,
                                               float T_thresh, float* weights_sum, float* depth, float* image, void* stream) {
    if (N == 0) return PN_OK;
    PN_REQUIRE(rays && weights_sum && depth && image && (M == 0 || (sigmas && rgbs && deltas)));
    static const char* form = getenv("PN_TRAIN_COMPOSITE");  // lane | wave; default: wave per ray for training batches, lane per ray for whole images
    if (form ? strcmp(form, "wave") == 0 : N <= 131072)
        k_composite_train_w<false><<<pn_div_up(N, 4), 256, 0, (hipStream_t)stream>>>(nullptr, nullptr, sigmas, rgbs, deltas, rays, nullptr, nullptr, M, N, T_thresh,
                                                                                    weights_sum, depth, image, nullptr, nullptr);
    else
        k_composite_train_fwd<<<pn_div_up(N, 128), 128, 0, (hipStream_t)stream>>>(sigmas, rgbs, deltas, rays, M, N, T_thresh, weights_sum, depth, image);
    PN_LAUNCH_CHECK();
    return PN_OK;
}

extern "C" int pn_composite_rays_train_backward(const float* grad_weights_sum, const float* grad_image, const float* sigmas, const float* rgbs,
                                                const float* deltas, const int* rays, const float* weights_sum, const float* image, uint32_t M,
                                                uint32_t N, float T_thresh, float* grad_sigmas, float* grad_rgbs, void* stream) {
    if (N == 0 || M == 0) return PN_OK;
    PN_REQUIRE(grad_weights_sum && grad_image && sigmas && rgbs && deltas && rays && weights_sum && image && grad_sigmas && grad_rgbs);
    static const char* form = getenv("PN_TRAIN_COMPOSITE");
    if (form ? strcmp(form, "wave") == 0 : N <= 131072)
        k_composite_train_w<true><<<pn_div_up(N, 4), 256, 0, (hipStream_t)stream>>>(grad_weights_sum, grad_image, sigmas, rgbs, deltas, rays, weights_sum, image, M, N,
                                                                                   T_thresh, nullptr, nullptr, nullptr, grad_sigmas, grad_rgbs);
    else
        k_composite_train_bwd<<<pn_div_up(N, 128), 128, 0, (hipStream_t)stream>>>(grad_weights_sum, grad_image, sigmas, rgbs, deltas, rays, weights_sum, image,
                                                                                M, N, T_thresh, grad_sigmas, grad_rgbs);
    PN_LAUNCH_CHECK();
    return PN_OK;
}
