// Density-grid state of the cuda_ray renderer for gfx950 (off the simulate-and-render hot path):
// NeRFRenderer.mark_untrained_grid (nerf/renderer.py:390-452) and NeRFRenderer.update_extra_state (:454-549) — paths relative to
// /root/reference.  The reference runs both as Python loops (five levels deep for the former: 8 blocks of 64^3 cells x cascades x
// pose batches of torch ops and a batched matmul each); here every cell is one lane:
//   k_mark_untrained      lane per (cascade, morton cell): morton decode, cell centre, frustum test against every pose (poses are read with
//                         uniform scalar loads), writes -1 where no camera sees the cell, counts those cells with one atomic per wave
//   k_cells_full          lane per (cascade, morton cell): jittered cell centre -> the sample the density query evaluates (full sweep)
//   k_occ_candidates      lane per cell: tmp = -1, candidate list of the cascade's occupied cells (compacted by pn_compact_rays)
//   k_cells_partial       lane per sample: N uniform cells + N picks from the occupied list (partial sweep)
//   k_scatter_sigma       tmp[index] = sigma
//   k_grid_ema / _mean    EMA-max with the old grid + deterministic mean of clamp(grid, 0) (per-block double partial sums, summed in order)
//   k_packbits_dev        packbits with the threshold min(mean, density_thresh) read from device memory (no host round trip in between)
// Compiled with -ffp-contract=off: every float operation of the cell-centre / frustum arithmetic rounds once, in the reference's order
// of torch ops, so the set of unseen cells equals the CPU oracle's bit for bit.
#include <float.h>

#include "pn_march_math.h"

namespace {
using namespace pnm;

__device__ __forceinline__ uint32_t compact_bits(uint32_t x) {  // raymarching.cu:73-81 (morton3D_invert)
    x = x & 0x49249249u;
    x = (x | (x >> 2)) & 0xc30c30c3u;
    x = (x | (x >> 4)) & 0x0f00f00fu;
    x = (x | (x >> 8)) & 0xff0000ffu;
    x = (x | (x >> 16)) & 0x0000ffffu;
    return x;
}

// cell `c` of axis length H -> cascade-space coordinate: (2 c / (H - 1) - 1) * (bound_c - half)   (renderer.py:420,428 / :481,489)
__device__ __forceinline__ float cell_centre(uint32_t c, float Hm1, float span) { return (2.0f * (float)c / Hm1 - 1.0f) * span; }

__global__ void __launch_bounds__(256) k_mark_untrained(const float* __restrict__ poses, uint32_t B, float cxfx, float cyfy, uint32_t cascade, uint32_t H,
                                                        float bound, float* __restrict__ grid, int* __restrict__ n_unseen) {
    const uint32_t H3 = H * H * H;
    const uint32_t i = threadIdx.x + blockIdx.x * blockDim.x;
    const bool live = i < cascade * H3;
    bool unseen = false;
    if (live) {
        const uint32_t cas = i / H3, m = i - cas * H3;
        const float bnd = fminf((float)(1u << cas), bound);            // min(2 ** cas, self.bound)
        const float half = bnd / (float)H;                             // half_grid_size
        const float span = bnd - half, pad = half * 2.0f, Hm1 = (float)(H - 1);
        const float wx = cell_centre(compact_bits(m), Hm1, span), wy = cell_centre(compact_bits(m >> 1), Hm1, span),
                    wz = cell_centre(compact_bits(m >> 2), Hm1, span);
        uint32_t count = 0;
        for (uint32_t b = 0; b < B; b++) {
            const float* __restrict__ P = poses + (size_t)b * 16;      // row-major cam2world; uniform address -> scalar loads
            const float dx = wx - P[3], dy = wy - P[7], dz = wz - P[11];
            // cam = (world - t) @ R  (renderer.py:437-438): column j of R
            const float cx_ = dx * P[0] + dy * P[4] + dz * P[8];
            const float cy_ = dx * P[1] + dy * P[5] + dz * P[9];
            const float cz_ = dx * P[2] + dy * P[6] + dz * P[10];
            const bool in = (cz_ > 0) && (fabsf(cx_) < cxfx * cz_ + pad) && (fabsf(cy_) < cyfy * cz_ + pad);  // :441-444
            count += in ? 1u : 0u;
        }
        unseen = count == 0;
        if (unseen) grid[i] = -1.0f;                                   // self.density_grid[count == 0] = -1 (:449)
    }
    const unsigned long long mask = __ballot(unseen);
    if ((threadIdx.x & 63) == 0 && mask) atomicAdd(n_unseen, (int)__popcll(mask));
}

__global__ void __launch_bounds__(256) k_cells_full(uint32_t cascade, uint32_t H, float bound, const float* __restrict__ noise, float* __restrict__ xyzs) {
    const uint32_t H3 = H * H * H;
    const uint32_t i = threadIdx.x + blockIdx.x * blockDim.x;
    if (i >= cascade * H3) return;
    const uint32_t cas = i / H3, m = i - cas * H3;
    const float bnd = fminf((float)(1u << cas), bound), half = bnd / (float)H, span = bnd - half, Hm1 = (float)(H - 1);
    const uint32_t c[3] = {compact_bits(m), compact_bits(m >> 1), compact_bits(m >> 2)};
#pragma unroll
    for (int d = 0; d < 3; d++)  // cas_xyzs = xyzs * (bound - hgs); cas_xyzs += (rand * 2 - 1) * hgs   (:489-491)
        xyzs[(size_t)i * 3 + d] = cell_centre(c[d], Hm1, span) + (noise[(size_t)i * 3 + d] * 2.0f - 1.0f) * half;
}

__global__ void __launch_bounds__(256) k_occ_candidates(uint32_t H3, const float* __restrict__ grid_cas, int* __restrict__ cand, float* __restrict__ tmp_cas) {
    const uint32_t i = threadIdx.x + blockIdx.x * blockDim.x;
    if (i >= H3) return;
    cand[i] = grid_cas[i] > 0 ? (int)i : -1;  // torch.nonzero(self.density_grid[cas] > 0) (:507)
    tmp_cas[i] = -1.0f;                        // tmp_grid = -ones_like(density_grid) (:462)
}

__global__ void __launch_bounds__(256) k_cells_partial(uint32_t cas, uint32_t H, float bound, uint32_t N, const int* __restrict__ rand_coords,
                                                       const float* __restrict__ rand_pick, const float* __restrict__ noise, const int* __restrict__ occ,
                                                       const int* __restrict__ n_occ_dev, int* __restrict__ indices, float* __restrict__ xyzs) {
    const uint32_t m = threadIdx.x + blockIdx.x * blockDim.x;
    if (m >= 2 * N) return;
    const float bnd = fminf((float)(1u << cas), bound), half = bnd / (float)H, span = bnd - half, Hm1 = (float)(H - 1);
    uint32_t c[3];
    int index;
    if (m < N) {  // coords = torch.randint(0, H, (N, 3)); indices = morton3D(coords)   (:503-504)
        c[0] = (uint32_t)rand_coords[m * 3]; c[1] = (uint32_t)rand_coords[m * 3 + 1]; c[2] = (uint32_t)rand_coords[m * 3 + 2];
        index = (int)morton3D(c[0], c[1], c[2]);
    } else {      // occ_indices[randint(0, n_occ)] (:508-509): floor(u * n_occ) with u uniform in [0, 1)
        const int n_occ = *n_occ_dev;
        if (n_occ <= 0) {  // the reference raises here (randint over an empty range); nothing to add
            indices[m] = -1;
            xyzs[(size_t)m * 3] = xyzs[(size_t)m * 3 + 1] = xyzs[(size_t)m * 3 + 2] = 0.0f;
            return;
        }
        int j = (int)(rand_pick[m - N] * (float)n_occ);
        j = j < 0 ? 0 : (j >= n_occ ? n_occ - 1 : j);
        index = occ[j];
        c[0] = compact_bits((uint32_t)index); c[1] = compact_bits((uint32_t)index >> 1); c[2] = compact_bits((uint32_t)index >> 2);  // morton3D_invert (:510)
    }
    indices[m] = index;
#pragma unroll
    for (int d = 0; d < 3; d++) xyzs[(size_t)m * 3 + d] = cell_centre(c[d], Hm1, span) + (noise[(size_t)m * 3 + d] * 2.0f - 1.0f) * half;
}

// tmp_grid[cas, indices] = sigmas (:527): duplicates (a cell drawn twice) keep whichever store lands last, as in the reference
__global__ void __launch_bounds__(256) k_scatter_sigma(uint32_t n, const int* __restrict__ indices, const float* __restrict__ sigmas, float* __restrict__ tmp_cas) {
    const uint32_t m = threadIdx.x + blockIdx.x * blockDim.x;
    if (m >= n) return;
    const int idx = indices[m];
    if (idx >= 0) tmp_cas[idx] = sigmas[m];
}

// valid = (grid >= 0) & (tmp >= 0); grid[valid] = max(grid * decay, tmp)   (:535-536) + this block's sum of clamp(grid, 0)
__global__ void __launch_bounds__(256) k_grid_ema(uint32_t n, float* __restrict__ grid, const float* __restrict__ tmp, float decay, double* __restrict__ partial) {
    __shared__ double wsum[4];
    const uint32_t i = threadIdx.x + blockIdx.x * blockDim.x;
    double v = 0.0;
    if (i < n) {
        float g = grid[i];
        const float t = tmp[i];
        if (g >= 0 && t >= 0) { g = fmaxf(g * decay, t); grid[i] = g; }
        v = (double)fmaxf(g, 0.0f);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        int2 t2 = *reinterpret_cast<int2*>(&v);
        t2.x = __shfl_xor(t2.x, o); t2.y = __shfl_xor(t2.y, o);
        v += *reinterpret_cast<double*>(&t2);
    }
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// mean_density = mean(clamp(grid, 0)) (:537); density_thresh = min(mean_density, self.density_thresh) (:542).  One workgroup, fixed order.
__global__ void __launch_bounds__(1024) k_grid_mean(uint32_t n_partial, const double* __restrict__ partial, uint32_t n, float density_thresh,
                                                    float* __restrict__ out /*[2]: mean, threshold*/) {
    __shared__ double acc[1024];
    double s = 0.0;
    for (uint32_t k = threadIdx.x; k < n_partial; k += 1024) s += partial[k];
    acc[threadIdx.x] = s;
    __syncthreads();
    for (int w = 512; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) acc[threadIdx.x] += acc[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float mean = (float)(acc[0] / (double)n);
        out[0] = mean;
        out[1] = fminf(mean, density_thresh);
    }
}

// kernel_packbits (raymarching.cu:270-292) with the threshold in device memory
__global__ void __launch_bounds__(256) k_packbits_dev(const float* __restrict__ grid, uint32_t N, const float* __restrict__ thresh_dev, uint8_t* __restrict__ bitfield) {
    const uint32_t n = threadIdx.x + blockIdx.x * blockDim.x;
    if (n >= N) return;
    const float th = *thresh_dev;
    const float4 a = reinterpret_cast<const float4*>(grid)[2 * (size_t)n], b = reinterpret_cast<const float4*>(grid)[2 * (size_t)n + 1];
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    uint32_t bits = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) bits |= (v[i] > th) ? (1u << i) : 0u;
    bitfield[n] = (uint8_t)bits;
}

}  // namespace

extern "C" int pn_mark_untrained_grid(const float* poses, uint32_t B, float fx, float fy, float cx, float cy, uint32_t cascade, uint32_t H, float bound,
                                      float* density_grid, int* n_unseen, void* stream) {
    PN_REQUIRE(poses && density_grid && n_unseen && B > 0 && cascade >= 1 && cascade <= 8 && H > 1 && H <= 1024);
    hipStream_t st = (hipStream_t)stream;
    PN_HIP_CHECK(hipMemsetAsync(n_unseen, 0, sizeof(int), st));
    const uint32_t n = cascade * H * H * H;
    // `cx / fx` and `cy / fy` are Python (double) quotients that become float32 scalars in the tensor expression (renderer.py:442-443)
    k_mark_untrained<<<pn_div_up(n, 256), 256, 0, st>>>(poses, B, (float)((double)cx / (double)fx), (float)((double)cy / (double)fy), cascade, H, bound,
                                                        density_grid, n_unseen);
    PN_LAUNCH_CHECK();
    return PN_OK;
}

extern "C" int pn_density_cells_full(uint32_t cascade, uint32_t H, float bound, const float* noise, float* xyzs, void* stream) {
    PN_REQUIRE(noise && xyzs && cascade >= 1 && cascade <= 8 && H > 1 && H <= 1024);
    const uint32_t n = cascade * H * H * H;
    k_cells_full<<<pn_div_up(n, 256), 256, 0, (hipStream_t)stream>>>(cascade, H, bound, noise, xyzs);
    PN_LAUNCH_CHECK();
    return PN_OK;
}

extern "C" int pn_density_cells_partial(uint32_t cas, uint32_t H, float bound, uint32_t N, const int* rand_coords, const float* rand_pick, const float* noise,
                                        const float* density_grid_cas, float* tmp_grid_cas, int* scratch, int* indices, float* xyzs, void* stream) {
    PN_REQUIRE(rand_coords && rand_pick && noise && density_grid_cas && tmp_grid_cas && scratch && indices && xyzs && N > 0 && H > 1 && H <= 1024 && cas < 8);
    hipStream_t st = (hipStream_t)stream;
    const uint32_t H3 = H * H * H;
    // scratch: candidates [H3] | occupied list [H3] | count [1] | compaction scratch
    int* cand = scratch;
    int* occ = scratch + H3;
    int* n_occ = occ + H3;
    int* cscratch = n_occ + 1;
    k_occ_candidates<<<pn_div_up(H3, 256), 256, 0, st>>>(H3, density_grid_cas, cand, tmp_grid_cas);
    PN_LAUNCH_CHECK();
    const int rc = pn_compact_rays(cand, H3, occ, n_occ, cscratch, stream);
    if (rc) return rc;
    k_cells_partial<<<pn_div_up(2 * N, 256), 256, 0, st>>>(cas, H, bound, N, rand_coords, rand_pick, noise, occ, n_occ, indices, xyzs);
    PN_LAUNCH_CHECK();
    return PN_OK;
}

extern "C" uint64_t pn_density_partial_scratch_ints(uint32_t H) {
    const uint64_t H3 = (uint64_t)H * H * H;
    return 2 * H3 + 1 + pn_compact_scratch_ints((uint32_t)H3);
}

extern "C" int pn_density_scatter(uint32_t n, const int* indices, const float* sigmas, float* tmp_grid_cas, void* stream) {
    if (n == 0) return PN_OK;
    PN_REQUIRE(indices && sigmas && tmp_grid_cas);
    k_scatter_sigma<<<pn_div_up(n, 256), 256, 0, (hipStream_t)stream>>>(n, indices, sigmas, tmp_grid_cas);
    PN_LAUNCH_CHECK();
    return PN_OK;
}

extern "C" int pn_density_grid_update(uint32_t n, float* density_grid, const float* tmp_grid, float decay, float density_thresh, uint8_t* bitfield,
                                      double* partial, float* mean_thresh, void* stream) {
    PN_REQUIRE(density_grid && tmp_grid && bitfield && partial && mean_thresh && n > 0 && n % 8 == 0 && ((uintptr_t)density_grid & 15) == 0);
    hipStream_t st = (hipStream_t)stream;
    const uint32_t blocks = pn_div_up(n, 256);
    k_grid_ema<<<blocks, 256, 0, st>>>(n, density_grid, tmp_grid, decay, partial);
    k_grid_mean<<<1, 1024, 0, st>>>(blocks, partial, n, density_thresh, mean_thresh);
    k_packbits_dev<<<pn_div_up(n / 8, 256), 256, 0, st>>>(density_grid, n / 8, mean_thresh + 1, bitfield);
    PN_LAUNCH_CHECK();
    return PN_OK;
}
