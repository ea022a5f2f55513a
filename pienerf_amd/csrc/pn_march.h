// Device-side ray marching with the inverse-GMLS ("quadratic bending") warp, for gfx950.
//
// Semantics follow kernel_march_rays_quadratic_bending and its helpers
// (/root/reference raymarching/src/raymarching.cu:930-1434) including the output-changing quirks listed in
// SURVEY.md §8a R7q.  This translation unit is compiled with -ffp-contract=off: every float operation rounds once, in
// the order written, so the integer decisions (search cell, accepted-IP count, mip level, voxel index) match the CPU
// oracle bit for bit.  Mixed-precision subexpressions of the reference are kept (double `0.5 * ... * H`, `bbmax - 1e-6`,
// the double Newton residual).
#pragma once
#include "pn_common.h"

namespace pnm {

__device__ __forceinline__ float clampf(float x, float lo, float hi) { return fminf(hi, fmaxf(lo, x)); }
__device__ __forceinline__ float signf(float x) { return copysignf(1.0f, x); }

__device__ __forceinline__ int mip_from_pos(float x, float y, float z, float max_cascade) {
    const float mx = fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z)));
    int e;
    frexpf(mx, &e);
    return (int)fminf(max_cascade - 1, fmaxf(0.0f, (float)e));
}
__device__ __forceinline__ int mip_from_dt(float dt, float H, float max_cascade) {
    const float mx = (float)((double)(dt * H) * 0.5);
    int e;
    frexpf(mx, &e);
    return (int)fminf(max_cascade - 1, fmaxf(0.0f, (float)e));
}
__device__ __forceinline__ uint32_t expand_bits(uint32_t v) {
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
__device__ __forceinline__ uint32_t morton3D(uint32_t x, uint32_t y, uint32_t z) {
    return expand_bits(x) | (expand_bits(y) << 1) | (expand_bits(z) << 2);
}

// 26 neighbour offsets in the reference's visiting order (raymarching.cu:1006-1016), packed 2 bits per component (+1).
__device__ __constant__ const signed char NBR26[26][3] = {
    {-1, 0, 0}, {0, -1, 0}, {0, 0, -1}, {1, 0, 0},  {0, 1, 0},  {0, 0, 1},  {-1, -1, 0}, {-1, 0, -1}, {0, -1, -1},
    {1, 1, 0},  {1, 0, 1},  {0, 1, 1},  {-1, 1, 0}, {-1, 0, 1}, {0, -1, 1}, {1, -1, 0},  {1, 0, -1},  {0, 1, -1},
    {-1, -1, 1}, {-1, 1, -1}, {1, -1, -1}, {1, 1, -1}, {1, -1, 1}, {-1, 1, 1}, {-1, -1, -1}, {1, 1, 1}};

struct Search {
    const int *cnt, *bgn, *idx;
    const float* p_def;
    int r0, r1, r2, n_grid;
};

__device__ __forceinline__ float dist2(const float* __restrict__ q, float x, float y, float z) {
    const float a = q[0] - x, b = q[1] - y, c = q[2] - z;
    return a * a + b * b + c * c;
}

// find_closest_IP, raymarching.cu:986-1043 — own cell, neighbours only when the own cell is empty; offsets applied as (g2+f, g1+g, g0+h).
__device__ inline int find_closest_IP(const Search& s, float x, float y, float z, int g0, int g1, int g2) {
    int gid = g2 * s.r1 * s.r0 + g1 * s.r0 + g0;
    float best = (float)9999.9;
    int ip = -1;
    const int c0 = s.cnt[gid], b0 = s.bgn[gid];
    for (int i = 0; i < c0; i++) {
        const int t = s.idx[b0 + i];
        const float d = dist2(s.p_def + t * 3, x, y, z);
        if (d < best) { best = d; ip = t; }
    }
    if (ip == -1) {
        for (int k = 0; k < 26; k++) {
            const int f = NBR26[k][0], g = NBR26[k][1], h = NBR26[k][2];
            if (g2 + f >= s.r2 || g2 + f < 0 || g1 + g >= s.r1 || g1 + g < 0 || g0 + h >= s.r0 || g0 + h < 0) continue;
            gid = (g2 + f) * s.r1 * s.r0 + (g1 + g) * s.r0 + g0 + h;
            const int c = s.cnt[gid], b = s.bgn[gid];
            for (int i = 0; i < c; i++) {
                const int t = s.idx[b + i];
                const float d = dist2(s.p_def + t * 3, x, y, z);
                if (d < best) { best = d; ip = t; }
            }
        }
    }
    return ip;
}

// find_closest_IPs, raymarching.cu:1045-1118 — all 27 cells, K <= 3 kept in registers, strict '<' insertion.
template <int K>
__device__ inline int find_closest_IPs(const Search& s, float x, float y, float z, int g0, int g1, int g2, int* ips) {
    float d0 = FLT_MAX, d1 = FLT_MAX, d2 = FLT_MAX;
    int i0 = -1, i1 = -1, i2 = -1;
    auto visit = [&](int gid) {
        if (gid < 0 || gid >= s.n_grid) return;
        const int c = s.cnt[gid], b = s.bgn[gid];
        for (int i = 0; i < c; i++) {
            const int t = s.idx[b + i];
            const float d = dist2(s.p_def + t * 3, x, y, z);
            if (d < d0) { d2 = d1; i2 = i1; d1 = d0; i1 = i0; d0 = d; i0 = t; }
            else if (K > 1 && d < d1) { d2 = d1; i2 = i1; d1 = d; i1 = t; }
            else if (K > 2 && d < d2) { d2 = d; i2 = t; }
        }
    };
    visit(g2 * s.r1 * s.r0 + g1 * s.r0 + g0);
    for (int k = 0; k < 26; k++) {
        const int n0 = g0 + NBR26[k][0], n1 = g1 + NBR26[k][1], n2 = g2 + NBR26[k][2];
        if (n0 >= 0 && n0 < s.r0 && n1 >= 0 && n1 < s.r1 && n2 >= 0 && n2 < s.r2) visit(n2 * s.r1 * s.r0 + n1 * s.r0 + n0);
    }
    ips[0] = i0;
    if (K > 1) ips[1] = i1;
    if (K > 2) ips[2] = i2;
    return (i0 != -1) + (K > 1 && i1 != -1) + (K > 2 && i2 != -1);
}

// raymarching.cu:940-984 literal flat-index helpers
__device__ __forceinline__ void dot31(const float* __restrict__ T, const float* V, float* M) {
#pragma unroll
    for (int m = 0; m < 9; m++) M[m] = T[m] * V[0] + T[9 + m] * V[1] + T[18 + m] * V[2];
}
__device__ __forceinline__ void mul31(const float* M, const float* V, float* R) {
    R[0] = M[0] * V[0] + M[3] * V[1] + M[6] * V[2];
    R[1] = M[1] * V[0] + M[4] * V[1] + M[7] * V[2];
    R[2] = M[2] * V[0] + M[5] * V[1] + M[8] * V[2];
}
__device__ __forceinline__ void inv3x3(const float* A, float* Ai) {
    const float det = A[0] * (A[4] * A[8] - A[5] * A[7]) - A[1] * (A[3] * A[8] - A[5] * A[6]) + A[2] * (A[3] * A[7] - A[4] * A[6]);
    if (det == 0) return;  // Ai stays 0 (the reference never acts on the failure code, :1285-1287)
    const float id = 1.0f / det;
    Ai[0] = id * (A[4] * A[8] - A[5] * A[7]);
    Ai[1] = id * (A[2] * A[7] - A[1] * A[8]);
    Ai[2] = id * (A[1] * A[5] - A[2] * A[4]);
    Ai[3] = id * (A[5] * A[6] - A[3] * A[8]);
    Ai[4] = id * (A[0] * A[8] - A[2] * A[6]);
    Ai[5] = id * (A[2] * A[3] - A[0] * A[5]);
    Ai[6] = id * (A[3] * A[7] - A[4] * A[6]);
    Ai[7] = id * (A[1] * A[6] - A[0] * A[7]);
    Ai[8] = id * (A[0] * A[4] - A[1] * A[3]);
}

struct MarchParams {
    const int *pig_cnt, *pig_bgn, *pig_idx;
    int n_vtx, n_grid;
    const float *p_ori, *p_def, *F_IP, *dF_IP;
    int max_iter_num;
    const float *bbmin, *bbmax;
    float hgs;
    const int* resolution;
    int num_seek_IP;
    float IP_dx;
    int cut;
    const float* cut_bounds;
    const float *rays_t, *rays_o, *rays_d;
    float bound, dt_gamma;
    uint32_t max_steps, C, H;
    const uint8_t* grid;
    const float* fars;
    int* err_flag;
    unsigned long long* stats;  // optional [4]: marching iterations, candidates scanned, IP warps, samples emitted (bench instrumentation)
};

// Inverse warp of the deformed-space point (x,y,z) through IP k (raymarching.cu:1262-1324): Newton on
// phi(p) = F q + 1/2 (dF.q) q - q' with q = p - p_ori_k, q' = x - p_def_k.  Returns true when |p - p_ori_k|_inf > IP_dx.
__device__ inline bool warp_through_IP(const MarchParams& a, int ip, float x, float y, float z, float* p_out) {
    const float* __restrict__ pk = a.p_ori + ip * 3;
    const float* __restrict__ pk_ = a.p_def + ip * 3;
    const float* __restrict__ Fg = a.F_IP + ip * 9;
    const float* __restrict__ dFk = a.dF_IP + ip * 27;
    float Fk[9];
#pragma unroll
    for (int j = 0; j < 9; j++) Fk[j] = Fg[j];
    const float pk0 = pk[0], pk1 = pk[1], pk2 = pk[2];
    float p[3] = {pk0, pk1, pk2};
    const float q_[3] = {x - pk_[0], y - pk_[1], z - pk_[2]};
    int num_itr = 0;
    while (num_itr < a.max_iter_num) {
        const float q[3] = {p[0] - pk0, p[1] - pk1, p[2] - pk2};
        float dFk_q[9], A[9], A_inv[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        dot31(dFk, q, dFk_q);
#pragma unroll
        for (int j = 0; j < 9; j++) A[j] = Fk[j] + dFk_q[j];
        inv3x3(A, A_inv);
        float Fk_q[3], dFk_q_q[3], b[3], dq[3];
        mul31(Fk, q, Fk_q);
        mul31(dFk_q, q, dFk_q_q);
#pragma unroll
        for (int i = 0; i < 3; i++) b[i] = (float)(((double)Fk_q[i] + 0.5 * (double)dFk_q_q[i]) - (double)q_[i]);
        mul31(A_inv, b, dq);
        p[0] -= dq[0];
        p[1] -= dq[1];
        p[2] -= dq[2];
        if ((double)(dq[0] * dq[0] + dq[1] * dq[1] + dq[2] * dq[2]) < 1e-12) break;
        num_itr++;
    }
    p_out[0] = p[0];
    p_out[1] = p[1];
    p_out[2] = p[2];
    return fabsf(p[0] - pk0) > a.IP_dx || fabsf(p[1] - pk1) > a.IP_dx || fabsf(p[2] - pk2) > a.IP_dx;
}

// One alive ray: march up to n_step samples.  xyzs/dirs/deltas point at the ray slot's first sample.
__device__ inline uint32_t march_one(const MarchParams& a, int index, float noise, uint32_t n_step, float* __restrict__ xyzs,
                                     float* __restrict__ dirs, float* __restrict__ deltas) {
    const float ox = a.rays_o[index * 3], oy = a.rays_o[index * 3 + 1], oz = a.rays_o[index * 3 + 2];
    const float dx = a.rays_d[index * 3], dy = a.rays_d[index * 3 + 1], dz = a.rays_d[index * 3 + 2];
    const float rdx = 1 / dx, rdy = 1 / dy, rdz = 1 / dz;
    const uint32_t H = a.H, C = a.C;
    const float rH = 1 / (float)H;
    const float H3 = (float)(H * H * H);
    float t = a.rays_t[index];
    const float far = a.fars[index];
    const float dt_min = 2 * 1.7320508075688772f / a.max_steps;
    const float dt_max = 2 * 1.7320508075688772f * (1 << (C - 1)) / H;
    uint32_t step = 0;
    t += clampf(t * a.dt_gamma, dt_min, dt_max) * noise;
    float last_t = t;
    if (!(t < far)) return 0;

    const float bmin0 = a.bbmin[0], bmin1 = a.bbmin[1], bmin2 = a.bbmin[2];
    const float bmax0 = a.bbmax[0], bmax1 = a.bbmax[1], bmax2 = a.bbmax[2];
    const float hi0 = (float)((double)bmax0 - 1e-6), hi1 = (float)((double)bmax1 - 1e-6), hi2 = (float)((double)bmax2 - 1e-6);
    Search s{a.pig_cnt, a.pig_bgn, a.pig_idx, a.p_def, a.resolution[0], a.resolution[1], a.resolution[2], 0};
    s.n_grid = s.r0 * s.r1 * s.r2;  // n_grid = res[2]*res[1]*res[0] (nerf/renderer.py:814)

    while (t < far && step < n_step) {
        bool found = false;
        float x, y, z;
        if (a.cut) {
            x = clampf(ox + t * dx, -a.bound, a.bound);
            y = clampf(oy + t * dy, -a.bound, a.bound);
            z = clampf(oz + t * dz, -a.bound, a.bound);
        } else {
            x = clampf(ox + t * dx, bmin0, hi0);
            y = clampf(oy + t * dy, bmin1, hi1);
            z = clampf(oz + t * dz, bmin2, hi2);
        }
        bool in_cut = true;
        if (a.cut) {
            const float* cb = a.cut_bounds;  // `x < cb[3]` is the reference's own test (:1210)
            in_cut = (x > cb[0] && x < cb[1] && y > cb[2] && x < cb[3] && z > cb[4] && z < cb[5]);
        }
        if (in_cut) {
            float x_map = 0.0f, y_map = 0.0f, z_map = 0.0f;
            const int g0 = (int)floorf((x - bmin0) / a.hgs);
            const int g1 = (int)floorf((y - bmin1) / a.hgs);
            const int g2 = (int)floorf((z - bmin2) / a.hgs);
            const bool oob = (g0 < 0 || g1 < 0 || g2 < 0 || g0 >= s.r0 || g1 >= s.r1 || g2 >= s.r2);
            int IPs[3] = {-1, -1, -1};
            int n_IP = 0;
            if (oob) {
                if (a.err_flag) atomicOr(a.err_flag, 1);
            } else if (a.num_seek_IP == 1) {
                const int ip = find_closest_IP(s, x, y, z, g0, g1, g2);
                if (ip != -1) { n_IP = 1; IPs[0] = ip; }
            } else if (a.num_seek_IP == 2) {
                n_IP = find_closest_IPs<2>(s, x, y, z, g0, g1, g2, IPs);
            } else {
                n_IP = find_closest_IPs<3>(s, x, y, z, g0, g1, g2, IPs);
            }
            found = n_IP > 0;
            if (found) {
                for (int k = 0; k < n_IP; k++) {  // n_IP shrinks inside the loop it bounds (:1246-1251)
                    const float* pk_ = a.p_def + IPs[k] * 3;
                    if (pk_[0] <= bmin0 || pk_[1] <= bmin1 || pk_[2] < bmin2 || pk_[0] >= bmax0 || pk_[1] >= bmax1 || pk_[2] >= bmax2) n_IP--;
                }
            }
            if (n_IP <= 0) found = false;
            if (found) {
                float ps[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    if (k < n_IP) {  // same shrinking bound (:1316-1319)
                        float p[3];
                        if (warp_through_IP(a, IPs[k], x, y, z, p)) n_IP--;
                        ps[3 * k] = p[0];
                        ps[3 * k + 1] = p[1];
                        ps[3 * k + 2] = p[2];
                    }
                }
                if (n_IP == 1) {
                    x_map = ps[0]; y_map = ps[1]; z_map = ps[2];
                } else if (n_IP == 2) {
                    float dist[2];
#pragma unroll
                    for (int k = 0; k < 2; k++) {
                        const float* pk = a.p_ori + IPs[k] * 3;
                        dist[k] = sqrtf((pk[0] - x) * (pk[0] - x) + (pk[1] - y) * (pk[1] - y) + (pk[2] - z) * (pk[2] - z));
                    }
                    const float dist_sum = dist[0] + dist[1];
                    const float w0 = dist[1] / dist_sum, w1 = dist[0] / dist_sum;
                    x_map = w0 * ps[0] + w1 * ps[3];
                    y_map = w0 * ps[1] + w1 * ps[4];
                    z_map = w0 * ps[2] + w1 * ps[5];
                } else if (n_IP == 3) {
                    float dist[3];
#pragma unroll
                    for (int k = 0; k < 3; k++) {
                        const float* pk = a.p_ori + IPs[k] * 3;
                        dist[k] = sqrtf((pk[0] - x) * (pk[0] - x) + (pk[1] - y) * (pk[1] - y) + (pk[2] - z) * (pk[2] - z));
                    }
                    const float dist_sum = dist[0] * dist[1] + dist[1] * dist[2] + dist[2] * dist[0];
                    const float w0 = dist[1] * dist[2] / dist_sum;
                    const float w1 = dist[0] * dist[2] / dist_sum;
                    const float w2 = dist[0] * dist[1] / dist_sum;
                    x_map = w0 * ps[0] + w1 * ps[3] + w2 * ps[6];
                    y_map = w0 * ps[1] + w1 * ps[4] + w2 * ps[7];
                    z_map = w0 * ps[2] + w1 * ps[5] + w2 * ps[8];
                }
                x = x_map; y = y_map; z = z_map;  // n_IP == 0 here maps the sample to the origin (:1372-1374)
            }
        } else {
            found = true;  // cut mode, outside the cut box: un-warped background sample (:1380-1383)
        }

        const float dt = clampf(t * a.dt_gamma, dt_min, dt_max);
        const int level = max(mip_from_pos(x, y, z, (float)C), mip_from_dt(dt, (float)H, (float)C));
        const float mip_bound = fminf(scalbnf(1.0f, level), a.bound);
        const float mip_rbound = 1 / mip_bound;
        const int nx = (int)clampf((float)(0.5 * (double)(x * mip_rbound + 1) * (double)H), 0.0f, (float)(H - 1));
        const int ny = (int)clampf((float)(0.5 * (double)(y * mip_rbound + 1) * (double)H), 0.0f, (float)(H - 1));
        const int nz = (int)clampf((float)(0.5 * (double)(z * mip_rbound + 1) * (double)H), 0.0f, (float)(H - 1));
        const uint32_t vox = (uint32_t)(level * H3 + (float)morton3D(nx, ny, nz));
        const bool occ = a.grid[vox / 8] & (1 << (vox % 8));

        if (occ && found) {
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
            do { t += clampf(t * a.dt_gamma, dt_min, dt_max); } while (t < tt);
        }
    }
    return step;
}

}  // namespace pnm
