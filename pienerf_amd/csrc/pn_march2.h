// Side tables and per-point helpers of the ray march (gfx950): the candidate lists, the packed IP records, the Newton inverse
// warp through one record and the one-lane-per-ray skip over IP-free cells.  The march itself is in pn_march3.h.
//
// Per-frame side tables (built by k_frame_lists / k_nb_* + k_pack_ip in pn_render_ops.hip):
//   nb_bgn[n_grid+1], nb[...]  per cell: the candidates of its 27-cell neighbourhood as float4(p_def.xyz, bitcast id), in the
//                              reference's visiting order (own cell first, then NBR26; own-cell order = ascending id), so
//                              "position in the list" is "visiting order" and ties resolve exactly as the sequential scan does
//   rec[n_vtx][PN_REC_FLOATS]  packed IP record: p_ori(3) p_def(3) F^-1(9) pad(1) | F(9) dF(27) — 208 B, float4-aligned.  The head (first 64 B) is
//                              all a one-step Newton warp needs: iteration 0 solves with A = F_k, whose inverse depends on the IP alone, so
//                              it is computed ONCE per IP and frame by the packing kernel — with pnm::inv3x3, the same expression the march
//                              used to evaluate per (sample, IP), hence the same bits — instead of 3 x per evaluated ray point
//
// History: an earlier march let the 8 lanes of a ray share ONE evaluation (parallel candidate scan, packed u64
// (dist2 bits << 32 | position) keys merged by DPP min, the K warps on K lanes); it is described in DESIGN.md 4.1 and was
// replaced by the windowed form, which retires 8 (or 64) evaluations of a ray per round.
//
// Semantics are those of the oracle, bit for bit (same -ffp-contract=off arithmetic):
//   * Newton iteration 0 starts at q = +0, where dF.q = 0 and mul31(F, q) = 0: it is evaluated as A = F, b = -q'
//     (identical results for finite F, dF; dF is only loaded if a second iteration runs).
#pragma once
#include "pn_march.h"

namespace pnm2 {
using namespace pnm;

#define PN_G 8  // lanes per ray in the first march launch
#define PN_REC_FLOATS 52
#define PN_REC_VEC4 (PN_REC_FLOATS / 4)

// One float of the packed record of IP `ip` (see the header comment); Finv = inv3x3(F) or all zeros when det F == 0 (raymarching.cu:1285-1287:
// the reference never acts on the failure code, its A_inv stays 0).
__device__ __forceinline__ float pack_ip_float(int j, int ip, const float* __restrict__ p_ori, const float* __restrict__ p_def, const float* __restrict__ F_IP,
                                               const float* __restrict__ dF_IP) {
    if (j < 3) return p_ori[ip * 3 + j];
    if (j < 6) return p_def[ip * 3 + j - 3];
    if (j < 15) {
        float Fk[9], Ai[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int q = 0; q < 9; q++) Fk[q] = F_IP[ip * 9 + q];
        inv3x3(Fk, Ai);
        float v = 0.f;
#pragma unroll
        for (int q = 0; q < 9; q++) if (q == j - 6) v = Ai[q];
        return v;
    }
    if (j < 16) return 0.f;
    if (j < 25) return F_IP[ip * 9 + j - 16];
    return dF_IP[ip * 27 + j - 25];
}

struct March2Tables {
    const int* nb_bgn;    // [n_grid + 1]
    const float4* nb;     // candidate entries
    const float4* rec;    // [n_vtx * PN_REC_VEC4]
};

// Newton inverse warp through one packed IP record (raymarching.cu:1262-1324).  Returns the reject flag.
// MULTI = false is the max_iter_num <= 1 build (the chair / trex demo setting, README.md:123,134): no dF, far fewer registers.
template <bool MULTI>
// `h` = the record's first four float4 (p_ori, p_def, F^-1), already loaded by the caller; `r` = the record (F and dF are read from it
// only if a second Newton step runs).
__device__ inline bool warp_record(const float4 (&h)[4], const float4* __restrict__ r, int max_iter_num, float IP_dx, float x, float y, float z,
                                   float* p_out, float* dist_out) {
    const float4 r0 = h[0], r1 = h[1], r2 = h[2], r3 = h[3];
    const float pk0 = r0.x, pk1 = r0.y, pk2 = r0.z;          // p_ori
    const float pd0 = r0.w, pd1 = r1.x, pd2 = r1.y;          // p_def
    const float A_inv[9] = {r1.z, r1.w, r2.x, r2.y, r2.z, r2.w, r3.x, r3.y, r3.z};  // inv3x3(F_k), precomputed per IP
    float p[3] = {pk0, pk1, pk2};
    const float q_[3] = {x - pd0, y - pd1, z - pd2};
    int num_itr = 0;
    if (max_iter_num > 0) {
        // iteration 0: q = p - pk = +0  =>  dFk_q = 0, A = Fk, b = (0 + 0.5*0) - q_ = -q_
        float b[3], dq[3];
#pragma unroll
        for (int i = 0; i < 3; i++) b[i] = (float)(-(double)q_[i]);
        mul31(A_inv, b, dq);
        p[0] -= dq[0];
        p[1] -= dq[1];
        p[2] -= dq[2];
        const bool conv = (double)(dq[0] * dq[0] + dq[1] * dq[1] + dq[2] * dq[2]) < 1e-12;
        num_itr = 1;
        if (MULTI && !conv && max_iter_num > 1) {
            float Fk[9], dFk[27];
            const float* rf = reinterpret_cast<const float*>(r);
#pragma unroll
            for (int j = 0; j < 9; j++) Fk[j] = rf[16 + j];
#pragma unroll
            for (int j = 0; j < 27; j++) dFk[j] = rf[25 + j];
            while (num_itr < max_iter_num) {
                const float q[3] = {p[0] - pk0, p[1] - pk1, p[2] - pk2};
                float dFk_q[9], A[9], Ai[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, Fk_q[3], dFk_q_q[3];
                dot31(dFk, q, dFk_q);
#pragma unroll
                for (int j = 0; j < 9; j++) A[j] = Fk[j] + dFk_q[j];
                inv3x3(A, Ai);
                mul31(Fk, q, Fk_q);
                mul31(dFk_q, q, dFk_q_q);
#pragma unroll
                for (int i = 0; i < 3; i++) b[i] = (float)(((double)Fk_q[i] + 0.5 * (double)dFk_q_q[i]) - (double)q_[i]);
                mul31(Ai, b, dq);
                p[0] -= dq[0];
                p[1] -= dq[1];
                p[2] -= dq[2];
                if ((double)(dq[0] * dq[0] + dq[1] * dq[1] + dq[2] * dq[2]) < 1e-12) break;
                num_itr++;
            }
        }
    }
    p_out[0] = p[0];
    p_out[1] = p[1];
    p_out[2] = p[2];
    *dist_out = sqrtf((pk0 - x) * (pk0 - x) + (pk1 - y) * (pk1 - y) + (pk2 - z) * (pk2 - z));  // blend weight distance (:1345,1362)
    return fabsf(p[0] - pk0) > IP_dx || fabsf(p[1] - pk1) > IP_dx || fabsf(p[2] - pk2) > IP_dx;
}

// Leading run of marching iterations that can neither emit nor warp, ONE lane per ray: search cells whose 27-neighbourhood holds no IP
// and, in --cut mode, static-background points (outside the cut box) in empty density voxels.  In --cut mode every ray crosses the
// whole +-bound volume, so without this pre-pass all of those iterations went through the 8-lane / wave-per-ray kernels
// (trex option set at 1008x756: 65.7 M visited points per frame).
// At 800x800 ~86 % of the first trip's iterations are of this kind (rays crossing the empty part of the IP bounding box): no
// candidate is found, so the sample is not warped and the ray just hops to the next density-grid voxel.  They need no memory but
// the cell's list range, and in the cooperative kernel 7 of 8 lanes would replicate them.  Returns the t at which march_group
// has to take over (first iteration whose cell has candidates, or t >= far); the arithmetic is march_group's, expression by
// expression, so resuming there is bit-identical to having run every iteration in march_group.
__device__ inline float skip_empty_cells(const MarchParams& a, const March2Tables& tb, int index, float noise, unsigned* n_iter_out) {
    const float ox = a.rays_o[index * 3], oy = a.rays_o[index * 3 + 1], oz = a.rays_o[index * 3 + 2];
    const float dx = a.rays_d[index * 3], dy = a.rays_d[index * 3 + 1], dz = a.rays_d[index * 3 + 2];
    const uint32_t H = a.H, C = a.C;
    const float far = a.fars[index];
    const float dt_min = 2 * 1.7320508075688772f / a.max_steps;
    const float dt_max = 2 * 1.7320508075688772f * (1 << (C - 1)) / H;
    *n_iter_out = 0;
    float t = a.rays_t[index];
    t += clampf(t * a.dt_gamma, dt_min, dt_max) * noise;
    if (!(t < far)) return t;
    const float rdx = 1 / dx, rdy = 1 / dy, rdz = 1 / dz;
    const float rH = 1 / (float)H;
    const float H3 = (float)(H * H * H);
    const bool cut = a.cut != 0;
    const float bmin0 = a.bbmin[0], bmin1 = a.bbmin[1], bmin2 = a.bbmin[2];
    // clamp range of the sample point: +-bound in --cut mode (raymarching.cu:1196-1199), [bbmin, bbmax - 1e-6] otherwise (:1203-1205)
    const float lo0 = cut ? -a.bound : bmin0, lo1 = cut ? -a.bound : bmin1, lo2 = cut ? -a.bound : bmin2;
    const float hi0 = cut ? a.bound : (float)((double)a.bbmax[0] - 1e-6), hi1 = cut ? a.bound : (float)((double)a.bbmax[1] - 1e-6),
                hi2 = cut ? a.bound : (float)((double)a.bbmax[2] - 1e-6);
    float cb[6] = {0, 0, 0, 0, 0, 0};
    if (cut)
        for (int i = 0; i < 6; i++) cb[i] = a.cut_bounds[i];
    const int r0 = a.resolution[0], r1 = a.resolution[1], r2 = a.resolution[2];
    const float rbound = 1 / a.bound;
    const float halfH = 0.5f * (float)H;
    int cell_id = -1;
    unsigned n_iter = 0;
    while (t < far) {
        const float x = clampf(ox + t * dx, lo0, hi0);
        const float y = clampf(oy + t * dy, lo1, hi1);
        const float z = clampf(oz + t * dz, lo2, hi2);
        // --cut: a point outside the cut box is a static-background sample (found = true, un-warped, :1380-1383); the cut test is the
        // reference's, `x < cut_bounds[3]` included (:1210)
        const bool searched = !cut || (x > cb[0] && x < cb[1] && y > cb[2] && x < cb[3] && z > cb[4] && z < cb[5]);
        if (searched) {
        const int g0 = (int)floorf((x - bmin0) / a.hgs);
        const int g1 = (int)floorf((y - bmin1) / a.hgs);
        const int g2 = (int)floorf((z - bmin2) / a.hgs);
        if (g0 < 0 || g1 < 0 || g2 < 0 || g0 >= r0 || g1 >= r1 || g2 >= r2) break;  // march_group raises the error flag
        const int gid = g2 * r1 * r0 + g1 * r0 + g0;
        if (gid != cell_id) {
            if (tb.nb_bgn[gid] != tb.nb_bgn[gid + 1]) break;  // candidates: hand over
            cell_id = gid;
        }
        }
        // found == false (or a static sample in an empty voxel): un-warped voxel skip (raymarching.cu:1386-1428)
        const float dt = clampf(t * a.dt_gamma, dt_min, dt_max);
        const int level = max(mip_from_pos(x, y, z, (float)C), mip_from_dt(dt, (float)H, (float)C));
        const float pw = scalbnf(1.0f, level);
        const bool use_pw = pw <= a.bound;
        const float mip_bound = use_pw ? pw : a.bound;
        const float mip_rbound = use_pw ? scalbnf(1.0f, -level) : rbound;
        // (float)(0.5 * (double)v * (double)H) == v * (0.5f * H): both round the exact product once (v has 24 significant bits, H < 2^24)
        const int nx = (int)clampf((x * mip_rbound + 1) * halfH, 0.0f, (float)(H - 1));
        const int ny = (int)clampf((y * mip_rbound + 1) * halfH, 0.0f, (float)(H - 1));
        const int nz = (int)clampf((z * mip_rbound + 1) * halfH, 0.0f, (float)(H - 1));
        if (!searched) {  // static sample: emitted when its voxel is occupied -> hand over to march_group at this element
            const uint32_t vox = (uint32_t)(level * H3 + (float)morton3D(nx, ny, nz));
            if (a.grid[vox / 8] & (1 << (vox % 8))) break;
        }
        n_iter++;
        const float tx = (((nx + 0.5f + 0.5f * signf(dx)) * rH * 2 - 1) * mip_bound - x) * rdx;
        const float ty = (((ny + 0.5f + 0.5f * signf(dy)) * rH * 2 - 1) * mip_bound - y) * rdy;
        const float tz = (((nz + 0.5f + 0.5f * signf(dz)) * rH * 2 - 1) * mip_bound - z) * rdz;
        const float tt = t + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
        do { t += clampf(t * a.dt_gamma, dt_min, dt_max); } while (t < tt);
    }
    *n_iter_out = n_iter;
    return t;
}

}  // namespace pnm2
