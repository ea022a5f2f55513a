// Cooperative ray march with the inverse-GMLS warp: 8 lanes per ray (gfx950, wave = 64 -> 8 rays per wave).
//
// Why: at 800x800 only ~70k rays cross the IP bounding box — one wave per SIMD with one lane per ray — and every
// marching step is a chain of dependent loads (27-cell stencil -> point ids -> positions -> F/dF -> bitfield).  The
// one-lane-per-ray form of the reference (raymarching.cu:1121-1434) is therefore latency-bound with nothing to overlap.
// Here the 8 lanes of a group (a) scan the candidate list of the sample's cell in parallel and merge their top-K by
// xor-shuffles, (b) run the <= 3 per-IP Newton warps on lanes 0..2 concurrently, (c) replicate the cheap scalar logic;
// 8x more waves are resident and the serial depth of a step drops from ~100 loads to ~6.
//
// Per-frame side tables (built by k_nb_* / k_pack_ip in pn_render_ops.hip):
//   nb_bgn[n_grid+1], nb[...]  per cell: the candidates of its 27-cell neighbourhood as float4(p_def.xyz, bitcast id), in the
//                              reference's visiting order (own cell first, then NBR26; own-cell order = ascending id), so
//                              "position in the list" is "visiting order" and ties resolve exactly as the sequential scan does
//   rec[n_vtx][44]             packed IP record: p_ori(3) p_def(3) F(9) dF(27) pad(2) — 176 B, float4-aligned
//
// Semantics are those of pn_march.h / the oracle, bit for bit (same -ffp-contract=off arithmetic):
//   * sequential insertion with strict '<' over candidates in visiting order  ==  top-K by the key (dist2, position);
//   * `n_IP--` inside the loops it bounds is replayed on the gathered per-IP flags;
//   * Newton iteration 0 starts at q = +0, where dF.q = 0 and mul31(F, q) = 0: it is evaluated as A = F, b = -q'
//     (identical results for finite F, dF; dF is only loaded if a second iteration runs).
#pragma once
#include "pn_march.h"

namespace pnm2 {
using namespace pnm;

#define PN_G 8  // lanes per ray
#ifndef PN_TAIL_ITERS
#define PN_TAIL_ITERS 12
#define PN_TAIL_PRIO 3
#endif

struct March2Tables {
    const int* nb_bgn;    // [n_grid + 1]
    const float4* nb;     // candidate entries
    const float4* rec;    // [n_vtx * 11]
};

// Candidate key: (dist2 bits << 32) | list position.  dist2 >= 0, so its IEEE bit pattern orders like its value, and the
// u64 order is exactly the lexicographic (dist2, visiting order) order that sequential strict-'<' insertion realises.
typedef unsigned long long key_t;
#define PN_KEY_NONE 0xFFFFFFFFFFFFFFFFull
__device__ __forceinline__ key_t make_key(float d, int ord) { return ((key_t)__float_as_uint(d) << 32) | (unsigned)ord; }
// DPP lane exchange inside a row of 16: all 8 lanes of a group are active together, so every source lane is live.
template <int CTRL>
__device__ __forceinline__ key_t dpp_key(key_t k) {
    const int lo = (int)(unsigned)k, hi = (int)(k >> 32);
    const unsigned rlo = (unsigned)__builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, false);
    const unsigned rhi = (unsigned)__builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, false);
    return ((key_t)rhi << 32) | rlo;
}

// Group-cooperative scan of nb[b..e): every lane of the group returns the same top-K list positions (-1 if none).
// dinit = FLT_MAX (find_closest_IPs, raymarching.cu:1056) or 9999.9f (find_closest_IP, :997): only d < dinit is accepted.
template <int K>
__device__ __forceinline__ void group_topk(const float4* __restrict__ nb, int b, int e, int sub, float x, float y, float z, float dinit, int* ord_out) {
    key_t k0 = PN_KEY_NONE, k1 = PN_KEY_NONE, k2 = PN_KEY_NONE;  // this lane's sorted best
    // four list entries per lane are fetched before any is consumed (clamped index, masked afterwards): one memory round
    // trip covers 32 candidates instead of 8
    for (int j0 = b + sub; j0 < e; j0 += 4 * PN_G) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) v[u] = nb[min(j0 + u * PN_G, e - 1)];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int j = j0 + u * PN_G;
            const float ax = v[u].x - x, ay = v[u].y - y, az = v[u].z - z;
            const float d = ax * ax + ay * ay + az * az;  // (pk_[0]-x)*(pk_[0]-x) + ... (raymarching.cu:1002)
            if (j < e && d < dinit) {  // `d < dinit` also rejects NaN, like the reference's `dist2_tmp < dist2`
                const key_t k = make_key(d, j);
                if (k < k0) { k2 = k1; k1 = k0; k0 = k; }
                else if (K > 1 && k < k1) { k2 = k1; k1 = k; }
                else if (K > 2 && k < k2) { k2 = k; }
            }
        }
    }
    // K rounds of group-min (DPP quad-perm / half-row mirror: VALU latency, no LDS crossbar); the owner of the winner pops it
#pragma unroll
    for (int r = 0; r < 3; r++) {
        if (r < K) {
            key_t m = k0;
            { key_t o = dpp_key<0xB1>(m); m = o < m ? o : m; }   // lane ^ 1
            { key_t o = dpp_key<0x4E>(m); m = o < m ? o : m; }   // lane ^ 2
            { key_t o = dpp_key<0x141>(m); m = o < m ? o : m; }  // lane <-> 7 - lane within each 8 (row_half_mirror)
            ord_out[r] = (m == PN_KEY_NONE) ? -1 : (int)(unsigned)m;
            if (m == k0 && m != PN_KEY_NONE) { k0 = k1; k1 = k2; k2 = PN_KEY_NONE; }
        } else {
            ord_out[r] = -1;
        }
    }
}

// Newton inverse warp through one packed IP record (raymarching.cu:1262-1324).  Returns the reject flag.
// MULTI = false is the max_iter_num <= 1 build (the chair / trex demo setting, README.md:123,134): no dF, far fewer registers.
template <bool MULTI>
__device__ inline bool warp_record(const float4* __restrict__ r, int max_iter_num, float IP_dx, float x, float y, float z, float* p_out, float* dist_out) {
    const float4 r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3];
    const float pk0 = r0.x, pk1 = r0.y, pk2 = r0.z;          // p_ori
    const float pd0 = r0.w, pd1 = r1.x, pd2 = r1.y;          // p_def
    const float Fk[9] = {r1.z, r1.w, r2.x, r2.y, r2.z, r2.w, r3.x, r3.y, r3.z};
    float p[3] = {pk0, pk1, pk2};
    const float q_[3] = {x - pd0, y - pd1, z - pd2};
    int num_itr = 0;
    if (max_iter_num > 0) {
        // iteration 0: q = p - pk = +0  =>  dFk_q = 0, A = Fk, b = (0 + 0.5*0) - q_ = -q_
        float A_inv[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, b[3], dq[3];
        inv3x3(Fk, A_inv);
#pragma unroll
        for (int i = 0; i < 3; i++) b[i] = (float)(-(double)q_[i]);
        mul31(A_inv, b, dq);
        p[0] -= dq[0];
        p[1] -= dq[1];
        p[2] -= dq[2];
        const bool conv = (double)(dq[0] * dq[0] + dq[1] * dq[1] + dq[2] * dq[2]) < 1e-12;
        num_itr = 1;
        if (MULTI && !conv && max_iter_num > 1) {
            float dFk[27];
            const float* rf = reinterpret_cast<const float*>(r);
#pragma unroll
            for (int j = 0; j < 27; j++) dFk[j] = rf[15 + j];
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

// Leading run of marching iterations through search cells whose 27-neighbourhood holds no IP, ONE lane per ray.
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
    if (!(t < far) || a.cut) return t;
    const float rdx = 1 / dx, rdy = 1 / dy, rdz = 1 / dz;
    const float rH = 1 / (float)H;
    const float bmin0 = a.bbmin[0], bmin1 = a.bbmin[1], bmin2 = a.bbmin[2];
    const float hi0 = (float)((double)a.bbmax[0] - 1e-6), hi1 = (float)((double)a.bbmax[1] - 1e-6), hi2 = (float)((double)a.bbmax[2] - 1e-6);
    const int r0 = a.resolution[0], r1 = a.resolution[1], r2 = a.resolution[2];
    const float rbound = 1 / a.bound;
    const float halfH = 0.5f * (float)H;
    int cell_id = -1;
    unsigned n_iter = 0;
    while (t < far) {
        const float x = clampf(ox + t * dx, bmin0, hi0);
        const float y = clampf(oy + t * dy, bmin1, hi1);
        const float z = clampf(oz + t * dz, bmin2, hi2);
        const int g0 = (int)floorf((x - bmin0) / a.hgs);
        const int g1 = (int)floorf((y - bmin1) / a.hgs);
        const int g2 = (int)floorf((z - bmin2) / a.hgs);
        if (g0 < 0 || g1 < 0 || g2 < 0 || g0 >= r0 || g1 >= r1 || g2 >= r2) break;  // march_group raises the error flag
        const int gid = g2 * r1 * r0 + g1 * r0 + g0;
        if (gid != cell_id) {
            if (tb.nb_bgn[gid] != tb.nb_bgn[gid + 1]) break;  // candidates: hand over
            cell_id = gid;
        }
        n_iter++;
        // found == false: un-warped voxel skip (raymarching.cu:1386-1428)
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
        const float tx = (((nx + 0.5f + 0.5f * signf(dx)) * rH * 2 - 1) * mip_bound - x) * rdx;
        const float ty = (((ny + 0.5f + 0.5f * signf(dy)) * rH * 2 - 1) * mip_bound - y) * rdy;
        const float tz = (((nz + 0.5f + 0.5f * signf(dz)) * rH * 2 - 1) * mip_bound - z) * rdz;
        const float tt = t + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
        do { t += clampf(t * a.dt_gamma, dt_min, dt_max); } while (t < tt);
    }
    *n_iter_out = n_iter;
    return t;
}

// One ray, executed by its 8 lanes in lock step.  `sub` = lane within the group; all per-ray state is replicated.
// Returns the number of samples emitted (same value on all 8 lanes); lane 0 of the group writes them.
template <int K, bool MULTI>
// `resume` (may be null): t left by skip_empty_cells for this ray; the loop starts there, `last_t` keeps the trip's start.
__device__ inline uint32_t march_group(const MarchParams& a, const March2Tables& tb, int index, float noise, uint32_t n_step, int sub, int gbase,
                                       float* __restrict__ xyzs, float* __restrict__ dirs, float* __restrict__ deltas, const float* resume) {
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
    if (resume) t = *resume;

    const float bmin0 = a.bbmin[0], bmin1 = a.bbmin[1], bmin2 = a.bbmin[2];
    const float bmax0 = a.bbmax[0], bmax1 = a.bbmax[1], bmax2 = a.bbmax[2];
    const float hi0 = (float)((double)bmax0 - 1e-6), hi1 = (float)((double)bmax1 - 1e-6), hi2 = (float)((double)bmax2 - 1e-6);
    const int r0 = a.resolution[0], r1 = a.resolution[1], r2 = a.resolution[2];

    const float rbound = 1 / a.bound;
    const float halfH = 0.5f * (float)H;
    int cell_id = -1, cell_b = 0, cell_e = 0;
    unsigned n_iter = 0, n_cand = 0, n_warp = 0;  // instrumentation, only reported when a.stats != nullptr
    while (t < far && step < n_step) {
        n_iter++;
        // A ray that is still marching after many iterations is on the trip's critical path (a few hundred rays need 60-90
        // serial iterations while the rest of the launch has long finished): its wave asks the SIMD arbiter for priority over
        // the wide, throughput-bound waves of other frames in flight.
        if (n_iter == PN_TAIL_ITERS) __builtin_amdgcn_s_setprio(PN_TAIL_PRIO);
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
            const float* cb = a.cut_bounds;  // `x < cb[3]` is the reference's own test (raymarching.cu:1210)
            in_cut = (x > cb[0] && x < cb[1] && y > cb[2] && x < cb[3] && z > cb[4] && z < cb[5]);
        }
        if (in_cut) {
            float x_map = 0.0f, y_map = 0.0f, z_map = 0.0f;
            const int g0 = (int)floorf((x - bmin0) / a.hgs);
            const int g1 = (int)floorf((y - bmin1) / a.hgs);
            const int g2 = (int)floorf((z - bmin2) / a.hgs);
            const bool oob = (g0 < 0 || g1 < 0 || g2 < 0 || g0 >= r0 || g1 >= r1 || g2 >= r2);
            int ord[3] = {-1, -1, -1};
            if (oob) {
                if (a.err_flag && sub == 0) atomicOr(a.err_flag, 1);
            } else {
                const int gid = g2 * r1 * r0 + g1 * r0 + g0;
                // consecutive steps usually stay in one 1.2*dx cell (a voxel skip is ~1/4 of it): reuse its list range, so a
                // run of steps through IP-free space costs no memory access at all
                if (gid != cell_id) { cell_id = gid; cell_b = tb.nb_bgn[gid]; cell_e = tb.nb_bgn[gid + 1]; }
                const int b = cell_b, e = cell_e;
                n_cand += (unsigned)(e - b);
                if (b == e) {
                    // no IP in the 27-cell neighbourhood: nothing found
                } else if (K == 1) {  // find_closest_IP: own cell first, the 26 neighbours only if that found nothing (:986-1043)
                    const int own = a.pig_cnt[gid];
                    if (own > 0) group_topk<1>(tb.nb, b, b + own, sub, x, y, z, (float)9999.9, ord);
                    if (ord[0] == -1) group_topk<1>(tb.nb, b + own, e, sub, x, y, z, (float)9999.9, ord);
                } else {
                    group_topk<K>(tb.nb, b, e, sub, x, y, z, FLT_MAX, ord);
                }
            }
            int n_IP = (ord[0] != -1) + (ord[1] != -1) + (ord[2] != -1);
            found = n_IP > 0;
            if (found) {
                // lanes 0..K-1 each own one selected IP: pre-filter flag, Newton warp, reject flag, blend distance
                const int mine = (sub < 3) ? ((sub == 0) ? ord[0] : (sub == 1 ? ord[1] : ord[2])) : -1;
                float pw[3] = {0.f, 0.f, 0.f}, dist = 0.f;
                int flags = 0;  // bit0: pre-filter hit, bit1: reject
                if (mine != -1) {
                    n_warp++;
                    const float4 c = tb.nb[mine];
                    const int ip = __float_as_int(c.w);
                    if (c.x <= bmin0 || c.y <= bmin1 || c.z < bmin2 || c.x >= bmax0 || c.y >= bmax1 || c.z >= bmax2) flags |= 1;  // (:1249)
                    if (warp_record<MULTI>(tb.rec + (size_t)ip * 11, a.max_iter_num, a.IP_dx, x, y, z, pw, &dist)) flags |= 2;
                }
                float ps[9], dk[3];
                int fl[3];
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    if (k < K) {
                        ps[3 * k] = __shfl(pw[0], gbase + k);
                        ps[3 * k + 1] = __shfl(pw[1], gbase + k);
                        ps[3 * k + 2] = __shfl(pw[2], gbase + k);
                        dk[k] = __shfl(dist, gbase + k);
                        fl[k] = __shfl(flags, gbase + k);
                    } else {
                        ps[3 * k] = ps[3 * k + 1] = ps[3 * k + 2] = 0.f;
                        dk[k] = 0.f;
                        fl[k] = 0;
                    }
                }
                // replay of the two loops whose bound shrinks inside them (:1246-1251, :1262-1324)
#pragma unroll
                for (int k = 0; k < 3; k++) if (k < n_IP && (fl[k] & 1)) n_IP--;
                if (n_IP <= 0) found = false;
                if (found) {
                    float pz[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                    for (int k = 0; k < 3; k++) {
                        if (k < n_IP) {
                            if (fl[k] & 2) n_IP--;
                            pz[3 * k] = ps[3 * k];
                            pz[3 * k + 1] = ps[3 * k + 1];
                            pz[3 * k + 2] = ps[3 * k + 2];
                        }
                    }
                    if (n_IP == 1) {
                        x_map = pz[0]; y_map = pz[1]; z_map = pz[2];
                    } else if (n_IP == 2) {
                        const float dist_sum = dk[0] + dk[1];
                        const float w0 = dk[1] / dist_sum, w1 = dk[0] / dist_sum;
                        x_map = w0 * pz[0] + w1 * pz[3];
                        y_map = w0 * pz[1] + w1 * pz[4];
                        z_map = w0 * pz[2] + w1 * pz[5];
                    } else if (n_IP == 3) {
                        const float dist_sum = dk[0] * dk[1] + dk[1] * dk[2] + dk[2] * dk[0];
                        const float w0 = dk[1] * dk[2] / dist_sum;
                        const float w1 = dk[0] * dk[2] / dist_sum;
                        const float w2 = dk[0] * dk[1] / dist_sum;
                        x_map = w0 * pz[0] + w1 * pz[3] + w2 * pz[6];
                        y_map = w0 * pz[1] + w1 * pz[4] + w2 * pz[7];
                        z_map = w0 * pz[2] + w1 * pz[5] + w2 * pz[8];
                    }
                    x = x_map; y = y_map; z = z_map;  // n_IP == 0 here maps the sample to the origin (:1372-1374)
                }
            }
        } else {
            found = true;  // cut mode, outside the cut box: un-warped background sample (:1380-1383)
        }

        const float dt = clampf(t * a.dt_gamma, dt_min, dt_max);
        const int level = max(mip_from_pos(x, y, z, (float)C), mip_from_dt(dt, (float)H, (float)C));
        // mip_bound = fminf(2^level, bound); 1 / mip_bound is exact for the power of two and loop-invariant for `bound`
        const float pw = scalbnf(1.0f, level);
        const bool use_pw = pw <= a.bound;
        const float mip_bound = use_pw ? pw : a.bound;
        const float mip_rbound = use_pw ? scalbnf(1.0f, -level) : rbound;
        // the reference's (float)(0.5 * (double)v * (double)H) rounds the exact product v*H/2 once (v: 24 significant bits, H < 2^24,
        // so the double products are exact); so does the float product v * (0.5f*H) — same value without the fp64 pipe
        const int nx = (int)clampf((x * mip_rbound + 1) * halfH, 0.0f, (float)(H - 1));
        const int ny = (int)clampf((y * mip_rbound + 1) * halfH, 0.0f, (float)(H - 1));
        const int nz = (int)clampf((z * mip_rbound + 1) * halfH, 0.0f, (float)(H - 1));
        const uint32_t vox = (uint32_t)(level * H3 + (float)morton3D(nx, ny, nz));
        // the occupancy bit only matters when an IP was found (`occ && found`), so the load is skipped otherwise
        const bool occ = found ? (bool)(a.grid[vox / 8] & (1 << (vox % 8))) : false;

        if (occ && found) {
            t += dt;
            if (sub == 0) {
                xyzs[0] = x; xyzs[1] = y; xyzs[2] = z;
                dirs[0] = dx; dirs[1] = dy; dirs[2] = dz;
                deltas[0] = dt;
                deltas[1] = t - last_t;
            }
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
    if (a.stats) {
        if (sub == 0) { atomicAdd(a.stats, (unsigned long long)n_iter); atomicAdd(a.stats + 1, (unsigned long long)n_cand); atomicAdd(a.stats + 3, (unsigned long long)step); }
        if (n_warp) atomicAdd(a.stats + 2, (unsigned long long)n_warp);
    }
    return step;
}

}  // namespace pnm2
