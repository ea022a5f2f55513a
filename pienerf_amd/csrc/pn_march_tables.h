// Side tables and per-point helpers of the ray march (gfx950): the candidate lists, the packed IP records, the Newton inverse
// warp through one record.  The march itself (windows, skip pre-pass) is in pn_march_window.h.
//
// Per-frame side tables (built by k_frame_lists / k_nb_* + k_pack_ip in pn_render_ops.hip):
//   nb_rng[n_grid], nb[...]    per cell (begin, end) of: the candidates of its 27-cell neighbourhood as float4(p_def.xyz, bitcast id), in the
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
#include "pn_march_math.h"

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
    const int2* nb_rng;   // [n_grid] candidates of the cell's 27-neighbourhood: nb[x .. y)
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

}  // namespace pnm2
