// Windowed ray march with the inverse-GMLS warp: G lanes per ray (G = 8 or 64), each lane evaluating ONE point of the ray.
//
// The reference loop (raymarching.cu:1121-1434) looks serial — evaluate the sample at t, then either emit it (t += dt) or hop
// to the end of its density voxel (do t += dt while t < tt) — but both branches advance t by the same recurrence
//     s_{k+1} = s_k + clamp(s_k * dt_gamma, dt_min, dt_max)          (float, rounded once per add)
// so the t values a ray can ever visit are the fixed sequence s_0 = t_start, s_1, s_2, ...  Which of them ARE visited depends on
// the evaluations, but the evaluation at s_k (search cell -> nearest IPs -> Newton warps -> blend -> density bit -> voxel exit)
// is a pure function of s_k.  So per round the G lanes of a ray evaluate s_0..s_{G-1} independently, each lane then walks its
// own voxel hop to find the index it would jump to, and every lane replays the visit chain 0 -> jump[0] -> jump[jump[0]] ...
// (G = 8: the 8 jump indices are OR-packed into one dword by DPP; G = 64: the wave-uniform chain is walked with scalar registers and
// v_readlane).  Visited lanes that found an occupied sample write it to the slot given by their rank in the chain.  The round loop is
// wave-uniform: the candidate lists and IP record heads the wave's points need are staged cooperatively in LDS (stage_lists, head_fetch).
//
// Two launches per loop trip use it (pn_render_ops.hip):
//   k_march       G = 8, 8 rays per wave, at most `max_rounds` rounds per ray (2 on a frame's first trip, 1 afterwards): a ray in a
//                 sample-dense region emits its 8 samples in one round.  Rays that are still going after the budget (they graze the
//                 object or have left it and hop through 60-90 voxels without emitting) are appended to a segmented tail list with
//                 their state and ray constants;
//   k_march_tail  G = 64, one wave per listed ray (handed out dynamically, long ones first): 64 sequence elements (~14 voxel hops)
//                 per round, so the critical path of a trip is ~6 rounds instead of ~80 dependent iterations.
// vs. the cooperative form (pn_march_tables.h: 8 lanes share ONE evaluation, one iteration at a time) the results are identical bit
// for bit: same -ffp-contract=off expressions, sequential strict-'<' insertion over the candidate list in the reference's
// visiting order, the `n_IP--` loops replayed literally.
#pragma once
#include "pn_march_tables.h"

namespace pnm3 {
using namespace pnm;
using pnm2::March2Tables;
using pnm2::warp_record;

#ifndef PN_CAND_FLIGHT
#define PN_CAND_FLIGHT 4  // candidate-list entries a lane has in flight per memory round trip (8: +16 VGPRs, no measurable gain)
#endif

struct RayConsts {
    float ox, oy, oz, dx, dy, dz, rdx, rdy, rdz;
    float bmin0, bmin1, bmin2, bmax0, bmax1, bmax2, hi0, hi1, hi2;
    int r0, r1, r2;
    float rH, H3, halfH, Hm1, Hf, Cf, rbound, dt_min, dt_max, far;
};

// Marching state of one ray inside a trip (what the reference keeps in t / last_t / step).
struct RayState {
    float t, last_t;
    uint32_t step;
};

__device__ __forceinline__ float dtf(const MarchParams& a, const RayConsts& c, float t) { return clampf(t * a.dt_gamma, c.dt_min, c.dt_max); }

// Fixed-step lattice arithmetic.  With dt_gamma == 0 (the chair option set) every step adds the same float D, and inside one binade
// [2^e, 2^(e+1)) — where every float is a multiple of u = 2^(e-23) — the rounded sum fl(s + D) is s + Dq with Dq = D rounded to the nearest
// multiple of u, for EVERY s of the binade whose successor stays inside it (the exact sum is s + Dq + r, |r| < u/2, unless D sits exactly
// half-way between two multiples: then round-to-even depends on s and `ok` is false).  So s_k = t + k * Dq, with the product and the sum
// both exact (k * (Dq / u) < 2^24 is part of `ok`), replaces the k-step recurrence, and "first element >= tt" is a rounded quotient
// corrected by two exact comparisons.  Anything else — a window that reaches the binade's top, a tie, dt_gamma != 0 — takes the
// reference's own step-by-step loops.
struct Binade {
    float Dq, rDq, top;
    bool ok;
};
template <int G>
__device__ __forceinline__ Binade binade_of(float t, float D) {
    Binade b;
    const uint32_t bits = __float_as_uint(t);
    const int E = (int)(bits >> 23);         // biased exponent; a negative t has the sign bit here and fails the range test
    const float Ds = scalbnf(D, 150 - E);    // D in units of u: a power-of-two scaling, exact
    const float nf = rintf(Ds);
    b.ok = E > 30 && E < 250 && fabsf(Ds - nf) != 0.5f && nf >= 1.0f && nf * (float)(G + 2) < 16777216.0f;
    b.Dq = scalbnf(nf, E - 150);
    b.rDq = __builtin_amdgcn_rcpf(b.Dq);
    b.top = __uint_as_float((uint32_t)(E + 1) << 23);
    return b;
}
// First k >= k_min with t + k * Dq >= tt, or k_cap + 1 when there is none up to k_cap (all of t + k * Dq, k <= k_cap, lie inside the binade).
__device__ __forceinline__ int lattice_first_at_least(const Binade& b, float t, float tt, int k_min, int k_cap) {
    // NaN tt: fmaxf drops it -> k_min, as the reference's `u < tt` loop (false at once) does
    int k = (int)fminf(fmaxf(ceilf((tt - t) * b.rDq), (float)k_min), (float)(k_cap + 1));
    if (k > k_min && t + (float)(k - 1) * b.Dq >= tt) k--;  // the quotient is off by at most one either way
    else if (k <= k_cap && t + (float)k * b.Dq < tt) k++;
    return k;
}

template <int CTRL>
__device__ __forceinline__ unsigned dpp_u32(unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xF, 0xF, false);
}

// three floats moved by ONE 12-byte access (global_{load,store}_dwordx3 need dword alignment only).  The march kernels are bound by the number
// of accesses the vector L1 processes (TCP_TOTAL_ACCESSES, ~2.2 cycles each per CU however many waves are resident), not by bytes
struct __attribute__((packed, aligned(4))) Float3 {
    float x, y, z;
};
// the part of RayConsts that does not depend on the ray
__device__ __forceinline__ void frame_consts(const MarchParams& a, RayConsts& c) {
    const uint32_t H = a.H, C = a.C;
    c.rH = 1 / (float)H;
    c.H3 = (float)(H * H * H);
    c.halfH = 0.5f * (float)H;
    c.Hm1 = (float)(H - 1);
    c.Hf = (float)H;
    c.Cf = (float)C;
    c.dt_min = 2 * 1.7320508075688772f / a.max_steps;
    c.dt_max = 2 * 1.7320508075688772f * (1 << (C - 1)) / H;
    c.bmin0 = a.bbmin[0]; c.bmin1 = a.bbmin[1]; c.bmin2 = a.bbmin[2];
    c.bmax0 = a.bbmax[0]; c.bmax1 = a.bbmax[1]; c.bmax2 = a.bbmax[2];
    c.hi0 = (float)((double)c.bmax0 - 1e-6); c.hi1 = (float)((double)c.bmax1 - 1e-6); c.hi2 = (float)((double)c.bmax2 - 1e-6);
    c.r0 = a.resolution[0]; c.r1 = a.resolution[1]; c.r2 = a.resolution[2];
    c.rbound = 1 / a.bound;
}
__device__ __forceinline__ void ray_consts(const MarchParams& a, int index, RayConsts& c) {
    const Float3 o = *reinterpret_cast<const Float3*>(a.rays_o + (size_t)index * 3), d = *reinterpret_cast<const Float3*>(a.rays_d + (size_t)index * 3);
    c.ox = o.x; c.oy = o.y; c.oz = o.z;
    c.dx = d.x; c.dy = d.y; c.dz = d.z;
    c.rdx = 1 / c.dx; c.rdy = 1 / c.dy; c.rdz = 1 / c.dz;
    c.far = a.fars[index];
    frame_consts(a, c);
}

// Trip prologue of one ray (raymarching.cu:1163-1172).  Returns false when the ray has nothing to march.
// `resume` (may be null): t left by skip_empty_cells; `last_t` keeps the trip's start.
__device__ __forceinline__ bool ray_start(const MarchParams& a, const RayConsts& c, int index, float noise, const float* resume, RayState& st) {
    float t = a.rays_t[index];
    t += clampf(t * a.dt_gamma, c.dt_min, c.dt_max) * noise;
    st.last_t = t;
    st.step = 0;
    st.t = t;
    if (!(t < c.far)) return false;
    if (resume) st.t = *resume;
    return true;
}

// Where a ray can stop.  A sample is only ever emitted at a point whose search cell has candidates, so beyond the last such cell on its way
// a ray does nothing but hop from voxel to voxel until `far` — for a ray that misses the object but crosses its bounding box that is ALL it
// does (most of k_march_skip's work), and a ray that has left the object behind walks on through the tail pass (the longest tail rays).
// `bits2` marks the cells within one cell of a cell with candidates (k_frame_lists); the ray is sampled backwards from `far` every 0.9 cell
// lengths, and the march may end at the last sample before the first marked one: any point beyond it lies within one cell (per axis) of an
// unmarked sample, hence in a cell without candidates — with a whole cell to spare for the rounding of the reference's own cell arithmetic.
// Returns `near` when no sample is marked (nothing to march).  Not for --cut (static samples need no candidates).
__device__ inline float ray_end_of_candidates(const MarchParams& a, const uint32_t* bits2, float ox, float oy, float oz, float dx, float dy, float dz,
                                              float near, float far) {
    const float bmin0 = a.bbmin[0], bmin1 = a.bbmin[1], bmin2 = a.bbmin[2];
    const int r0 = a.resolution[0], r1 = a.resolution[1], r2 = a.resolution[2];
    const float rhgs = __builtin_amdgcn_rcpf(a.hgs);
    const float dts = 0.9f * a.hgs * __builtin_amdgcn_rsqf(dx * dx + dy * dy + dz * dz);
    if (!(dts > 0.0f) || !(far - near < 1e3f * dts)) return far;  // degenerate direction or an absurdly long ray: no shortcut
    float s_prev = far, s = far;
    while (true) {
        const int g0 = min(max((int)floorf((ox + s * dx - bmin0) * rhgs), 0), r0 - 1);
        const int g1 = min(max((int)floorf((oy + s * dy - bmin1) * rhgs), 0), r1 - 1);
        const int g2 = min(max((int)floorf((oz + s * dz - bmin2) * rhgs), 0), r2 - 1);
        const int gid = g2 * r1 * r0 + g1 * r0 + g0;
        if ((bits2[gid >> 5] >> (gid & 31)) & 1u) return s_prev;
        if (!(s > near)) return near;
        s_prev = s;
        s = fmaxf(s - dts, near);
    }
}

// floor((v - lo) / hgs) as the reference computes it (IEEE division), without the division in the common case: the quotient by reciprocal
// is within a few 1e-7 relative of the correctly rounded one, so the two can only floor differently when the quotient sits within 1e-3 of an
// integer — then, and for anything not finite, the real division decides.
__device__ __forceinline__ int cell_coord(float v, float lo, float hgs, float rhgs) {
    const float d = v - lo;
    const float q = d * rhgs;
    const float f = floorf(q);
    const float fr = q - f;
    if (!(fr >= 1e-3f && fr <= 0.999f && fabsf(q) < 1e6f)) return (int)floorf(d / hgs);
    return (int)f;
}

// ---- --cut frames: the leading run through EMPTY REGIONS of the static background --------------------------------------------------------
// With --cut every ray crosses the whole +-bound volume, hopping from density voxel to density voxel (raymarching.cu:1386-1431) until it meets an occupied
// voxel of the static background or a search cell with candidates inside the cut box: ~86 hops of ~230 instructions per ray on the trex option set, and the
// pre-pass is issue-bound on them (12 k waves; keeping the bitfield's block map in LDS so that a hop waits for no load moved nothing).  Most of that way leads
// through space in which NOTHING can happen, and all the caller needs is an element of the ray's t-sequence that the reference's chain certainly visits
// just before something can.  Two facts, as for the fixed-step DDA start of skip_empty_cells:
//   * the t-sequence is grid-independent: T_{k+1} = T_k + clamp(T_k dt_gamma, dt_min, dt_max), whatever the chain visits of it — walking it costs five
//     instructions per element where visiting one costs 230;
//   * a REGION is one 8^3-voxel block of the top cascade level (a 64-byte line of the bitfield; the caller may choose 4^3); its bit says "interesting": some voxel of it is occupied on
//     ANY level >= L that a point inside it can be tested on (one map per L: a ray asks the map of the lowest level its dt still allows), or it meets the cut box (inside which the search cells decide).  A point in a region without the
//     bit is a static-background sample in an empty voxel whatever its mip level: the chain emits nothing there.
// region_dda walks the regions the ray crosses until the next one is interesting or a crossing lies within 2e-3 region widths of another face (where the
// reference's float arithmetic could mean a lateral neighbour: the walk ends at the last crossing that is clear of that); the chain is then restarted at an element e whose predecessor p satisfies
//   p < exit(p) - margin,  e >= exit(p) + margin   (exit = the chain's own voxel-exit expression at p, on p's own mip level),
//   every element of the t-sequence within one top-level voxel diagonal before p is on p's mip level,
// which makes e a visited element whichever element of p's voxel the chain stands on: an element v <= p whose voxel (on v's level) contains p is on p's
// level by the second condition, so it is p's voxel, its exit is exit(p) to rounding, and the chain lands on the first element behind it: e.
#ifndef PN_REGION_RETRY_HOPS
#define PN_REGION_RETRY_HOPS 6   // exact hops behind a region look-ahead before the next one (measured alone on trex: 2 / 6 / 10 / 16 / 24 hops -> 201 / 175 / 178 / 185 / 205 us; hop by hop 285)
#endif
struct RegionGrid {
    const uint32_t* bits;  // LDS: bit (b2 R + b1) R + b0
    int R;                 // regions per axis = H / 8
    float lo, w, rw;       // -bound, region width 2 bound / R, its reciprocal
};
__device__ __forceinline__ bool region_set(const RegionGrid& g, int b0, int b1, int b2) {
    const int i = (b2 * g.R + b1) * g.R + b0;
    return ((g.bits[i >> 5] >> (i & 31)) & 1u) != 0;
}
// From parameter t (a point strictly inside the volume): the parameter t_stop of the last face crossed into a region without the bit before an interesting
// region, the volume's boundary or a doubt.  Returns 0: nothing crossed (t_stop = t); 1: t_stop set (axis m_stop); 2: no interesting region before `far`.
__device__ inline int region_dda(const RegionGrid& g, float ox, float oy, float oz, float dx, float dy, float dz, float rdx, float rdy, float rdz, float t,
                                 float far, float* t_stop_out, int* m_stop_out) {
    // the start may lie ON (or a padding in front of) the volume's boundary — a frame's rays start where they enter the box +-(bound + 1e-3) — and the chain clamps
    // its sample positions onto +-bound: a clamp along the face normal only.  So the start's coordinates are clamped into the grid, and a lateral neighbour
    // beyond the volume does not exist — nothing can be filed there — instead of blocking the look-ahead
    const float Rf = (float)g.R;
    const float q0 = fminf(fmaxf(((ox + t * dx) - g.lo) * g.rw, 0.0f), Rf), q1 = fminf(fmaxf(((oy + t * dy) - g.lo) * g.rw, 0.0f), Rf),
                q2 = fminf(fmaxf(((oz + t * dz) - g.lo) * g.rw, 0.0f), Rf);
    *t_stop_out = t;
    *m_stop_out = 0;
    if (!(q0 >= 0.0f && q1 >= 0.0f && q2 >= 0.0f)) return 0;   // NaN
    const float f0 = fminf(floorf(q0), Rf - 1.0f), f1 = fminf(floorf(q1), Rf - 1.0f), f2 = fminf(floorf(q2), Rf - 1.0f);
    int c0 = (int)f0, c1 = (int)f1, c2 = (int)f2;
    if (region_set(g, c0, c1, c2)) return 0;   // the common early exit: inside an interesting region
    const float e0 = q0 - f0, e1 = q1 - f1, e2 = q2 - f2;   // in [0, 1] (1: on the volume's far face)
    const int s_l0 = e0 < 2e-3f ? -1 : (e0 > 0.998f ? 1 : 0), s_l1 = e1 < 2e-3f ? -1 : (e1 > 0.998f ? 1 : 0), s_l2 = e2 < 2e-3f ? -1 : (e2 > 0.998f ? 1 : 0);
    if (s_l0 | s_l1 | s_l2) {   // a start next to a face: the regions it could be taken for
        for (int q = 1; q < 8; q++) {
            if (((q & 1) && !s_l0) || ((q & 2) && !s_l1) || ((q & 4) && !s_l2)) continue;
            const int y0 = c0 + ((q & 1) ? s_l0 : 0), y1 = c1 + ((q & 2) ? s_l1 : 0), y2 = c2 + ((q & 4) ? s_l2 : 0);
            if (y0 < 0 || y1 < 0 || y2 < 0 || y0 >= g.R || y1 >= g.R || y2 >= g.R) continue;   // beyond the volume
            if (region_set(g, y0, y1, y2)) return 0;
        }
    }
    const int st0 = dx > 0.0f ? 1 : -1, st1 = dy > 0.0f ? 1 : -1, st2 = dz > 0.0f ? 1 : -1;
    const bool use0 = fabsf(dx) > 1e-6f, use1 = fabsf(dy) > 1e-6f, use2 = fabsf(dz) > 1e-6f;
    float tf0 = use0 ? ((g.lo + (float)(c0 + (st0 > 0)) * g.w) - ox) * rdx : FLT_MAX;
    float tf1 = use1 ? ((g.lo + (float)(c1 + (st1 > 0)) * g.w) - oy) * rdy : FLT_MAX;
    float tf2 = use2 ? ((g.lo + (float)(c2 + (st2 > 0)) * g.w) - oz) * rdz : FLT_MAX;
    // where a crossing lies between the faces of another axis, from the face parameters alone: u = (tf_axis - tx) |d_axis| / w is the way left to that
    // axis's next face in region widths (0: on it, 1: on the one behind).  An axis the ray runs along keeps its start position between its faces.
    const float k0 = fabsf(dx) * g.rw, k1 = fabsf(dy) * g.rw, k2 = fabsf(dz) * g.rw;
    const float u_fix0 = st0 > 0 ? 1.0f - e0 : e0, u_fix1 = st1 > 0 ? 1.0f - e1 : e1, u_fix2 = st2 > 0 ? 1.0f - e2 : e2;
    int crossed = 0;
    float t_stop = t;
    int m_stop = 0;
    for (int it = 0; it < 3 * g.R + 3; it++) {
        const int m = (tf0 <= tf1 && tf0 <= tf2) ? 0 : (tf1 <= tf2 ? 1 : 2);
        const float tx = m == 0 ? tf0 : (m == 1 ? tf1 : tf2);
        if (!(tx < far)) { *t_stop_out = t_stop; *m_stop_out = m_stop; return 2; }
        const float u0 = use0 ? (tf0 - tx) * k0 : u_fix0, u1 = use1 ? (tf1 - tx) * k1 : u_fix1, u2 = use2 ? (tf2 - tx) * k2 : u_fix2;
        const bool ok0 = m == 0 || (u0 >= 2e-3f && u0 <= 0.998f), ok1 = m == 1 || (u1 >= 2e-3f && u1 <= 0.998f), ok2 = m == 2 || (u2 >= 2e-3f && u2 <= 0.998f);
        // a crossing within 2e-3 region widths of another face: the reference's float arithmetic may file the elements around it under the lateral neighbour
        // across that face — of the region being left or of the one being entered.  ONE such face: those two regions are looked at, and the walk goes on if
        // neither has the bit.  Two (an edge or a corner, or two faces at one parameter): the walk ends at the last crossing that was clear — between two clear
        // crossings the ray's distance to the lateral faces lies between its values at the two — and the exact hops take the ray past the doubtful one.
        // (The first version ran a 14-way neighbour loop here, which one lane in a hundred takes and its whole wave waits for: 26 of the first look-ahead's 62 us
        // on the trex option set; ending the walk at EVERY doubtful crossing instead cost more than that in the extra look-aheads of the rays that meet nothing.)
        const int n0 = c0 + (m == 0 ? st0 : 0), n1 = c1 + (m == 1 ? st1 : 0), n2 = c2 + (m == 2 ? st2 : 0);
        if (!(tx > t_stop)) break;
        if (!(ok0 && ok1 && ok2)) {
            const int l0 = (m == 0 || ok0) ? 0 : (u0 < 0.5f ? st0 : -st0), l1 = (m == 1 || ok1) ? 0 : (u1 < 0.5f ? st1 : -st1), l2 = (m == 2 || ok2) ? 0 : (u2 < 0.5f ? st2 : -st2);
            if ((l0 != 0) + (l1 != 0) + (l2 != 0) != 1) break;
            const int a0 = c0 + l0, a1 = c1 + l1, a2 = c2 + l2, b0 = n0 + l0, b1 = n1 + l1, b2 = n2 + l2;
            if (a0 < 0 || a1 < 0 || a2 < 0 || a0 >= g.R || a1 >= g.R || a2 >= g.R || b0 < 0 || b1 < 0 || b2 < 0 || b0 >= g.R || b1 >= g.R || b2 >= g.R) break;
            if (region_set(g, a0, a1, a2) || region_set(g, b0, b1, b2)) break;
        }
        t_stop = tx;
        m_stop = m;
        if (n0 < 0 || n1 < 0 || n2 < 0 || n0 >= g.R || n1 >= g.R || n2 >= g.R) break;   // the volume's boundary: points beyond are clamped onto it
        if (region_set(g, n0, n1, n2)) break;
        c0 = n0; c1 = n1; c2 = n2;
        if (m == 0) tf0 = ((g.lo + (float)(c0 + (st0 > 0)) * g.w) - ox) * rdx;
        else if (m == 1) tf1 = ((g.lo + (float)(c1 + (st1 > 0)) * g.w) - oy) * rdy;
        else tf2 = ((g.lo + (float)(c2 + (st2 > 0)) * g.w) - oz) * rdz;
        crossed++;
    }
    *t_stop_out = t_stop;
    *m_stop_out = m_stop;
    return crossed >= 1 ? 1 : 0;
}

// Leading run of marching iterations that can neither emit nor warp, ONE lane per ray: search cells whose 27-neighbourhood holds no IP
// and, in --cut mode, static-background points (outside the cut box) in empty density voxels.  In --cut mode every ray crosses the
// whole +-bound volume, so without this pre-pass all of those iterations went through the 8-lane / wave-per-ray kernels
// (trex option set at 1008x756: 65.7 M visited points per frame).  At 800x800 ~86 % of the first trip's iterations are of this kind
// (rays crossing the empty part of the IP bounding box): no candidate is found, the sample is not warped, the only memory touched is
// the cell's list range.  Returns the t at which the windowed march has to take over (first iteration whose cell has candidates, or
// t >= far); the arithmetic is eval_point's, expression by expression, so resuming there is bit-identical to having run every
// iteration in the windowed march.  The kernel's duration is its longest lane's chain (~130 hops), so the hop is kept short: the cell
// coordinates without divisions (cell_coord), one cascade without the mip functions, and with a fixed step the do-while that advances t
// past the voxel exit replaced by lattice arithmetic (Binade).
// `cell_bits` (may be null): bit c set <=> search cell c has candidates, held in LDS by the caller — the emptiness test then costs an LDS read
// instead of a dependent global round trip every third hop or so.
// `far_override` >= 0: the ray's end (see ray_end_of_candidates) instead of a.fars[index].
// `cell_bits2` (may be null; with cell_bits, no --cut, fixed step, one cascade, bound <= 1): DDA START.  A hop is ~230 dependent instructions
// (0.7 us for a lone wave) and the rays that graze the object do up to ~85 of them across cells in which nothing can happen: 65 of the pre-pass's
// 77 us were its longest such walk.  All the caller needs is the element at which the chain first meets a cell with candidates, and two facts make it
// computable without walking:
//   * the t-sequence is a lattice (Binade) whatever the chain visits, and WHICH elements it visits is local: from any visited element of a voxel V the
//     chain computes tt ~ the parameter at which the ray leaves V and lands on the first lattice element >= tt — so an element e is visited whenever
//     its predecessor p lies safely inside its voxel (p < tt_p - margin, tt_p = the chain's own exit expression evaluated at p) and e lies safely
//     behind that exit (e >= tt_p + margin): whichever element of p's voxel the chain is on, it computes an exit within the margin of tt_p and lands
//     on e (and if an ambiguity further back made it skip p's voxel, it landed on the first element behind p: e again);
//   * a cell DDA over the search grid knows which cells the ray crosses.  It goes on while the next cell has no candidates; where the crossing point
//     lies within 2e-3 cell lengths of another face (the reference's float cell arithmetic, errors ~1e-6 cell lengths, may put elements around it
//     into a lateral neighbour) it also looks at those lateral neighbours of both cells and stops if one of them has candidates or does not exist.
// The chain is restarted at the last certainly-visited element more than a margin before the stopping face and the exact hops take over; if the DDA
// reaches `far` the chain emits nothing.  No such element within 32 tries, another binade in between, a start next to a face AND to candidates: the
// hops start where they always did.  ONE attempt per ray, before the loop, so that the lanes of a wave stay together.
// `hop_budget` (> 0, with the DDA start): a wave is as slow as its slowest ray, and 0.04 % of the rays (face-hugging or axis-parallel ones: ~270 of
// 640 000, spread over a fifth of the busy waves) still walk 40-80 hops; after `hop_budget` hops the pre-pass hands the ray — standing on a visited
// element like any other resume point — to the windowed march, which evaluates 64 elements (~14 hops) per round and has 12 000 other rays to hide
// it behind.  Bit-identical by construction; the march tests compare every ray with the oracle and with the reference's own kernel.
__device__ inline float skip_empty_cells(const MarchParams& a, const March2Tables& tb, int index, float noise, unsigned* n_iter_out,
                                         const uint32_t* cell_bits = nullptr, float far_override = -1.0f, const uint32_t* cell_bits2 = nullptr,
                                         int hop_budget = 0, const uint32_t* grid_regions = nullptr, int regions_R = 0) {
    const Float3 o = *reinterpret_cast<const Float3*>(a.rays_o + (size_t)index * 3), d = *reinterpret_cast<const Float3*>(a.rays_d + (size_t)index * 3);
    const float ox = o.x, oy = o.y, oz = o.z, dx = d.x, dy = d.y, dz = d.z;
    const uint32_t H = a.H, C = a.C;
    const float far = far_override >= 0.0f ? far_override : a.fars[index];
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
    const float rhgs = __builtin_amdgcn_rcpf(a.hgs);
    const float sx = 0.5f + 0.5f * signf(dx), sy = 0.5f + 0.5f * signf(dy), sz = 0.5f + 0.5f * signf(dz);  // exact: 0, 0.5 or 1
    const bool one_cascade = C == 1;
    const bool fixed = a.dt_gamma == 0.0f;
    const float D = clampf(0.0f, dt_min, dt_max);
    Binade bn;
    bn.Dq = bn.rDq = 0.f; bn.top = -1.f; bn.ok = false;
    float bn_lo = 0.f, k_max = 0.f;
    int cell_id = -1;
    unsigned n_iter = 0;
    if (!cut && cell_bits) {
        // The common case (no --cut, emptiness bits in LDS) with as few branches as the semantics allow: the kernel is issue-bound (its busy
        // waves share a third of the SIMDs) and the general loop below costs ~25 exec-mask branches per hop.  Same expressions, same order.
        const float Hm1 = (float)(H - 1);
        if (cell_bits2 && fixed && one_cascade && a.bound <= 1.0f) {  // DDA start (see above); bound <= 1: mip_bound = fminf(2^0, bound) = bound
            const float xs0 = clampf(ox + t * dx, lo0, hi0), ys0 = clampf(oy + t * dy, lo1, hi1), zs0 = clampf(oz + t * dz, lo2, hi2);
            const float q0 = (xs0 - bmin0) * rhgs, q1 = (ys0 - bmin1) * rhgs, q2 = (zs0 - bmin2) * rhgs;
            const float f0 = floorf(q0), f1 = floorf(q1), f2 = floorf(q2);
            const float e0 = q0 - f0, e1 = q1 - f1, e2 = q2 - f2;
            const bool sure0 = e0 >= 2e-3f && e0 <= 0.998f && e1 >= 2e-3f && e1 <= 0.998f && e2 >= 2e-3f && e2 <= 0.998f &&
                               fmaxf(fabsf(q0), fmaxf(fabsf(q1), fabsf(q2))) < 1e6f;
            int c0 = (int)f0, c1 = (int)f1, c2 = (int)f2;
            const bool in0 = (c0 | c1 | c2) >= 0 && c0 < r0 && c1 < r1 && c2 < r2;
            bool start_blocked = !in0;  // the start cell has candidates, or the start lies next to a face and within one cell of candidates: the hops decide
            if (in0) {
                const int cg = c2 * r1 * r0 + c1 * r0 + c0;
                start_blocked = (((cell_bits[cg >> 5] >> (cg & 31)) & 1u) != 0) || (!sure0 && (((cell_bits2[cg >> 5] >> (cg & 31)) & 1u) != 0));
            }
            if (!start_blocked) {
                const int st0 = dx > 0.0f ? 1 : -1, st1 = dy > 0.0f ? 1 : -1, st2 = dz > 0.0f ? 1 : -1;
                const bool use0 = fabsf(dx) > 1e-6f, use1 = fabsf(dy) > 1e-6f, use2 = fabsf(dz) > 1e-6f;
                // parameter of the next face per axis, recomputed from the cell index (no accumulation) for the axis that moved only
                float tf0 = use0 ? ((bmin0 + (float)(c0 + (st0 > 0)) * a.hgs) - ox) * rdx : FLT_MAX;
                float tf1 = use1 ? ((bmin1 + (float)(c1 + (st1 > 0)) * a.hgs) - oy) * rdy : FLT_MAX;
                float tf2 = use2 ? ((bmin2 + (float)(c2 + (st2 > 0)) * a.hgs) - oz) * rdz : FLT_MAX;
                int crossed = 0, m_stop = 0;
                float t_stop = t;
                bool to_far = false;
                for (int it = 0; it < 96; it++) {
                    const int m = (tf0 <= tf1 && tf0 <= tf2) ? 0 : (tf1 <= tf2 ? 1 : 2);
                    const float tx = m == 0 ? tf0 : (m == 1 ? tf1 : tf2);
                    if (!(tx < far)) { to_far = true; break; }
                    const float fr0 = ((ox + tx * dx) - bmin0) * rhgs - (float)c0, fr1 = ((oy + tx * dy) - bmin1) * rhgs - (float)c1,
                                fr2 = ((oz + tx * dz) - bmin2) * rhgs - (float)c2;
                    const bool ok0 = m == 0 || (fr0 >= 2e-3f && fr0 <= 0.998f), ok1 = m == 1 || (fr1 >= 2e-3f && fr1 <= 0.998f),
                               ok2 = m == 2 || (fr2 >= 2e-3f && fr2 <= 0.998f);
                    const bool unsafe = !(ok0 && ok1 && ok2) || !(tx > t_stop);  // next to another face, or an edge / corner (two faces at one parameter)
                    const int n0 = c0 + (m == 0 ? st0 : 0), n1 = c1 + (m == 1 ? st1 : 0), n2 = c2 + (m == 2 ? st2 : 0);
                    if (tx > t_stop) { t_stop = tx; m_stop = m; }
                    if (n0 < 0 || n1 < 0 || n2 < 0 || n0 >= r0 || n1 >= r1 || n2 >= r2) break;
                    const int ng = n2 * r1 * r0 + n1 * r0 + n0;
                    if ((cell_bits[ng >> 5] >> (ng & 31)) & 1u) break;
                    if (unsafe) {
                        // the lateral neighbours the rounding could mean: one step towards every face the crossing point is close to, for the cell
                        // being left and the cell being entered; every one of them must exist and be free of candidates
                        const int l0 = (m == 0 || ok0) ? 0 : (fr0 < 0.5f ? -1 : 1), l1 = (m == 1 || ok1) ? 0 : (fr1 < 0.5f ? -1 : 1),
                                  l2 = (m == 2 || ok2) ? 0 : (fr2 < 0.5f ? -1 : 1);
                        bool blocked = !(tx > t_stop - 1e-6f * fmaxf(1.0f, fabsf(tx)));  // a face clearly BEHIND the last one: not a rounding matter
                        for (int q = 1; q < 8 && !blocked; q++) {  // the non-empty subsets of the (at most two) lateral steps
                            const int s0 = (q & 1) ? l0 : 0, s1 = (q & 2) ? l1 : 0, s2 = (q & 4) ? l2 : 0;
                            if ((s0 | s1 | s2) == 0 || ((q & 1) && !l0) || ((q & 2) && !l1) || ((q & 4) && !l2)) continue;
                            for (int side = 0; side < 2; side++) {
                                const int y0 = (side ? n0 : c0) + s0, y1 = (side ? n1 : c1) + s1, y2 = (side ? n2 : c2) + s2;
                                if (y0 < 0 || y1 < 0 || y2 < 0 || y0 >= r0 || y1 >= r1 || y2 >= r2) { blocked = true; break; }
                                const int yg = y2 * r1 * r0 + y1 * r0 + y0;
                                if ((cell_bits[yg >> 5] >> (yg & 31)) & 1u) { blocked = true; break; }
                            }
                        }
                        if (blocked) break;
                    }
                    c0 = n0; c1 = n1; c2 = n2;
                    if (m == 0) tf0 = ((bmin0 + (float)(c0 + (st0 > 0)) * a.hgs) - ox) * rdx;
                    else if (m == 1) tf1 = ((bmin1 + (float)(c1 + (st1 > 0)) * a.hgs) - oy) * rdy;
                    else tf2 = ((bmin2 + (float)(c2 + (st2 > 0)) * a.hgs) - oz) * rdz;
                    crossed++;
                }
                if (to_far) { *n_iter_out = 0; return far; }  // no candidates and no doubt all the way: the chain walks to `far` and emits nothing
                bn = binade_of<1>(t, D);
                const float kcap = bn.ok ? floorf(16777215.0f / (bn.Dq * scalbnf(1.0f, 150 - (int)(__float_as_uint(t) >> 23)))) - 2.0f : 0.0f;
                // margins: ~60 ulp of t + the error of (face - x) * (1 / d) for the axis in question (an axis the ray is nearly parallel to has a huge
                // 1 / d, but its face is never the nearest one)
                const float rd_stop = fabsf(m_stop == 0 ? rdx : (m_stop == 1 ? rdy : rdz));
                const float m_face = 2e-4f * fmaxf(1.0f, fabsf(t)) + 1e-5f * fminf(rd_stop, 1e3f);  // elements this far before the face are in the crossed cells
                const float target = t_stop - m_face;
                if (crossed >= 1 && bn.ok && target > t && target < bn.top && rd_stop < 1e3f) {
                    float kf = fminf(floorf((target - t) * bn.rDq), kcap);
                    if (kf >= 1.0f && t + kf * bn.Dq > target) kf -= 1.0f;
                    for (int tries = 0; tries < 32 && kf >= 2.0f; tries++, kf -= 1.0f) {
                        const float e = t + kf * bn.Dq, p = t + (kf - 1.0f) * bn.Dq;  // exact: both inside the binade, k below the exactness cap
                        const float x = clampf(ox + p * dx, lo0, hi0), y = clampf(oy + p * dy, lo1, hi1), z = clampf(oz + p * dz, lo2, hi2);
                        const int nx = (int)clampf((x * rbound + 1) * halfH, 0.0f, Hm1);
                        const int ny = (int)clampf((y * rbound + 1) * halfH, 0.0f, Hm1);
                        const int nz = (int)clampf((z * rbound + 1) * halfH, 0.0f, Hm1);
                        const float ux = ((((float)nx + sx) * rH * 2 - 1) * a.bound - x) * rdx;
                        const float uy = ((((float)ny + sy) * rH * 2 - 1) * a.bound - y) * rdy;
                        const float uz = ((((float)nz + sz) * rH * 2 - 1) * a.bound - z) * rdz;
                        const float um = fminf(ux, fminf(uy, uz));
                        const float rd_min = fabsf(um == ux ? rdx : (um == uy ? rdy : rdz));
                        const float m_vox = 3e-5f * fmaxf(1.0f, fabsf(t)) + 2e-6f * rd_min;
                        const float ttp = p + fmaxf(0.0f, um);
                        if (rd_min < 1e3f && p < ttp - m_vox && e >= ttp + m_vox && e < far) { t = e; break; }
                    }
                }
                bn.Dq = bn.rDq = 0.f; bn.top = -1.f; bn.ok = false;  // the hop loop below sets its own
            }
        }
        while (true) {
            const float x = clampf(ox + t * dx, lo0, hi0);
            const float y = clampf(oy + t * dy, lo1, hi1);
            const float z = clampf(oz + t * dz, lo2, hi2);
            const float d0 = x - bmin0, d1 = y - bmin1, d2 = z - bmin2;
            const float q0 = d0 * rhgs, q1 = d1 * rhgs, q2 = d2 * rhgs;
            float f0 = floorf(q0), f1 = floorf(q1), f2 = floorf(q2);
            const float e0 = q0 - f0, e1 = q1 - f1, e2 = q2 - f2;
            const bool sure = e0 >= 1e-3f && e0 <= 0.999f && e1 >= 1e-3f && e1 <= 0.999f && e2 >= 1e-3f && e2 <= 0.999f &&
                              fmaxf(fabsf(q0), fmaxf(fabsf(q1), fabsf(q2))) < 1e6f;
            if (!sure) { f0 = floorf(d0 / a.hgs); f1 = floorf(d1 / a.hgs); f2 = floorf(d2 / a.hgs); }  // see cell_coord
            const int g0 = (int)f0, g1 = (int)f1, g2 = (int)f2;
            const bool inside = (g0 | g1 | g2) >= 0 && g0 < r0 && g1 < r1 && g2 < r2;
            const int gid = inside ? g2 * r1 * r0 + g1 * r0 + g0 : 0;
            const bool has = ((cell_bits[gid >> 5] >> (gid & 31)) & 1u) != 0;
            if (!inside || has) break;  // outside the hash (the windowed march raises the error flag) or candidates: hand over
            if (hop_budget > 0 && (int)n_iter >= hop_budget) break;  // a straggler: the windowed march goes on from this (visited) element
            int level = 0;
            if (!one_cascade) level = max(mip_from_pos(x, y, z, (float)C), mip_from_dt(clampf(t * a.dt_gamma, dt_min, dt_max), (float)H, (float)C));
            const float pw = scalbnf(1.0f, level);
            const bool use_pw = pw <= a.bound;
            const float mip_bound = use_pw ? pw : a.bound;
            const float mip_rbound = use_pw ? scalbnf(1.0f, -level) : rbound;
            const int nx = (int)clampf((x * mip_rbound + 1) * halfH, 0.0f, Hm1);
            const int ny = (int)clampf((y * mip_rbound + 1) * halfH, 0.0f, Hm1);
            const int nz = (int)clampf((z * mip_rbound + 1) * halfH, 0.0f, Hm1);
            n_iter++;
            const float tx = ((((float)nx + sx) * rH * 2 - 1) * mip_bound - x) * rdx;
            const float ty = ((((float)ny + sy) * rH * 2 - 1) * mip_bound - y) * rdy;
            const float tz = ((((float)nz + sz) * rH * 2 - 1) * mip_bound - z) * rdz;
            const float tt = t + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
            bool stepped = false;
            if (fixed) {
                if (!(t >= bn_lo && t < bn.top)) {  // entered another binade
                    bn = binade_of<1>(t, D);
                    bn_lo = bn.top * 0.5f;
                    k_max = bn.ok ? floorf(16777215.0f / (bn.Dq * scalbnf(1.0f, 150 - (int)(__float_as_uint(t) >> 23)))) - 2.0f : 0.0f;
                }
                float kf = fminf(fmaxf(ceilf((tt - t) * bn.rDq), 1.0f), k_max);
                const bool dec = kf > 1.0f && t + (kf - 1.0f) * bn.Dq >= tt;
                const bool inc = !dec && t + kf * bn.Dq < tt;
                kf = dec ? kf - 1.0f : (inc ? kf + 1.0f : kf);
                const float tn = t + kf * bn.Dq;
                stepped = bn.ok && kf <= k_max && tn < bn.top && tn >= tt;  // everything inside the binade: exact
                if (stepped) t = tn;
            }
            if (!stepped)
                do { t += clampf(t * a.dt_gamma, dt_min, dt_max); } while (t < tt);
            if (!(t < far)) break;
        }
        *n_iter_out = n_iter;
        return t;
    }
    // --cut with the region map (see region_dda): skip the empty regions in front of the first interesting one.  Levels: bound == 2^(C - 1) (the caller checked),
    // so level l's voxels are 2^(l + 1) / H wide and a region is 8^3 voxels of level C - 1.
    const bool regions_on = cut && grid_regions != nullptr && regions_R > 0;
    RegionGrid rg{grid_regions, regions_R, -a.bound, 2.0f * a.bound / (float)max(regions_R, 1), (float)max(regions_R, 1) / (2.0f * a.bound)};
    const float rnorm = __builtin_amdgcn_rsqf(dx * dx + dy * dy + dz * dz);
    const float w_t = 1.7320508f * 2.0f * a.bound * rH * rnorm * 1.02f;   // one top-level voxel diagonal in units of t (+ 2 %)
    int hops_since_try = 0, retry_hops = PN_REGION_RETRY_HOPS;
    bool try_regions = regions_on && w_t > 0.0f && w_t < 1e3f;
#if PN_DBG_SKIP_HOPS  // timing experiment: at most this many hops per ray (results invalid)
    while (t < far && n_iter < PN_DBG_SKIP_HOPS) {
#else
    while (t < far) {
#endif
        if (try_regions) {
            try_regions = false;
            hops_since_try = 0;
            const float px = ox + t * dx, py = oy + t * dy, pz = oz + t * dz;
            // inside the volume, or just in front of it: a frame's rays start on the box +-(bound + 1e-3) (renderer.py:782-791 pads the box), and the chain clamps its
            // sample positions onto +-bound — which region_dda's clamped start coordinates reproduce.  (Rounds of the first version began with a look-ahead that
            // this test refused for every ray: the real first one came six hops later.)
            const float b_in = a.bound + 2e-3f;
            if (fabsf(px) <= b_in && fabsf(py) <= b_in && fabsf(pz) <= b_in) {
                float t_stop;
                int m_stop;
                // the map of the ray's minimum mip level from here on: level = max(from the position, from dt) >= mip_from_dt(dt(t)), and dt does not fall along the ray —
                // on the trex option set dt puts every point on level 1, and what is occupied on level 0 only cannot be met
                const int lv_min = one_cascade ? 0 : mip_from_dt(clampf(t * a.dt_gamma, dt_min, dt_max), (float)H, (float)C);
                rg.bits = grid_regions + (size_t)lv_min * (size_t)((rg.R * rg.R * rg.R) >> 5);
                const int kind = region_dda(rg, ox, oy, oz, dx, dy, dz, rdx, rdy, rdz, t, far, &t_stop, &m_stop);
                if (kind == 2) { *n_iter_out = n_iter; return far; }   // nothing interesting before `far`: the chain walks there and emits nothing
                const float rd_stop = fabsf(m_stop == 0 ? rdx : (m_stop == 1 ? rdy : rdz));
                const float m_face = 2e-4f * fmaxf(1.0f, fabsf(t_stop)) + 1e-5f * fminf(rd_stop, 1e3f);
                const float target = t_stop - m_face;
                // the zone in front of `target` in which the restart pair is looked for: a voxel diagonal of same-level elements behind p, then p and e
                const float zone_len = 1.1f * w_t + 3.0f * clampf(target * a.dt_gamma, dt_min, dt_max);
                if (kind == 1 && rd_stop < 1e3f && target - zone_len - w_t > t) {
                    // walk the t-sequence (the chain's own step expression) to the zone [target - zone_len, target), then look for the LAST pair (p, e) of consecutive
                    // elements in it that restarts the chain (conditions above)
                    float T = t;
                    const float zone = target - zone_len;
                    while (true) {   // four elements per exit test (the walk is a large part of a long look-ahead: up to ~150 elements)
                        const float T1 = T + clampf(T * a.dt_gamma, dt_min, dt_max);
                        const float T2 = T1 + clampf(T1 * a.dt_gamma, dt_min, dt_max);
                        const float T3 = T2 + clampf(T2 * a.dt_gamma, dt_min, dt_max);
                        const float T4 = T3 + clampf(T3 * a.dt_gamma, dt_min, dt_max);
                        if (T4 < zone) { T = T4; continue; }
                        T = T3 < zone ? T3 : (T2 < zone ? T2 : (T1 < zone ? T1 : T));   // the last element in front of the zone
                        break;
                    }
                    float best = -1.0f, run_start = FLT_MAX;
                    int run_level = -1;
                    for (int it = 0; it < 512; it++) {   // (w_t / dt_min elements at most: a few dozen)
                        const float pT = T, eT = T + clampf(T * a.dt_gamma, dt_min, dt_max);
                        if (!(eT < target)) break;
                        const float x = clampf(ox + pT * dx, lo0, hi0), y = clampf(oy + pT * dy, lo1, hi1), z = clampf(oz + pT * dz, lo2, hi2);
                        const float dtp = clampf(pT * a.dt_gamma, dt_min, dt_max);
                        const int level = one_cascade ? 0 : max(mip_from_pos(x, y, z, (float)C), mip_from_dt(dtp, (float)H, (float)C));
                        if (level != run_level) { run_level = level; run_start = pT; }
                        if (run_start <= pT - w_t) {
                            const float pw = scalbnf(1.0f, level);
                            const bool use_pw = pw <= a.bound;
                            const float mip_bound = use_pw ? pw : a.bound;
                            const float mip_rbound = use_pw ? scalbnf(1.0f, -level) : rbound;
                            const int nx = (int)clampf((x * mip_rbound + 1) * halfH, 0.0f, (float)(H - 1));
                            const int ny = (int)clampf((y * mip_rbound + 1) * halfH, 0.0f, (float)(H - 1));
                            const int nz = (int)clampf((z * mip_rbound + 1) * halfH, 0.0f, (float)(H - 1));
                            const float ux = ((((float)nx + sx) * rH * 2 - 1) * mip_bound - x) * rdx;
                            const float uy = ((((float)ny + sy) * rH * 2 - 1) * mip_bound - y) * rdy;
                            const float uz = ((((float)nz + sz) * rH * 2 - 1) * mip_bound - z) * rdz;
                            const float um = fminf(ux, fminf(uy, uz));
                            const float rd_min = fabsf(um == ux ? rdx : (um == uy ? rdy : rdz));
                            const float m_vox = 3e-5f * fmaxf(1.0f, fabsf(pT)) + 2e-6f * rd_min;
                            const float ttp = pT + fmaxf(0.0f, um);
                            if (rd_min < 1e3f && pT < ttp - m_vox && eT >= ttp + m_vox && eT < far) best = eT;
                        }
                        T = eT;
                    }
                    if (best > t) t = best;   // a visited element, in a region without the bit: the exact hops go on from here
                    else retry_hops = min(retry_hops * 4, 64);   // (no pair: a mip level change or a grazing ray — do not pay for the same look-ahead every few hops)
                }
            }
        }
        const float x = clampf(ox + t * dx, lo0, hi0);
        const float y = clampf(oy + t * dy, lo1, hi1);
        const float z = clampf(oz + t * dz, lo2, hi2);
        // --cut: a point outside the cut box is a static-background sample (found = true, un-warped, :1380-1383); the cut test is the
        // reference's, `x < cut_bounds[3]` included (:1210)
        const bool searched = !cut || (x > cb[0] && x < cb[1] && y > cb[2] && x < cb[3] && z > cb[4] && z < cb[5]);
        if (searched) {
            const int g0 = cell_coord(x, bmin0, a.hgs, rhgs);
            const int g1 = cell_coord(y, bmin1, a.hgs, rhgs);
            const int g2 = cell_coord(z, bmin2, a.hgs, rhgs);
            if (g0 < 0 || g1 < 0 || g2 < 0 || g0 >= r0 || g1 >= r1 || g2 >= r2) break;  // the windowed march raises the error flag
            const int gid = g2 * r1 * r0 + g1 * r0 + g0;
            if (gid != cell_id) {
                const bool has = cell_bits ? ((cell_bits[gid >> 5] >> (gid & 31)) & 1u) != 0 : tb.nb_rng[gid].x != tb.nb_rng[gid].y;
                if (has) break;  // candidates: hand over
                cell_id = gid;
            }
        }
        // found == false (or a static sample in an empty voxel): un-warped voxel skip (raymarching.cu:1386-1428)
        const float dt = clampf(t * a.dt_gamma, dt_min, dt_max);
        const int level = one_cascade ? 0 : max(mip_from_pos(x, y, z, (float)C), mip_from_dt(dt, (float)H, (float)C));
        const float pw = scalbnf(1.0f, level);
        const bool use_pw = pw <= a.bound;
        const float mip_bound = use_pw ? pw : a.bound;
        const float mip_rbound = use_pw ? scalbnf(1.0f, -level) : rbound;
        // (float)(0.5 * (double)v * (double)H) == v * (0.5f * H): both round the exact product once (v has 24 significant bits, H < 2^24)
        const int nx = (int)clampf((x * mip_rbound + 1) * halfH, 0.0f, (float)(H - 1));
        const int ny = (int)clampf((y * mip_rbound + 1) * halfH, 0.0f, (float)(H - 1));
        const int nz = (int)clampf((z * mip_rbound + 1) * halfH, 0.0f, (float)(H - 1));
        if (!searched) {  // static sample: emitted when its voxel is occupied -> hand over to the windowed march at this element
            const uint32_t vox = (uint32_t)(level * H3 + (float)morton3D(nx, ny, nz));
            if (a.grid[vox / 8] & (1 << (vox % 8))) break;
        }
        n_iter++;
        if (regions_on && ++hops_since_try >= retry_hops) try_regions = true;   // past an interesting region without an event: look ahead again
        // (n + 0.5f + 0.5f * sign) == n + (0.5f + 0.5f * sign): n is an integer below 2^23 and the addend 0, 0.5 or 1 — both sums are exact
        const float tx = ((((float)nx + sx) * rH * 2 - 1) * mip_bound - x) * rdx;
        const float ty = ((((float)ny + sy) * rH * 2 - 1) * mip_bound - y) * rdy;
        const float tz = ((((float)nz + sz) * rH * 2 - 1) * mip_bound - z) * rdz;
        const float tt = t + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
        // do { t += dt(t); } while (t < tt);
        bool stepped = false;
        if (fixed) {
            if (!(t >= bn_lo && t < bn.top)) {  // entered another binade
                bn = binade_of<1>(t, D);
                bn_lo = bn.top * 0.5f;
                k_max = bn.ok ? floorf(16777215.0f / (bn.Dq * scalbnf(1.0f, 150 - (int)(__float_as_uint(t) >> 23)))) - 2.0f : 0.0f;
            }
            if (bn.ok) {
                // at least one step; k = first k >= 1 with t + k * Dq >= tt (quotient off by one at most, corrected by exact comparisons)
                float kf = fminf(fmaxf(ceilf((tt - t) * bn.rDq), 1.0f), k_max);
                if (kf > 1.0f && t + (kf - 1.0f) * bn.Dq >= tt) kf -= 1.0f;
                else if (t + kf * bn.Dq < tt) kf += 1.0f;
                const float tn = t + kf * bn.Dq;
                if (kf <= k_max && tn < bn.top && tn >= tt) { t = tn; stepped = true; }  // everything inside the binade: exact
            }
        }
        if (!stepped)
            do { t += clampf(t * a.dt_gamma, dt_min, dt_max); } while (t < tt);
    }
    *n_iter_out = n_iter;
    return t;
}

// Timing experiment (-DPN_DBG_PHASES=1 builds only, tools/build_variant.py): per-wave clocks of the phases of a march kernel, summed into
// MarchParams::stats[4 + base ...].  Every phase boundary drains the memory counters, so the phases add up to the wave's lifetime.
#if PN_DBG_PHASES
#define PN_DBG_PHASES_ON 1
struct PhaseClock {
    unsigned long long t0, acc[6];
};
__device__ __forceinline__ unsigned long long dbg_clock() {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    return __builtin_readcyclecounter();
}
#define PN_PHASE_DECL(pk) pnm3::PhaseClock pk; { for (int _i = 0; _i < 6; _i++) pk.acc[_i] = 0; pk.t0 = pnm3::dbg_clock(); }
#define PN_PHASE(pk, i) do { const unsigned long long _n = pnm3::dbg_clock(); (pk).acc[i] += _n - (pk).t0; (pk).t0 = _n; } while (0)
#define PN_PHASE_FLUSH(pk, stats, base, lane) do { if ((stats) && (lane) == 0) for (int _i = 0; _i < 6; _i++) atomicAdd((stats) + 4 + (base) + _i, (pk).acc[_i]); } while (0)
#define PN_PHASE_ARG , pnm3::PhaseClock& pk
#define PN_PHASE_PASS , pk
#else
#define PN_DBG_PHASES_ON 0
#define PN_PHASE_DECL(pk)
#define PN_PHASE(pk, i)
#define PN_PHASE_FLUSH(pk, stats, base, lane)
#define PN_PHASE_ARG
#define PN_PHASE_PASS
#endif

struct PointEval {
    float x, y, z;  // sample position (warped when IPs were found)
    float dt;       // step at this t
    float tt;       // voxel exit target when the point is not emitted
    bool emit;      // occupied && found
    bool oob;       // search cell outside the spatial hash (error flag of the reference's printf branch)
    unsigned n_cand, n_warp;
};

// First half of one marching iteration at ray parameter t (raymarching.cu:1190-1232): the sample position and the candidate range of
// its search cell.  Split from the rest so that the lanes of a wave can stage the lists they are about to scan together.
struct PointCell {
    float x, y, z;
    bool in_cut, oob;
    int gid, b, e;  // candidates of the search cell: nb[b .. e)  (b == e: none / not searched)
};
__device__ __forceinline__ void point_cell(const MarchParams& a, const March2Tables& tb, const RayConsts& c, float t, PointCell& p) {
    if (a.cut) {
        p.x = clampf(c.ox + t * c.dx, -a.bound, a.bound);
        p.y = clampf(c.oy + t * c.dy, -a.bound, a.bound);
        p.z = clampf(c.oz + t * c.dz, -a.bound, a.bound);
    } else {
        p.x = clampf(c.ox + t * c.dx, c.bmin0, c.hi0);
        p.y = clampf(c.oy + t * c.dy, c.bmin1, c.hi1);
        p.z = clampf(c.oz + t * c.dz, c.bmin2, c.hi2);
    }
    p.in_cut = true;
    if (a.cut) {
        const float* cb = a.cut_bounds;  // `x < cb[3]` is the reference's own test (raymarching.cu:1210)
        p.in_cut = (p.x > cb[0] && p.x < cb[1] && p.y > cb[2] && p.x < cb[3] && p.z > cb[4] && p.z < cb[5]);
    }
    p.oob = false;
    p.gid = -1;
    p.b = p.e = 0;
    if (p.in_cut) {
        // the three cell coordinates as in cell_coord, with ONE (rare) branch for all of them
        const float rhgs = __builtin_amdgcn_rcpf(a.hgs);
        const float d0 = p.x - c.bmin0, d1 = p.y - c.bmin1, d2 = p.z - c.bmin2;
        const float q0 = d0 * rhgs, q1 = d1 * rhgs, q2 = d2 * rhgs;
        float f0 = floorf(q0), f1 = floorf(q1), f2 = floorf(q2);
        const float e0 = q0 - f0, e1 = q1 - f1, e2 = q2 - f2;
        const bool sure = e0 >= 1e-3f && e0 <= 0.999f && e1 >= 1e-3f && e1 <= 0.999f && e2 >= 1e-3f && e2 <= 0.999f &&
                          fmaxf(fabsf(q0), fmaxf(fabsf(q1), fabsf(q2))) < 1e6f;
        if (!sure) { f0 = floorf(d0 / a.hgs); f1 = floorf(d1 / a.hgs); f2 = floorf(d2 / a.hgs); }
        const int g0 = (int)f0, g1 = (int)f1, g2 = (int)f2;
        p.oob = (g0 < 0 || g1 < 0 || g2 < 0 || g0 >= c.r0 || g1 >= c.r1 || g2 >= c.r2);
        if (!p.oob) {
            p.gid = g2 * c.r1 * c.r0 + g1 * c.r0 + g0;
            const int2 rng = tb.nb_rng[p.gid];
            p.b = rng.x;
            p.e = rng.y;
        }
    }
}

// Wave-cooperative staging of candidate lists in LDS.  The lanes of a wave scan a handful of DISTINCT lists per round (the 8 lanes of a
// ray sit in the same search cell or two; in the tail pass 64 points of one ray cross four or five cells), but every lane used to fetch
// its list itself: ~30 of the ~58 vector-memory instructions of a wave-round, each a 64-lane x 16 B gather.  Measured (rocprofv3
// VmemLatency): 2 400 cycles per vector-memory instruction with the chip full of march waves against 560 when nearly empty — the vector
// L1 path was the bottleneck, not HBM and not the ALUs.  Here each distinct list is copied once, by all 64 lanes with coalesced 16-byte
// LDS-DMA loads (global_load_lds_dwordx4: no staging registers, one wait for all lists), and the per-lane scans read LDS.
// Must be called with all 64 lanes active (destination = wave-uniform base + lane * 16).  Returns this lane's list offset in `stage`
// (entries), or -1 when the wave's lists did not all fit: those lanes scan from global memory as before.
#ifndef PN_HEAD_SPLIT
#define PN_HEAD_SPLIT 0
#endif
#ifndef PN_STAGE_CAP
#define PN_STAGE_CAP 768  // entries per wave: 12 KB (the record heads of a round need 4 * 3 * 64); 4 waves per workgroup, 3 workgroups per CU -> 144 of the 160 KB
#endif
__device__ __forceinline__ void glds16(const float4* g, float4* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
template <int CAP = PN_STAGE_CAP>
__device__ __forceinline__ int stage_lists(float4* stage, const float4* __restrict__ nb, int b, int len, int lane) {
    int my_off = -1;
    int cursor = 0;
    unsigned long long todo = __ballot(len > 0);
    while (todo) {  // one iteration per distinct list (lists are per cell: same start <=> same list)
        const int leader = __builtin_ctzll(todo);
        const int lb = __builtin_amdgcn_readlane(b, leader), ll = __builtin_amdgcn_readlane(len, leader);
        const bool mine = len > 0 && b == lb;
        todo &= ~__ballot(mine);
        if (cursor + ll > CAP) continue;
        if (mine) my_off = cursor;
        for (int i0 = 0; i0 < ll; i0 += 64)
            if (i0 + lane < ll) glds16(nb + lb + i0 + lane, stage + cursor + i0);
        cursor += ll;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return my_off;
}

// Candidate search of one marching iteration, one lane (raymarching.cu:1232-1244 -> find_closest_IP(s)): the K nearest IPs of the point,
// nearest first, -1 = none.  `stage` / `my_off`: this lane's candidate list in LDS (stage_lists), my_off < 0: read it from tb.nb.
template <int K>
__device__ __forceinline__ void eval_scan(const MarchParams& a, const March2Tables& tb, const PointCell& pc, const float4* stage, int my_off, int (&ips)[3],
                                          unsigned& n_cand) {
    const float x = pc.x, y = pc.y, z = pc.z;
    ips[0] = ips[1] = ips[2] = -1;
    n_cand = 0;
    if (pc.in_cut && !pc.oob) {
        {
            const int n_list = pc.e - pc.b;
            n_cand = (unsigned)n_list;
            if (K == 1) {
                // find_closest_IP (:986-1043): own cell first, the 26 neighbours only if that found nothing; `d < best` from 9999.9
                const int own = a.pig_cnt[pc.gid];
                float best = (float)9999.9;
                auto scan1 = [&](const float4* base) {
                    for (int j = 0; j < own; j++) {
                        const float4 v = base[j];
                        const float ax = v.x - x, ay = v.y - y, az = v.z - z;
                        const float d = ax * ax + ay * ay + az * az;
                        if (d < best) { best = d; ips[0] = __float_as_int(v.w); }
                    }
                    if (ips[0] == -1) {
                        for (int j = own; j < n_list; j++) {
                            const float4 v = base[j];
                            const float ax = v.x - x, ay = v.y - y, az = v.z - z;
                            const float d = ax * ax + ay * ay + az * az;
                            if (d < best) { best = d; ips[0] = __float_as_int(v.w); }
                        }
                    }
                };
                if (my_off >= 0) scan1(stage + my_off); else scan1(tb.nb + pc.b);
            } else {
                // find_closest_IPs (:1045-1118): all 27 cells in visiting order (= list order), insertion on strict '<'.  Written as selects
                // on the three comparisons (d0 <= d1 <= d2 always, so d < d0 implies d < d1 implies d < d2 and the reference's
                // if / else-if chain is exactly this): the compiler's branchy form of the chain cost ~55 instructions and four
                // exec-mask branches per candidate, this one 21 and none.  The IP id travels with the distance, so the selected
                // entries need no second fetch.
                // Round 3: the three running minima are 64-bit KEYS (distance bits << 32 | position in the list), compared as doubles: for
                // non-negative floats the bit pattern orders like the value, the position breaks ties towards the earlier candidate — exactly the
                // sequential strict-'<' insertion — and any such pattern is a finite positive double (a float's exponent field never fills the
                // double's), so v_min_f64 / v_max_f64 order the keys: five instructions per candidate instead of thirteen selects (a wave pays for
                // its longest list, ~60 entries in the dense cells: the scan is more than half of pass 1's VALU instructions).  The ids of the
                // winners are read back from the list afterwards.  Sentinel = (FLT_MAX bits, 0): a candidate at distance FLT_MAX, inf or NaN
                // never enters, as `d < FLT_MAX` never holds for it.
                const double sent = __hiloint2double(0x7F7FFFFF, 0);
                double m0 = sent, m1 = sent, m2 = sent;
                auto kmin = [](double a_, double b_) { double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a_), "v"(b_)); return r; };
                auto kmax = [](double a_, double b_) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a_), "v"(b_)); return r; };
                auto insert = [&](const float4& v, int pos, bool valid) {
                    const float ax = v.x - x, ay = v.y - y, az = v.z - z;
                    const float d = ax * ax + ay * ay + az * az;
                    const double k = valid ? __hiloint2double(__float_as_int(d), pos) : sent;
                    const double t0 = kmax(m0, k);
                    m0 = kmin(m0, k);
                    if (K > 2) {
                        const double t1 = kmax(m1, t0);
                        m1 = kmin(m1, t0);
                        m2 = kmin(m2, t1);
                    } else {
                        m1 = kmin(m1, t0);
                    }
                };
                auto scan = [&](const float4* base) {  // entries base[0 .. n_list)
                    int j0 = 0;
                    for (; j0 + PN_CAND_FLIGHT <= n_list; j0 += PN_CAND_FLIGHT) {  // entries in flight per round trip
                        float4 v[PN_CAND_FLIGHT];
#pragma unroll
                        for (int u = 0; u < PN_CAND_FLIGHT; u++) v[u] = base[j0 + u];
#pragma unroll
                        for (int u = 0; u < PN_CAND_FLIGHT; u++) insert(v[u], j0 + u, true);
                    }
                    if (j0 < n_list) {
                        float4 v[PN_CAND_FLIGHT - 1];
#pragma unroll
                        for (int u = 0; u < PN_CAND_FLIGHT - 1; u++) v[u] = base[min(j0 + u, n_list - 1)];
#pragma unroll
                        for (int u = 0; u < PN_CAND_FLIGHT - 1; u++) insert(v[u], j0 + u, j0 + u < n_list);
                    }
                    // a key still at the sentinel's distance bits found nothing (FLT_MAX itself is never inserted)
                    const int h0 = __double2hiint(m0), h1 = __double2hiint(m1), h2 = __double2hiint(m2);
                    if (h0 != 0x7F7FFFFF) ips[0] = __float_as_int(base[__double2loint(m0)].w);
                    if (h1 != 0x7F7FFFFF) ips[1] = __float_as_int(base[__double2loint(m1)].w);
                    if (K > 2 && h2 != 0x7F7FFFFF) ips[2] = __float_as_int(base[__double2loint(m2)].w);
                };
                if (my_off >= 0) scan(stage + my_off); else scan(tb.nb + pc.b);
            }
        }
    }
}

// The selected IPs' record heads (p_ori, p_def, F^-1: 64 B each) for all 64 lanes at once.  Per lane they are three scattered 64-byte
// records; fetched lane by lane (3 x 4 global_load_dwordx4) every instruction costs the vector L1 one access per LANE (~770 per wave-round,
// a third of all the kernel's accesses, and the accesses are what bounds it).  Here the four lanes of a quad fetch each other's records:
// instruction (k, j) has every lane load part (lane & 3) of the k-th record of lane j of its quad — one contiguous 64 B per quad, 16
// accesses per instruction — by LDS-DMA into stage[(k * 4 + j) * 64 + lane]; lane (Q, j) then reads its four parts back from
// stage[(k * 4 + j) * 64 + Q * 4 ...].  All 64 lanes must be active; lanes without an IP fetch record 0.  Needs 4 * K * 64 entries of `stage`
// (the candidate lists staged there have been consumed by then).
template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }
template <int K, int SPLIT = PN_HEAD_SPLIT>
__device__ __forceinline__ void head_fetch(float4* stage, const float4* __restrict__ rec, const int (&ips)[3], int lane, float4 (&rh)[3][4]) {
    const int q = lane & 3;
    // SPLIT = 1: the first two records, then the third, through the same 8 KB (one more round trip per round, a third less LDS per wave)
    constexpr int K0 = (SPLIT && K > 2) ? 2 : K;
#pragma unroll
    for (int pass = 0; pass < ((K0 < K) ? 2 : 1); pass++) {
        const int kb = pass ? K0 : 0, ke = pass ? K : K0;
#pragma unroll
        for (int k = kb; k < ke; k++) {
            const int ipk = ips[k] >= 0 ? ips[k] : (ips[0] >= 0 ? ips[0] : 0);
            const int i0 = dpp_i32<0x00>(ipk), i1 = dpp_i32<0x55>(ipk), i2 = dpp_i32<0xAA>(ipk), i3 = dpp_i32<0xFF>(ipk);  // quad_perm broadcasts
            glds16(rec + (size_t)i0 * PN_REC_VEC4 + q, stage + ((k - kb) * 4 + 0) * 64);
            glds16(rec + (size_t)i1 * PN_REC_VEC4 + q, stage + ((k - kb) * 4 + 1) * 64);
            glds16(rec + (size_t)i2 * PN_REC_VEC4 + q, stage + ((k - kb) * 4 + 2) * 64);
            glds16(rec + (size_t)i3 * PN_REC_VEC4 + q, stage + ((k - kb) * 4 + 3) * 64);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int k = kb; k < ke; k++) {
            const float4* h = stage + ((k - kb) * 4 + q) * 64 + (lane & ~3);
#pragma unroll
            for (int u = 0; u < 4; u++) rh[k][u] = h[u];
        }
        if (K0 < K) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the reads above complete before the region is overwritten
    }
}

// Rest of one marching iteration, one lane: pre-filter, inverse warps, blend, density bit, voxel exit (raymarching.cu:1246-1431).
template <int K, bool MULTI>
__device__ __forceinline__ void eval_point(const MarchParams& a, const March2Tables& tb, const RayConsts& c, float t, const PointCell& pc,
                                           const int (&ips)[3], const float4 (&rh)[3][4], unsigned n_cand, PointEval& r) {
    bool found = false;
    float x = pc.x, y = pc.y, z = pc.z;
    r.oob = pc.oob;
    r.n_cand = n_cand;
    r.n_warp = 0;
    if (pc.in_cut) {
        float x_map = 0.0f, y_map = 0.0f, z_map = 0.0f;
        int n_IP = (ips[0] != -1) + (ips[1] != -1) + (ips[2] != -1);
        found = n_IP > 0;
        if (found) {
            // the selected IPs' record heads (p_ori, p_def, F^-1: 64 B each) arrive in rh[][] (head_fetch below) — the pre-filter loop, whose
            // bound shrinks as it runs, and the first Newton step work on registers
            // pre-filter (:1246-1251) on the candidates' deformed positions: `n_IP--` inside the loop it bounds, strict '<' on z only
#pragma unroll
            for (int k = 0; k < K; k++) {
                if (k < n_IP) {
                    const float cx = rh[k][0].w, cy = rh[k][1].x, cz = rh[k][1].y;  // p_def
                    if (cx <= c.bmin0 || cy <= c.bmin1 || cz < c.bmin2 || cx >= c.bmax0 || cy >= c.bmax1 || cz >= c.bmax2) n_IP--;
                }
            }
            if (n_IP <= 0) found = false;
            if (found) {
                float ps[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, dk[3] = {0, 0, 0};
                // per-IP Newton warp (:1262-1324): a rejected result still fills its slot and shrinks the bound
#pragma unroll
                for (int k = 0; k < K; k++) {
                    if (k < n_IP) {
                        float pw[3];
                        r.n_warp++;
                        const float4* __restrict__ rp = tb.rec + (size_t)ips[k] * PN_REC_VEC4;
                        if (warp_record<MULTI>(rh[k], rp, a.max_iter_num, a.IP_dx, x, y, z, pw, &dk[k])) n_IP--;
                        ps[3 * k] = pw[0]; ps[3 * k + 1] = pw[1]; ps[3 * k + 2] = pw[2];
                    }
                }
                if (n_IP == 1) {
                    x_map = ps[0]; y_map = ps[1]; z_map = ps[2];
                } else if (n_IP == 2) {
                    const float dist_sum = dk[0] + dk[1];
                    const float w0 = dk[1] / dist_sum, w1 = dk[0] / dist_sum;
                    x_map = w0 * ps[0] + w1 * ps[3];
                    y_map = w0 * ps[1] + w1 * ps[4];
                    z_map = w0 * ps[2] + w1 * ps[5];
                } else if (n_IP == 3) {
                    const float dist_sum = dk[0] * dk[1] + dk[1] * dk[2] + dk[2] * dk[0];
                    const float w0 = dk[1] * dk[2] / dist_sum;
                    const float w1 = dk[0] * dk[2] / dist_sum;
                    const float w2 = dk[0] * dk[1] / dist_sum;
                    x_map = w0 * ps[0] + w1 * ps[3] + w2 * ps[6];
                    y_map = w0 * ps[1] + w1 * ps[4] + w2 * ps[7];
                    z_map = w0 * ps[2] + w1 * ps[5] + w2 * ps[8];
                }
                x = x_map; y = y_map; z = z_map;  // n_IP == 0 here maps the sample to the origin (:1372-1374)
            }
        }
    } else {
        found = true;  // cut mode, outside the cut box: un-warped background sample (:1380-1383)
    }

    const float dt = dtf(a, c, t);
    // one cascade: both mip functions clamp to [0, C - 1] = 0
    const int level = (c.Cf == 1.0f) ? 0 : max(mip_from_pos(x, y, z, c.Cf), mip_from_dt(dt, c.Hf, c.Cf));
    const float pw2 = scalbnf(1.0f, level);  // mip_bound = fminf(2^level, bound); 1 / 2^level is exact
    const bool use_pw = pw2 <= a.bound;
    const float mip_bound = use_pw ? pw2 : a.bound;
    const float mip_rbound = use_pw ? scalbnf(1.0f, -level) : c.rbound;
    // (float)(0.5 * (double)v * (double)H) == v * (0.5f * H): both round the exact product once (pn_march_tables.h)
    const int nx = (int)clampf((x * mip_rbound + 1) * c.halfH, 0.0f, c.Hm1);
    const int ny = (int)clampf((y * mip_rbound + 1) * c.halfH, 0.0f, c.Hm1);
    const int nz = (int)clampf((z * mip_rbound + 1) * c.halfH, 0.0f, c.Hm1);
    bool occ = false;
    if (found) {  // the occupancy bit only matters when an IP was found (`occ && found`)
        const uint32_t vox = (uint32_t)(level * c.H3 + (float)morton3D(nx, ny, nz));
        occ = (bool)(a.grid[vox / 8] & (1 << (vox % 8)));
    }
    r.emit = occ && found;
    r.x = x; r.y = y; r.z = z;
    r.dt = dt;
    const float tx = (((nx + 0.5f + 0.5f * signf(c.dx)) * c.rH * 2 - 1) * mip_bound - x) * c.rdx;
    const float ty = (((ny + 0.5f + 0.5f * signf(c.dy)) * c.rH * 2 - 1) * mip_bound - y) * c.rdy;
    const float tz = (((nz + 0.5f + 0.5f * signf(c.dz)) * c.rH * 2 - 1) * mip_bound - z) * c.rdz;
    r.tt = t + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
}

// Up to `max_rounds` windows of G sequence elements of one ray; the G lanes [gbase, gbase + G) run in lock step.
// Returns true when the ray is done for this trip (t >= far, or n_step samples), false when the round budget ran out; `st` is
// advanced either way (identically on all G lanes).  Samples go to xyzs/dirs/deltas[slot], slot = number emitted before.
// Called by ALL 64 lanes of the wave (have_ray = false for the lanes of a group without a ray): the round loop is wave-uniform so that
// the candidate lists of a round can be staged cooperatively (stage_lists) in `stage`, the wave's CAP entries of LDS (CAP >= 512; below 768 the three
// record heads of a lane go through it in two passes: SPLIT).
template <int K, bool MULTI, int G, int CAP = PN_STAGE_CAP, int SPLIT = PN_HEAD_SPLIT>
__device__ inline bool march_window(const MarchParams& a, const March2Tables& tb, const RayConsts& c, uint32_t n_step, int sub, int gbase, int lane,
                                    float4* stage, float* __restrict__ xyzs, float* __restrict__ dirs, float* __restrict__ deltas, RayState& st,
                                    int max_rounds, bool have_ray PN_PHASE_ARG) {
    float t = st.t, last_t = st.last_t;
    uint32_t step = st.step;
    const float far = c.far;
    unsigned st_iter = 0, st_cand = 0, st_warp = 0;  // instrumentation (visited points only), reported when a.stats != nullptr
    bool done = true;
    bool running = have_ray;
    int rounds = 0;
    const bool fixed = a.dt_gamma == 0.0f;
    const float D = clampf(0.0f, c.dt_min, c.dt_max);  // dtf() of any finite t when dt_gamma == 0
    while (true) {
        bool go = running && t < far && step < n_step;  // the same on the G lanes of a ray
        if (go && rounds == max_rounds) { done = false; go = false; }
        running = go;
        if (!__any(go)) break;
        rounds++;
        // this lane's point s_sub and its successor
        Binade bn;
        bn.Dq = bn.rDq = bn.top = 0.f; bn.ok = false;
        bool fast = false;  // the whole window s_0 .. s_G lies inside t's binade: lattice arithmetic
        float s = 0.f, nxt = 0.f;
        bool active = false;
        PointCell pc;
        pc.x = pc.y = pc.z = 0.f; pc.in_cut = false; pc.oob = false; pc.gid = -1; pc.b = pc.e = 0;
        if (go) {
            if (fixed) {
                // (G == 1, one lane per ray: the lattice reaches 64 elements ahead, so that a voxel hop is one product instead of a stepping loop)
                bn = binade_of<(G == 1 ? 64 : G)>(t, D);
                fast = bn.ok && t + (float)(G == 1 ? 65 : G) * bn.Dq < bn.top;
            }
            if (fast) {
                s = t + (float)sub * bn.Dq;
                nxt = t + (float)(sub + 1) * bn.Dq;
            } else {
                s = t;
                for (int j = 0; j < G - 1; j++)
                    if (j < sub) s += dtf(a, c, s);
                nxt = s + dtf(a, c, s);
            }
            active = s < far;
            if (active) point_cell(a, tb, c, s, pc);
        }
        PN_PHASE(pk, 1);
        // a window none of whose points has a candidate (rays that have left the object and walk on to `far`, grazing rays between two parts of
        // it) needs no lists, no scan and no record heads: what is left of the round is the voxel arithmetic and the chain
        const bool any_list = __any(pc.e != pc.b);
        int my_off = -1;
#ifndef PN_G1_NOSTAGE
#define PN_G1_NOSTAGE 0   // 1: the one-lane form scans its candidate lists from global memory instead of staging the wave's distinct lists in LDS (A/B build)
#endif
        if (any_list && !(G == 1 && PN_G1_NOSTAGE)) my_off = stage_lists<CAP>(stage, tb.nb, pc.b, pc.e - pc.b, lane);
        PN_PHASE(pk, 2);
        PointEval ev;
        ev.emit = false; ev.oob = false; ev.tt = 0.f; ev.dt = 0.f; ev.x = ev.y = ev.z = 0.f; ev.n_cand = 0; ev.n_warp = 0;
        int ips[3] = {-1, -1, -1};
        unsigned n_cand = 0;
        float4 rh[3][4];
#pragma unroll
        for (int k = 0; k < 3; k++)
#pragma unroll
            for (int u = 0; u < 4; u++) rh[k][u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (any_list) {
            if (go && active) eval_scan<K>(a, tb, pc, stage, my_off, ips, n_cand);
            head_fetch<K, SPLIT>(stage, tb.rec, ips, lane, rh);
        }
        if (go && active) eval_point<K, MULTI>(a, tb, c, s, pc, ips, rh, n_cand, ev);
        PN_PHASE(pk, 3);
        if (go) {
            // where the chain goes from this point: the next element (emitted) or the first element not below the voxel exit;
            // G = first element of the next window, G + 1 = beyond it
            int jump = sub + 1;
            float t_hop = nxt;  // G == 1: where the chain goes from this (visited) point
            if (G == 1 && active && !ev.emit) {
                if (fast) {
                    const int k = lattice_first_at_least(bn, t, ev.tt, 1, 64);
                    t_hop = t + (float)min(k, 64) * bn.Dq;       // exact: inside the binade
                    if (k > 64) while (t_hop < ev.tt) t_hop += dtf(a, c, t_hop);
                } else {
                    while (t_hop < ev.tt) t_hop += dtf(a, c, t_hop);  // do { t += dt } while (t < tt): the first step is `nxt`
                }
            } else if (active && !ev.emit) {
                if (fast) {
                    jump = lattice_first_at_least(bn, t, ev.tt, sub + 1, G);
                } else {
                    float u = nxt;
                    int k = sub + 1;
                    while (k < G && u < ev.tt) { u += dtf(a, c, u); k++; }
                    jump = (k == G && u < ev.tt) ? G + 1 : k;
                }
            }
            unsigned word = 0;
            if (G == 8) {
                word = (unsigned)jump << (4 * sub);
                word |= dpp_u32<0xB1>(word);   // lane ^ 1
                word |= dpp_u32<0x4E>(word);   // lane ^ 2
                word |= dpp_u32<0x141>(word);  // lane <-> 7 - lane within each 8
            }
            const unsigned long long gmask = (G == 64) ? ~0ull : ((1ull << G) - 1ull);
            const unsigned long long emitm = (__ballot(ev.emit) >> gbase) & gmask;
            const unsigned long long actm = (__ballot(active) >> gbase) & gmask;
            // replay of the visit chain, identically on the G lanes
            int cur = 0, prev_emit = -1, n_emit = 0, last_vis = 0;
            int my_ord = -1, my_prev = -1;
            bool visited = false, ended = false;
            if (G == 1) {
                // one ray per lane: its single point is the visited one
                if (!active) ended = true;
                else {
                    visited = true; my_ord = 0; my_prev = -1; last_vis = 0;
                    if (ev.emit) { prev_emit = 0; n_emit = 1; if (step + 1u == n_step) ended = true; }
                }
            } else if (G == 64) {
                // one ray per wave: the chain is wave-uniform — walked with scalar registers and v_readlane instead of 64 lanes each
                // following it through ds_bpermute (~14 dependent LDS round trips per round)
                unsigned long long vis = 0ull;
                const int need = (int)(n_step - step);
                while (cur < 64) {
                    if (!((actm >> cur) & 1ull)) { ended = true; break; }
                    vis |= 1ull << cur;
                    last_vis = cur;
                    if ((emitm >> cur) & 1ull) {
                        prev_emit = cur;
                        n_emit++;
                        if (n_emit == need) { ended = true; break; }
                    }
                    cur = __builtin_amdgcn_readlane(jump, cur);
                }
                const unsigned long long ev_m = vis & emitm, below = (1ull << sub) - 1ull;
                visited = ((vis >> sub) & 1ull) != 0;
                my_ord = (int)__popcll(ev_m & below);
                my_prev = (ev_m & below) ? 63 - (int)__clzll(ev_m & below) : -1;
            } else {
            while (cur < G) {
                if (!((actm >> cur) & 1ull)) { ended = true; break; }  // s_cur >= far: the march is over
                if (cur == sub) { visited = true; my_ord = n_emit; my_prev = prev_emit; }
                last_vis = cur;
                const bool em = (emitm >> cur) & 1ull;
                const int nx_idx = (int)((word >> (4 * cur)) & 0xFu);
                if (em) {
                    prev_emit = cur;
                    n_emit++;
                    if (step + (uint32_t)n_emit == n_step) { ended = true; break; }
                }
                cur = nx_idx;
            }
            }
            // emitted samples: slot = step + rank in the chain; deltas[1] = t_after - last_t (raymarching.cu:1395-1410)
            const float prev_nxt = __shfl(nxt, gbase + max(my_prev, 0));
            if (visited) {
                st_iter++; st_cand += ev.n_cand; st_warp += ev.n_warp;
                if (ev.oob && a.err_flag) atomicOr(a.err_flag, 1);
                if (ev.emit) {
                    const uint32_t slot = step + (uint32_t)my_ord;
                    float* X = xyzs + (size_t)slot * 3;
                    float* Dd = dirs + (size_t)slot * 3;
                    float* L = deltas + (size_t)slot * 2;
                    *reinterpret_cast<Float3*>(X) = Float3{ev.x, ev.y, ev.z};
                    if (dirs) *reinterpret_cast<Float3*>(Dd) = Float3{c.dx, c.dy, c.dz};   // (nullptr: a caller that keeps the rays' directions itself)
                    *reinterpret_cast<float2*>(L) = make_float2(ev.dt, nxt - (my_prev >= 0 ? prev_nxt : last_t));
                }
            }
            // (G == 64: wave-uniform lane indices — v_readlane instead of a ds_bpermute round trip)
            const float last_emit_nxt = (G == 64) ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(nxt), max(prev_emit, 0))) : __shfl(nxt, gbase + max(prev_emit, 0));
            if (n_emit > 0) last_t = last_emit_nxt;
            step += (uint32_t)n_emit;
            // next window start: s_G, or further when the last visited point's voxel exit lies beyond the window
            const float sG = (G == 64) ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(nxt), 63)) : __shfl(nxt, gbase + G - 1);
            const float tt_last = (G == 64) ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ev.tt), last_vis)) : __shfl(ev.tt, gbase + last_vis);
            if (ended) {
                running = false;  // done for this trip; t is not needed any more (composite tracks rays_t itself)
            } else if (G == 1) {
                t = t_hop;
            } else {
                t = sG;
                if (cur == G + 1)
                    while (t < tt_last) t += dtf(a, c, t);
            }
        }
        PN_PHASE(pk, 4);
    }
#if !PN_DBG_PHASES  // (the phase-clock build reuses the counter block and must not be disturbed by these atomics)
    if (a.stats) {
        if (st_iter) { atomicAdd(a.stats, (unsigned long long)st_iter); atomicAdd(a.stats + 1, (unsigned long long)st_cand); }
        if (st_warp) atomicAdd(a.stats + 2, (unsigned long long)st_warp);
    }
#endif
    st.t = t;
    st.last_t = last_t;
    st.step = step;
    return done;
}

}  // namespace pnm3
