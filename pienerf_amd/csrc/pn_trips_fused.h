// Every loop trip after the first ones of a deformed frame as ONE launch (gfx950).  Included by pn_render_ops.hip behind composite_one / pn_frame.
//
// The reference's render loop (nerf/renderer.py:836-891) is, per trip: march n_step samples for every alive ray -> network on all samples -> composite
// -> rays_alive = rays_alive[rays_alive >= 0], with n_step = max(min(N // n_alive, 8), 1).  Rounds 1-3 ran that as 4-6 launches per trip (march pass,
// wave-per-ray tail pass, network, composite + compaction), 5-7 trips per frame: each launch fills the GPU for a fraction of its duration (a trip's tail
// pass: 35 us for a few hundred rays), every trip pays the latency of its slowest ray three times over, and ~45 launch boundaries sit on a frame's chain.
//
// What couples the rays of a trip is ONLY n_step — and once n_alive <= N / 8 it is 8 for the rest of the frame, because n_alive never grows.  From that
// trip on every ray can run its own loop { march 8 samples; network; composite } until it dies or reaches max_steps, with no global step in between and no
// compaction at all: exactly the reference's arithmetic per ray (the march restarts from the rays_t the composite accumulated, eight samples at a time,
// as in the per-trip launches), so samples, pixels and per-trip counts are those of the trip-by-trip loop bit for bit (tests: every frame test compares
// the two forms through the oracle / the reference's own kernels; test_gpu_fused.py compares them directly).
//
// k_trips_fused: persistent workgroups of PN_FUSED_WAVES waves, one per CU.  A wave holds 8 rays, 8 lanes each (pn_march_window.h: march_window<K, MULTI, 8>);
// per round it
//   1. refills the groups whose ray has died from its workgroup's share of the trip's alive list (packets of 8 dealt round-robin; an LDS cursor),
//   2. marches 8 samples per ray into the wave's own 64 sample slots — one window round; a ray still going after it (1 % of them: it grazes the object or
//      has left it) is walked on by the whole wave in 64-element windows (march_window<K, MULTI, 64>: what k_march_tail does in the per-trip form),
//   3. runs the network on its 64 slots (two 32-sample tiles of pn_net_tile.h: the LDS weight image is shared by the workgroup),
//   4. composites its 8 rays (composite_one, one lane per ray) and keeps the survivors for the next round.
// Waves in the march phase (chains of dependent VALU instructions) and waves in the network phase (gathers + MFMA) share a SIMD: what three render
// lanes did for each other in the pipelined harness happens inside one launch.  Per-trip bookkeeping (rays entering each trip, samples emitted, rays that
// needed the wave-per-ray windows) is counted in LDS, flushed once per workgroup, and the LAST workgroup to finish turns the counts into the trip
// records the per-trip launches would have written (PnTrip) — frame statistics, trip-record tests and pn_render_continue see no difference.
//
// Round 4: the frame's FIRST trip in the same launch too (template flag WHOLE, see the kernel) — a deformed frame is then prologue, skip pre-pass, this
// launch, epilogue.
//
// Precondition, checked on the device: the record of the first fused trip says n_step == 8 (and dense).  If not (a scene with more than N / 8 rays alive
// after the classic trips) the kernel does nothing and the frame is left unfinished exactly like a captured render that ran out of trips: the caller
// continues it (pn_render_continue / the blocking driver's own loop, which runs one more classic trip and tries again).
#pragma once
#define PN_TU_FP_CONTRACT_OFF 1  // this header lives in pn_render_ops.hip (-ffp-contract=off); pn_net_tile.h contracts inside itself and switches back
#include <cstddef>

#include "pn_net_tile.h"

#define PN_FUSED_MAX_TRIPS 128  // fused trips per frame: (max_steps - 1) / 8 for max_steps <= 1024
// (-DPN_DBG_PHASES=1 timing builds give march_window a phase clock: the fused launch has clocks of its own and hands it a dummy)
#if PN_DBG_PHASES
#define PN_FUSED_PK , pk_dummy_
#else
#define PN_FUSED_PK
#endif
#ifndef PN_FUSED_WROUNDS
#define PN_FUSED_WROUNDS 1   // 8-lane window rounds a later-trip ray gets before the whole wave walks it on in 64-element windows (2 / 3: measured, round 6)
#endif
#ifndef PN_FUSED_WAVES
#define PN_FUSED_WAVES 12       // waves per workgroup, one workgroup per CU = 3 waves per SIMD: 61 KB weight image + 8 KB of march staging per wave
#endif
#ifndef PN_FUSED_LDS_OUT
#define PN_FUSED_LDS_OUT 1       // the later trips' network outputs go to the composite through the wave's LDS staging area (free between two march rounds)
#endif
#ifndef PN_FUSED_LDS_XYZ
#define PN_FUSED_LDS_XYZ 1       // ... and the later trips' sample positions reach the network through LDS as well (768 B per wave; directions from the rays' registers):
#endif                           // no store -> wait -> load round trip through global memory between a round's march and its network tiles (not with the 61-KB bf16 image)
#define PN_FUSED_STAGE 512      // staging entries per wave (the record heads of a round go through it in two passes: pn_march_window.h, SPLIT)
// control block of a frame's fused launch, ints: [3 x PN_FUSED_MAX_TRIPS counters][workgroups done]; all zero at launch
#define PN_FUSED_CTL_HIST 0
#define PN_FUSED_CTL_DONE (PN_FUSED_CTL_HIST + 3 * PN_FUSED_MAX_TRIPS)
#define PN_FUSED_CTL_INTS (PN_FUSED_CTL_DONE + 32)

struct FusedArgs {
    // network (pn_net)
    const PnFusedLevel* lv;   // the 16 level records of the network form that runs: PnFusedLevel (fp16) or PnByteLevel (fp32 forms) — the same size
    const float* emb;
    const uint32_t* emb_h;
    uint32_t emb_bytes;
    const uint4* wimg_g;  // the LDS weight image in global memory (wsplit / whalf / wx)
    float net_bound, net_inv2b, density_scale;   // net_inv2b = 1.0f / (2 * net_bound) (pn_net_tile.h: tile_sigma_net)
    const float* x_scales;    // fp16 hi/lo form: device words {xs[0], 1 / xs[2]} beside the weight image (pn_common.h: pn_net::x_scales) — read by the kernel, so a
                              // captured launch follows an in-place weight refresh (pn_net_update)
    // frame
    PnTrip* trips;  // record of the first fused trip
    uint32_t N_rays, max_steps;
    float T_thresh;
    const int* alive;  // that trip's alive list
    float *rays_t, *weights_sum, *depth, *image;
    float *xyzs, *dirs, *deltas, *sigmas, *rgbs;  // sample slots: 64 per wave of the launch
    int* ctl;
    PnFrameDev* dev;
    int* tail_diag;              // per-trip diagnostics (rays that needed the 64-lane windows), aligned with `trips`
    unsigned long long* clocks;  // optional [16]: shader-clock cycles per phase summed over waves (refill, march, windows, network, composite), wave-rounds, waves;
                                 // [10..14] whole-frame form: first-trip march, its 64-lane windows, its network, its composite + hand-over, wait at the barrier
    // whole-frame form (WHOLE): the frame's FIRST trip inside the launch as well
    const int* active;           // [PN_SEGS x active_seg_cap] alive slots k_march_skip left something to march for
    const int* active_counts;    // their segmented counters
    int active_seg_cap;
    const float* t_resume;       // [N] per alive slot: where k_march_skip left the ray
    int* blist;                  // [2 x blist_cap] ray ids of the workgroups' shares of the first trip / of the rays that outlive it, one region per workgroup
    int4* strag;                 // [blist_cap] the first trip's rays still searching after the one-lane rounds
    uint32_t blist_cap;          // positions; the sample arrays hold gridDim * PN_FUSED_WAVES * 64 + blist_cap slots
    int a_rounds;                // one-lane rounds of the first trip before a ray goes to the 64-lane windows
    // the frame's epilogue inside the launch (QUEUED modes, `finalize` != 0): a ray's pixel is final when the ray leaves the launch, written there as k_frame_finish
    // would (renderer.py:896-899); the rays that never had a sample got theirs from the frame prologue
    int finalize; float bg; const float* nears; const float* fars_full; float* image_out; float* depth_out;
    // FOLD (MODE 2): the first trip's segmented sample list and the march's tail counters
    const int* list_seg; const int* samp_counts; int list_seg_cap; const int* seg_tail; const int* seg_back;
};

// The launch's own copy of `fa` in the kernarg segment, re-read WHERE it is used.  The kernel's uniform state (three argument structs: ~45 pointers and
// as many scalars) is twice what a wave has scalar registers for; what does not fit is parked in lanes of a vector register and every use costs a
// v_readlane — a VECTOR instruction, on CUs that are bound by vector issue: 132 of the 366 vector instructions of the composite step were such reloads of
// its ten array pointers.  A scalar load from the kernarg segment costs none; the opaque zero keeps the compiler from treating the address as loop-invariant
// (it would hoist the loads out of the round loop and park them again).  The offset is the AMDGPU kernarg layout of (MarchParams, March2Tables, FusedArgs):
// every argument at its natural alignment; the kernel checks it once per launch against the by-value copy (PN_ERR flag 16).
#ifndef PN_FUSED_FRESH_ARGS
#define PN_FUSED_FRESH_ARGS 1
#endif
__device__ __forceinline__ constexpr size_t pn_align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
template <typename T, int WHICH>   // WHICH: 0 = MarchParams, 1 = March2Tables, 2 = FusedArgs
__device__ __forceinline__ const T& fused_karg_fresh(const T& by_value) {
#if PN_FUSED_FRESH_ARGS
    // where the three by-value arguments lie in the kernarg segment: in order, each at its natural alignment (the HSA kernarg rule) — which is also how a struct of
    // the three is laid out, so the offsets are tied to offsetof() of that struct at compile time (round-5 advisor).  Passing that struct as the ONE argument
    // instead was built and measured (round 6): the same pipeline rate, but the launch alone 0.602 -> 0.624 ms in three alternating runs — the three-argument
    // form stays, and block 0 still compares the re-read fields with the by-value copies (error bit 32: a flag of its own).
    struct Layout { pnm::MarchParams a; pnm2::March2Tables tb; FusedArgs fa; };
    constexpr size_t off_tb = pn_align_up(sizeof(pnm::MarchParams), alignof(pnm2::March2Tables));
    constexpr size_t off_fa = pn_align_up(off_tb + sizeof(pnm2::March2Tables), alignof(FusedArgs));
    static_assert(off_tb == offsetof(Layout, tb) && off_fa == offsetof(Layout, fa), "kernarg placement of (MarchParams, March2Tables, FusedArgs)");
    constexpr size_t off = WHICH == 0 ? 0 : (WHICH == 1 ? off_tb : off_fa);
    uint32_t z;
    asm volatile("s_mov_b32 %0, 0" : "=s"(z));
    const char __attribute__((address_space(4)))* kp = (const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr();
    return *reinterpret_cast<const T*>((const char*)kp + off + z);
#else
    return by_value;
#endif
}
__device__ __forceinline__ const FusedArgs& fused_args_fresh(const FusedArgs& by_value) { return fused_karg_fresh<FusedArgs, 2>(by_value); }

// composite_one (kernel_composite_rays, raymarching.cu:827-923) for the 8 slots of one ray of the fused launch: the same operations in the same order,
// with the eight samples' sigma / rgb / deltas requested up front (one memory round trip instead of one per sample: a lone lane waiting for each
// was 12 000 cycles of every wave-round) and the ray's t carried in a register (`t` in: rays_t; out: t behind the last composited sample).
// `srgb` != nullptr (PN_FUSED_LDS_OUT, the later trips' rounds): the 8 samples' (sigma, r, g, b) come from the wave's own LDS area, where its network tiles
// have just left them — not through global memory and back (four stores per sample, a wait for them, 32 loads per composite lane: the composite phase was 10 k
// of a wave-round's 80 k cycles, most of it that round trip).
template <bool LDS_IN>
__device__ __forceinline__ bool composite_slots8(int index, uint32_t slot0, float T_thresh, float& t, float* rays_t, const float* __restrict__ sigmas,
                                                 const float* __restrict__ rgbs, const float* __restrict__ deltas, float* weights_sum, float* depth,
                                                 float* image, const float4* srgb) {
    float sg[8], d0[8], d1[8], cr[8], cg[8], cb[8];
    // ONE base address per array and constant offsets from it: with `slot0 + k` formed in 32 bits the compiler cannot rule out a wrap once slot0 has an
    // unknown addend (FOLD: N_rays), builds 24 separate 64-bit addresses, hoists them out of the round loop, spills them and reloads them every round
    // (the later trips' composite: 7 -> 30 k cycles per round)
    const float* __restrict__ sgp = sigmas + (size_t)slot0;
    const float* __restrict__ dlp = deltas + (size_t)slot0 * 2;
    const float* __restrict__ cp = rgbs + (size_t)slot0 * 3;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const float2 dd = *reinterpret_cast<const float2*>(dlp + k * 2);
        d0[k] = dd.x; d1[k] = dd.y;
        if (LDS_IN) {
            const float4 v = srgb[k];
            sg[k] = v.x; cr[k] = v.y; cg[k] = v.z; cb[k] = v.w;
        } else {
            sg[k] = sgp[k];
            const pnm3::Float3 c3 = *reinterpret_cast<const pnm3::Float3*>(cp + k * 3);
            cr[k] = c3.x; cg[k] = c3.y; cb[k] = c3.z;
        }
    }
    float ws = weights_sum[index], d = depth[index];
    float r = image[index * 3], g = image[index * 3 + 1], b = image[index * 3 + 2];
    bool go = true;
    int step = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        if (go) {
            if (d0[k] == 0) {
                go = false;
            } else {
                const float alpha = 1.0f - __expf(-sg[k] * d0[k]);
                const float T = 1 - ws;
                const float w = alpha * T;
                ws += w;
                t += d1[k];
                d += w * t;
                r += w * cr[k];
                g += w * cg[k];
                b += w * cb[k];
                if (T < T_thresh) go = false;
                else step++;
            }
        }
    }
    const bool alive = step == 8;
    if (alive) rays_t[index] = t;
    weights_sum[index] = ws;
    depth[index] = d;
    image[index * 3] = r; image[index * 3 + 1] = g; image[index * 3 + 2] = b;
    return alive;
}

__device__ __forceinline__ float readlane_f(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }

// image = acc + (1 - weights_sum) * bg ; depth = clamp(depth - nears, 0) / (fars - nears) (renderer.py:896-899; k_frame_finish, expression by expression) for
// ONE ray whose composite has just been written by this lane
__device__ __forceinline__ void finalize_ray(const FusedArgs& fa, int index) {
    const float k = (1 - fa.weights_sum[index]) * fa.bg;
    fa.image_out[index * 3] = fa.image[index * 3] + k;
    fa.image_out[index * 3 + 1] = fa.image[index * 3 + 1] + k;
    fa.image_out[index * 3 + 2] = fa.image[index * 3 + 2] + k;
    fa.depth_out[index] = fmaxf(fa.depth[index] - fa.nears[index], 0.0f) / (fa.fars_full[index] - fa.nears[index]);
}

#define PN_FUSED_MAXCHUNKS 96  // first-trip chunks per workgroup (a counter each in LDS)
#ifndef PN_FUSED_ACHUNK
#define PN_FUSED_ACHUNK 64  // rays of the frame's first trip a wave takes at a time (whole-frame form): one lane each
#endif

// WHOLE = false: the trips from `fa.trips` on (n_step == 8 there), rays from that trip's alive list.
// WHOLE = true: the whole frame behind k_march_skip.  The frame's first trip (every ray looks for its first sample, n_step = 1) couples the rays
//   through nothing but the NEXT trip's n_step = max(min(N // n_alive, 8), 1) — and n_alive there cannot exceed the rays the skip pre-pass left
//   anything to march for: when those are at most N / 8 (checked here, on the device), every later trip marches 8 samples per ray whatever the first
//   trip finds, and the first trip can run per ray like the others.  Per workgroup:
//     A. its share of the ACTIVE list (chunks of PN_FUSED_ACHUNK rays dealt round-robin over the workgroups, an LDS cursor inside) with ONE lane per ray
//        (march_window<K, MULTI, 1>: every evaluated point is a visited one — the throughput form of k_march), a ray still searching after `a_rounds`
//        rounds in the 64-lane windows; the network on the wave's 64 slots; composite (one sample); the rays that go on are appended to the
//        workgroup's own list;
//     B. (behind a workgroup barrier) the loop below over that list.
//   Not applicable (more than N / 8 active rays, a frame stopped by an error flag): the launch does nothing and says so (fused_trips stays 0).
// NF: form of the network tile — 0: fp32 accuracy through the three-way bf16 split, 1: the autocast (fp16) form, 2: fp32 accuracy through the fp16 hi/lo split
template <int K, bool MULTI, int NF, int MODE>
__global__ void __launch_bounds__(PN_FUSED_WAVES * 64, (PN_FUSED_WAVES + 3) / 4) k_trips_fused(pnm::MarchParams a, pnm2::March2Tables tb, FusedArgs fa) {
    extern __shared__ __attribute__((aligned(16))) uint4 fused_lds[];
    constexpr bool HALF = NF == 1, XF = NF == 2;
    constexpr int IMG16 = (HALF ? PN_NET_HALF_BYTES : (XF ? PN_NET_X_BYTES : PN_NET_SPLIT_BYTES)) / 16;
    constexpr int MAXT = PN_FUSED_MAX_TRIPS;
    constexpr int AC = PN_FUSED_ACHUNK;
    static_assert(AC == 64 || AC == 32, "rays per first-trip chunk");
    constexpr bool WHOLE = MODE == 1;   // the frame's first trip inside the launch, its march included
    constexpr bool FOLD = MODE == 2;    // ... its network, composite and hand-over only: the march ran as launches of its own and left the trip's sample list
    constexpr bool QUEUED = MODE != 0;  // later-trip rays come from the workgroup's own list (filled while the launch runs)
    uint4* wimg = fused_lds;  // the weight image, then the 16 level records (512 B), as in k_nerf_forward
    float4* stage_all = reinterpret_cast<float4*>(fused_lds + IMG16 + 32);
    int* hist = reinterpret_cast<int*>(stage_all + PN_FUSED_WAVES * PN_FUSED_STAGE);  // [3][MAXT]: rays entering trip j, samples emitted, rays through the 64-lane windows
    constexpr bool LXYZ = PN_FUSED_LDS_XYZ != 0 && PN_FUSED_LDS_OUT != 0 && NF != 0;
    float* xyz_all = reinterpret_cast<float*>(hist + 3 * MAXT);   // LXYZ: [waves][64 sample slots][3]
    __shared__ int s_last, s_cursor;
    __shared__ int s_bres, s_bready, s_bhead, s_sres, s_sready, s_shead, s_pending, s_a1done;  // WHOLE: the workgroup's list of rays going on, its straggler queue, chunks without their A3
    __shared__ int s_pend[PN_FUSED_MAXCHUNKS];  // WHOLE: positions of each chunk still open
    __shared__ int s_pref[PN_SEGS + 1];

#if PN_DBG_PHASES
    pnm3::PhaseClock pk_dummy_;
#endif
    const PnTrip* tr = fa.trips;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (PN_FUSED_FRESH_ARGS && blockIdx.x == 0 && threadIdx.x == 0) {   // the kernarg layout fused_args_fresh assumes, checked against the by-value copy
        const FusedArgs& fk = fused_args_fresh(fa);
        const pnm::MarchParams& ak = fused_karg_fresh<pnm::MarchParams, 0>(a);
        const pnm2::March2Tables& tk = fused_karg_fresh<pnm2::March2Tables, 1>(tb);
        if (fk.trips != fa.trips || fk.rays_t != fa.rays_t || fk.N_rays != fa.N_rays || fk.image_out != fa.image_out || fk.seg_back != fa.seg_back || ak.rays_o != a.rays_o ||
            ak.grid != a.grid || ak.hgs != a.hgs || tk.nb != tb.nb || tk.rec != tb.rec) atomicOr(&fa.dev->err, 32);
    }
    const float x_scale = XF ? fa.x_scales[0] : 1.0f, x_rscale = XF ? fa.x_scales[1] : 1.0f;  // uniform: two scalar loads
    int A = 0, sb0 = 0, n_active = 0;
    if (QUEUED) {
        // exclusive prefix of the segment counts of the active list (WHOLE) / of the first trip's sample list (FOLD): the same in every workgroup
        if (threadIdx.x < 64) {
            const int cnt = seg_count(WHOLE ? fa.active_counts : fa.samp_counts, lane);
            int inc = cnt;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int u = __shfl_up(inc, o);
                if (lane >= o) inc += u;
            }
            s_pref[lane + 1] = inc;
            if (lane == 0) s_pref[0] = 0;
        }
        __syncthreads();
        n_active = s_pref[PN_SEGS];   // (FOLD: the rays that found a sample on the first trip — n_alive of the next trip cannot exceed them either)
        const long long chunks = ((long long)n_active + AC - 1) / AC;
        const bool ok = tr->n_alive == (int)fa.N_rays && tr->n_step == 1 && tr->step_base == 0 && (long long)n_active * 8 <= (long long)fa.N_rays &&
                        (chunks + (long long)gridDim.x) * AC <= (long long)fa.blist_cap &&  // (also the room behind the waves' sample slots)
                        (WHOLE ? (chunks + (long long)gridDim.x - 1) / (long long)gridDim.x <= PN_FUSED_MAXCHUNKS : true);
        if (!ok) return;
        sb0 = 1;  // `step` behind the first trip
    } else {
        A = tr->n_alive; sb0 = tr->step_base;
        if (!(A > 0 && tr->n_step == 8 && tr->dense != 0)) return;  // nothing to do / not applicable: the records stay as they are (see the header comment)
    }

    for (int i = threadIdx.x; i < IMG16; i += PN_FUSED_WAVES * 64) wimg[i] = fa.wimg_g[i];
    if (threadIdx.x < 16 * sizeof(PnFusedLevel) / 16) wimg[IMG16 + threadIdx.x] = reinterpret_cast<const uint4*>(fa.lv)[threadIdx.x];
    for (int i = threadIdx.x; i < 3 * MAXT; i += PN_FUSED_WAVES * 64) hist[i] = 0;
    if (threadIdx.x == 0) { s_cursor = 0; s_bres = s_bready = s_bhead = s_sres = s_sready = s_shead = s_a1done = 0; }
    if (QUEUED) {   // work items that can still hand rays on: WHOLE: chunks of AC active rays; FOLD: tiles of 32 first-trip samples
        const int n_items0 = WHOLE ? (n_active + AC - 1) / AC : (n_active + 31) / 32;
        const int mine0 = (int)blockIdx.x < n_items0 ? (n_items0 - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;
        if (threadIdx.x == 0) s_pending = mine0;
        if (WHOLE)
            for (int i = threadIdx.x; i < PN_FUSED_MAXCHUNKS; i += PN_FUSED_WAVES * 64) s_pend[i] = 64;
    }
    __syncthreads();

    const int sub = lane & 7, gbase = lane & ~7, grp = lane >> 3;
    const uint32_t wave_g = blockIdx.x * PN_FUSED_WAVES + wv;
    const uint32_t slotw = (FOLD ? fa.N_rays : 0u) + wave_g * 64u, slot0 = slotw + (uint32_t)grp * 8u;  // (FOLD: the first trip's samples sit in the slots of their rays)
    float4* stage = stage_all + wv * PN_FUSED_STAGE;
    const uint4* __restrict__ wl = wimg + lane;
    const int half = lane >> 5, s32 = lane & 31;
    const PnFusedLevel* lds_lv = reinterpret_cast<const PnFusedLevel*>(wimg + IMG16) + 8 * half;
    __amdgpu_buffer_rsrc_t emb_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(fa.emb_h), 0, (int)fa.emb_bytes, 0x00020000);
    float* const X = LXYZ ? xyz_all + wv * 192 + grp * 24 : fa.xyzs + (size_t)slot0 * 3;   // the group's 8 sample slots
    float* const Dd = LXYZ ? nullptr : fa.dirs + (size_t)slot0 * 3;                           // (LXYZ: nobody reads them)
    float* const dl = fa.deltas + (size_t)slot0 * 2;

    const bool clk = fa.clocks != nullptr;
    unsigned long long c_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, c_t = 0, rounds = 0;
    auto tick = [&](int k) __attribute__((always_inline)) {
        if (clk) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            const unsigned long long n = __builtin_readcyclecounter();
            c_acc[k] += n - c_t;
            c_t = n;
        }
    };
    unsigned long long rt0 = 0;
    if (clk) { c_t = __builtin_readcyclecounter(); rt0 = __builtin_amdgcn_s_memrealtime(); }

    // what stays in registers while a group's ray lives: origin, direction, 1 / direction, end, and the t the composite has reached (rays_t)
    float r_ox = 0.f, r_oy = 0.f, r_oz = 0.f, r_dx = 1.f, r_dy = 1.f, r_dz = 1.f, r_rdx = 1.f, r_rdy = 1.f, r_rdz = 1.f, r_far = 0.f, r_t = 0.f;
    // the network on 32 consecutive sample slots (two lanes per sample: lanes s32 and s32 + 32 hold sample slot_of_lane's two level halves)
    // `lds_xyz` != nullptr (LXYZ, a later trip's round): the sample's position from the wave's LDS area, its direction from the registers of its ray's group
    // (lane `dir_lane`) instead of the global sample slot
    auto network_tile = [&](uint32_t slot, float4* lds_out = nullptr, const float* lds_xyz = nullptr, int dir_lane = 0) __attribute__((always_inline)) {
        pnm3::Float3 p, d;
        if (LXYZ && lds_xyz) {   // (uniform)
            p = *reinterpret_cast<const pnm3::Float3*>(lds_xyz);
            d.x = __shfl(r_dx, dir_lane); d.y = __shfl(r_dy, dir_lane); d.z = __shfl(r_dz, dir_lane);
        } else {
            p = *reinterpret_cast<const pnm3::Float3*>(fa.xyzs + (size_t)slot * 3);
            d = *reinterpret_cast<const pnm3::Float3*>(fa.dirs + (size_t)slot * 3);
        }
        float sigma_logit, e[3];
        if (HALF) {
            float g2[8];
            tile_sigma_net_h<4>(fa.lv, wimg, emb_rsrc, wl, half, fa.net_bound, p.x, p.y, p.z, g2);
            sigma_logit = g2[0];
            __builtin_amdgcn_sched_barrier(0);
            tile_color_net_h(wl, wimg, half, g2, d.x, d.y, d.z, e);
        } else if (XF) {
            const f32x16 h2 = tile_sigma_net_x<PN_BF_LU>(reinterpret_cast<const PnByteLevel*>(lds_lv), fa.emb, wl, half, fa.net_bound, fa.net_inv2b, p.x, p.y, p.z, x_scale);
            sigma_logit = h2[0] * x_rscale;
            __builtin_amdgcn_sched_barrier(0);
            tile_color_net_x(wl, wimg, half, h2, d.x, d.y, d.z, e);
        } else {
            const f32x16 h2 = tile_sigma_net<PN_BF_LU>(reinterpret_cast<const PnByteLevel*>(lds_lv), fa.emb, wl, half, fa.net_bound, fa.net_inv2b, p.x, p.y, p.z);
            sigma_logit = h2[0];
            __builtin_amdgcn_sched_barrier(0);
            tile_color_net(wl, wimg, half, h2, d.x, d.y, d.z, e);
        }
        if (half == 0) {
            const float sg_ = tile_sigma_out(fa.density_scale, sigma_logit);
            const float c0_ = HALF ? tile_rgb_out_h(e[0]) : tile_rgb_out(e[0]), c1_ = HALF ? tile_rgb_out_h(e[1]) : tile_rgb_out(e[1]),
                        c2_ = HALF ? tile_rgb_out_h(e[2]) : tile_rgb_out(e[2]);
            if (lds_out) {   // (uniform) a later trip's round: the composite behind it reads the wave's own LDS area
                *lds_out = make_float4(sg_, c0_, c1_, c2_);
            } else {
                fa.sigmas[slot] = sg_;
                fa.rgbs[(size_t)slot * 3] = c0_; fa.rgbs[(size_t)slot * 3 + 1] = c1_; fa.rgbs[(size_t)slot * 3 + 2] = c2_;
            }
        }
    };
    // ... on the wave's 64 slots: tile 0 = slots 0..31, tile 1 = 32..63; `rm`: lanes whose slots carry something
    // (`own` = false: ONE tile on the slots the lanes name themselves — a tile of the first trip's sample list, FOLD)
    auto network64 = [&](unsigned long long rm, bool own = true, uint32_t lane_slot = 0u) __attribute__((always_inline)) {
#pragma unroll 1
        for (int tile = 0; tile < 2; tile++) {
            if (!((rm >> (32 * tile)) & 0xFFFFFFFFull)) continue;
            network_tile(own ? slotw + 32u * (uint32_t)tile + (uint32_t)s32 : lane_slot, (PN_FUSED_LDS_OUT && own) ? stage + 32 * tile + s32 : nullptr,
                         (LXYZ && own) ? xyz_all + wv * 192 + (32 * tile + s32) * 3 : nullptr, (4 * tile + (s32 >> 3)) * 8);
        }
    };
    auto wave_sync_mem = [&]() __attribute__((always_inline)) {  // the wave's own stores before its own loads of the same addresses by other lanes
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    };

    // ---- WHOLE: the frame's first trip on this workgroup's share of the active list, as three kinds of work items the waves take whenever they hold no
    // ray of a later trip.  Position p of the share (chunk p / AC of the workgroup, lane p % AC) owns sample slot a_base + p — behind the slots of the
    // later trips' waves — and entry p of the workgroup's index list.
    //   A1 (one per chunk)  ONE lane per ray for a_rounds rounds (march_window<K, MULTI, 1>: every evaluated point is a visited one); the rays still
    //                       searching go to the workgroup's straggler queue;
    //   A2 (one per such ray)  64 sequence elements per round with a whole wave;
    //   A3 (one per chunk, run by the wave that finishes the chunk's last ray — s_pend[chunk] counts them down)  network on the chunk's two tiles of 32
    //                       positions, composite (one sample), and the rays that go on appended to the workgroup's list, from which step 1 below refills.
    // Queues are (reserved, ready, head) counters in LDS over arrays in global memory: a producer reserves, writes, waits for its stores and publishes in
    // reservation order; consumers claim below `ready` with a compare-and-swap.  Nothing waits for a consumer, so every reserved entry gets published and
    // every published entry taken: a wave only sleeps while another one is inside an item that can still produce work (s_pending: chunks without their A3).
    int share = 0;
    const int* my_list = nullptr;
    int my_chunks = 0;
    int *a_index = nullptr, *blist = nullptr;
    int4* strag = nullptr;
    uint32_t a_base = 0;
    bool go_on = true;
    auto lds_ld = [](int* p_) __attribute__((always_inline)) { return __hip_atomic_load(p_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    auto lds_st = [](int* p_, int v) __attribute__((always_inline)) { __hip_atomic_store(p_, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    // lane 0: claim up to `want` entries of a queue; returns the first one, `take` = how many (0: none ready)
    auto q_claim = [&](int* head, int* ready, int want, int& take) __attribute__((always_inline)) {
        int base = 0;
        take = 0;
        for (;;) {
            const int h = lds_ld(head), r = lds_ld(ready);
            const int t_ = min(want, r - h);
            if (t_ <= 0) break;
            if (atomicCAS(head, h, h + t_) == h) { base = h; take = t_; break; }
        }
        return base;
    };
    // lane 0, after the wave's stores to the reserved entries [base, base + n) have completed: publish them in reservation order
    auto q_publish = [&](int* ready, int base, int n) __attribute__((always_inline)) {
        while (lds_ld(ready) != base) __builtin_amdgcn_s_sleep(1);
        lds_st(ready, base + n);
    };
    // A3 of chunk ci (see above)
    auto chunk_finish = [&](int ci) __attribute__((always_inline)) {
#pragma unroll 1
        for (int tile = 0; tile < AC / 32; tile++) {
            const int pos = ci * AC + tile * 32 + s32;
            const uint32_t slot = a_base + (uint32_t)pos;
            const bool has = fa.deltas[(size_t)slot * 2] != 0.0f;
            const unsigned long long em = __ballot(has && half == 0);
            if (!em) continue;
            if (lane == 0) {
                atomicAdd(&hist[MAXT], (int)__popcll(em));
                if (a.stats) atomicAdd(a.stats + 3, (unsigned long long)__popcll(em));
            }
            network_tile(slot);
            wave_sync_mem();
            tick(7);
            bool on = false;
            int idx = -1;
            if (has && half == 0) {
                idx = a_index[pos];
                on = composite_one(idx, slot, 1u, fa.T_thresh, fa.rays_t, fa.sigmas, fa.rgbs, fa.deltas, fa.weights_sum, fa.depth, fa.image) && go_on;
                if (!on && fa.finalize) finalize_ray(fa, idx);
            }
            const unsigned long long om = __ballot(on);
            wave_sync_mem();  // the composite's stores (rays_t: read by the wave that takes the ray on) before the ray is published
            if (om) {
                int base = 0;
                if (lane == 0) base = atomicAdd(&s_bres, (int)__popcll(om));
                base = __builtin_amdgcn_readfirstlane(base);
                if (on) blist[base + (int)__popcll(om & ((1ull << lane) - 1ull))] = idx;
                wave_sync_mem();
                if (lane == 0) q_publish(&s_bready, base, (int)__popcll(om));
            }
            tick(8);
        }
        if (lane == 0) atomicSub(&s_pending, 1);
    };
    // lane 0: `n` positions of chunk ci are final; true on the wave that made them all
    auto chunk_arrive = [&](int ci, int n) __attribute__((always_inline)) {
        int left = 1;
        if (lane == 0) left = atomicSub(&s_pend[ci], n) - n;
        return __builtin_amdgcn_readfirstlane(left) == 0;
    };
    if (QUEUED) {
        const int n_chunks = WHOLE ? (n_active + AC - 1) / AC : (n_active + 31) / 32;   // FOLD: tiles of 32 samples, `my_chunks` of them here
        my_chunks = (int)blockIdx.x < n_chunks ? (n_chunks - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;
        const int bcap = ((n_chunks + (int)gridDim.x - 1) / (int)gridDim.x) * (WHOLE ? AC : 32);
        const size_t wg_off = (size_t)blockIdx.x * (size_t)bcap;
        a_index = fa.blist + wg_off;                         // ray id per position
        blist = fa.blist + (size_t)fa.blist_cap + wg_off;    // the rays that go on behind the first trip
        strag = fa.strag + wg_off;                           // (position, t, last_t, ray id) of the rays still searching after the one-lane rounds
        a_base = gridDim.x * (uint32_t)(PN_FUSED_WAVES * 64) + (uint32_t)wg_off;
        my_list = blist;
        go_on = 1u < fa.max_steps;  // renderer.py:836: the loop ends when `step` (1 behind the first trip) reaches max_steps
    } else {
        const int n_packets = (A + 7) >> 3;
        const int my_packets = (int)blockIdx.x < n_packets ? (n_packets - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;
        share = my_packets * 8;
    }

    if (WHOLE) {
        // ---- the first trip's work items (see above), in the order of its dependency chain: a chunk if one is left, else a ray still searching; a wave
        // moves on to the later trips when neither is left nor can appear (every chunk has been through its one-lane rounds)
        for (;;) {
            int fin_chunk = -1;  // a chunk this wave has just made complete
            int ci = my_chunks;
            if (lane == 0 && lds_ld(&s_cursor) < my_chunks) ci = atomicAdd(&s_cursor, 1);
            ci = __builtin_amdgcn_readfirstlane(ci);
            if (ci < my_chunks) {
                const int gi = ((int)blockIdx.x + ci * (int)gridDim.x) * AC + lane;  // position in the concatenation of the active segments
                const bool in_chunk = lane < AC;
                const bool mine = in_chunk && gi < n_active;
                const int pos = ci * AC + lane;
                int idx = -1;
                pnm3::RayConsts c;
                pnm3::frame_consts(a, c);
                c.ox = c.oy = c.oz = 0.f; c.dx = c.dy = c.dz = 1.f; c.rdx = c.rdy = c.rdz = 1.f; c.far = 0.f;
                pnm3::RayState st{0.f, 0.f, 0u};
                bool have = false;
                if (mine) {
                    int lo = 0, hi = PN_SEGS;  // segment with s_pref[seg] <= gi < s_pref[seg + 1]
                    while (hi - lo > 1) {
                        const int mid = (lo + hi) >> 1;
                        if (s_pref[mid] <= gi) lo = mid; else hi = mid;
                    }
                    const int n = fa.active[(size_t)lo * fa.active_seg_cap + (gi - s_pref[lo])];
                    idx = fa.alive[n];
                    pnm3::ray_consts(a, idx, c);
                    have = pnm3::ray_start(a, c, idx, 0.0f, fa.t_resume + n, st);
                }
                if (in_chunk) a_index[pos] = idx;
                const size_t sl = (size_t)a_base + (size_t)(in_chunk ? pos : ci * AC);
                float* const Xa = fa.xyzs + sl * 3;
                float* const Da = fa.dirs + sl * 3;
                float* const La = fa.deltas + sl * 2;
                tick(0);
                const bool done = pnm3::march_window<K, MULTI, 1, PN_FUSED_STAGE, 1>(a, tb, c, 1u, 0, lane, lane, stage, Xa, Da, La, st, fa.a_rounds, have PN_FUSED_PK);
                const bool deferred = have && !done;
                const unsigned long long dm = __ballot(deferred);
                const int n_def = (int)__popcll(dm);
                if (in_chunk && !deferred && !(have && st.step != 0u)) {  // no sample: the slot ends nothing (its tile runs it like the dense trips' empty slots)
                    La[0] = 0.0f; La[1] = 0.0f;
                    Xa[0] = Xa[1] = Xa[2] = 0.0f;
                    Da[0] = Da[1] = Da[2] = 0.0f;
                }
                int qb = 0;
                if (dm) {
                    if (lane == 0) { qb = atomicAdd(&s_sres, n_def); atomicAdd(&hist[2 * MAXT], n_def); }
                    qb = __builtin_amdgcn_readfirstlane(qb);
                    if (deferred) strag[qb + (int)__popcll(dm & ((1ull << lane) - 1ull))] = make_int4(pos, __float_as_int(st.t), __float_as_int(st.last_t), idx);
                }
                wave_sync_mem();
                if (dm && lane == 0) q_publish(&s_sready, qb, n_def);
                if (lane == 0) atomicAdd(&s_a1done, 1);  // (behind the publication: whoever sees every chunk done sees every straggler)
                tick(5);
                if (chunk_arrive(ci, 64 - n_def)) fin_chunk = ci;  // (lanes beyond AC count as final positions of their own)
            }
            int e = 0, got = 0;
            if (ci >= my_chunks && lane == 0) e = q_claim(&s_shead, &s_sready, 1, got);
            got = __builtin_amdgcn_readfirstlane(got);
            if (got) {
                e = __builtin_amdgcn_readfirstlane(e);
                const int4 se = strag[e];
                pnm3::RayConsts c2;
                pnm3::ray_consts(a, se.w, c2);
                pnm3::RayState s2{__int_as_float(se.y), __int_as_float(se.z), 0u};
                const size_t sl = (size_t)a_base + (size_t)se.x;
                pnm3::march_window<K, MULTI, 64, PN_FUSED_STAGE, 1>(a, tb, c2, 1u, lane, 0, lane, stage, fa.xyzs + sl * 3, fa.dirs + sl * 3, fa.deltas + sl * 2, s2, 0x7fffffff, true PN_FUSED_PK);
                if (s2.step == 0u && lane < 8) {
                    if (lane < 2) fa.deltas[sl * 2 + lane] = 0.0f;
                    else if (lane < 5) fa.xyzs[sl * 3 + (lane - 2)] = 0.0f;
                    else fa.dirs[sl * 3 + (lane - 5)] = 0.0f;
                }
                wave_sync_mem();
                tick(6);
                if (chunk_arrive(se.x / AC, 1)) fin_chunk = se.x / AC;
            }
            if (fin_chunk >= 0) chunk_finish(fin_chunk);
            if (ci < my_chunks || got) continue;
            int gone = 0;
            if (lane == 0) gone = (lds_ld(&s_a1done) >= my_chunks && lds_ld(&s_shead) >= lds_ld(&s_sready)) ? 1 : 0;
            if (__builtin_amdgcn_readfirstlane(gone)) break;
            __builtin_amdgcn_s_sleep(16);
            tick(9);
        }
    }

    // Hand-out (WHOLE = false): the alive list is dealt to the workgroups in packets of 8 consecutive entries (neighbouring pixels: a wave's rays share candidate
    // lists), packet p to workgroup p % gridDim — every CU gets the same number of rays from all over the image — and inside a workgroup the waves
    // draw from its share through ONE LDS cursor.  No global atomics (a returning atomic on a shared word was 17 000 cycles of every wave-round with
    // 3 072 waves drawing from 64 cursors), and — what matters more — an even END: the launch is bound by each CU's gather path, so a last
    // generation of rays spread over all CUs at 60 % load takes 60 % of the time, while the same rays on 60 % of the CUs (a global pool: the waves
    // that find it empty exit) take all of it.  (WHOLE: the workgroup's own list, in the order its first trip left it.)
    bool pool_empty = false;
    int index = -1;  // this group's ray (the same on its 8 lanes), -1: none
    int j = 0;       // trips it has been through in this launch
    // ... and what stays in registers while it lives: origin, direction, 1 / direction, end, and the t the composite has reached (rays_t)

    for (;;) {
        // ---- 1. refill the groups without a ray
        if (QUEUED || !pool_empty) {
            const unsigned long long em = __ballot(index < 0 && sub == 0);
            const int need = (int)__popcll(em);
            if (need > 0) {
                const int my_rank = (int)__popcll(em & ((1ull << gbase) - 1ull));
                int base = 0, take = need;
                if (QUEUED) {
                    if (lane == 0) base = q_claim(&s_bhead, &s_bready, need, take);
                    take = __builtin_amdgcn_readfirstlane(take);
                } else {
                    if (lane == 0) base = atomicAdd(&s_cursor, need);
                }
                base = __builtin_amdgcn_readfirstlane(base);
                if (!QUEUED) pool_empty = base + need >= share;
                const int p = base + my_rank;
                const int gpos = QUEUED ? p : ((p >> 3) * (int)gridDim.x + (int)blockIdx.x) * 8 + (p & 7);
                if (index < 0 && (QUEUED ? my_rank < take : (p < share && gpos < A))) {
                    index = QUEUED ? my_list[gpos] : fa.alive[gpos];
                    j = QUEUED ? 1 : 0;
                    pnm3::RayConsts cn;
                    pnm3::ray_consts(a, index, cn);
                    r_ox = cn.ox; r_oy = cn.oy; r_oz = cn.oz; r_dx = cn.dx; r_dy = cn.dy; r_dz = cn.dz; r_rdx = cn.rdx; r_rdy = cn.rdy; r_rdz = cn.rdz; r_far = cn.far;
                    r_t = a.rays_t[index];
                }
            }
        }
        // FOLD: a wave without a ray takes a tile of the first trip's sample list (32 entries of its segments' concatenation, dealt round-robin over the
        // workgroups, an LDS cursor inside): the round below is then network on that tile, composite (one sample), survivors to the workgroup's list.  The
        // tile runs through the SAME network code as the later trips' rounds (a second inlined copy of the tile cost 40 more spilled registers, and the
        // later trips' composite went from 7 to 30 k cycles per round reloading them)
        bool fold_round = false, f_has = false;
        uint32_t f_slot = 0u;
        if (!__any(index >= 0)) {
            if (!QUEUED) break;
            if (FOLD) {
                int ti = my_chunks;
                if (lane == 0 && lds_ld(&s_cursor) < my_chunks) ti = atomicAdd(&s_cursor, 1);
                ti = __builtin_amdgcn_readfirstlane(ti);
                if (ti < my_chunks) {
                    fold_round = true;
                    const int e = ((int)blockIdx.x + ti * (int)gridDim.x) * 32 + s32;   // position in the concatenation of the list's segments
                    f_has = e < n_active;
                    if (f_has) {
                        int lo = 0, hi = PN_SEGS;
                        while (hi - lo > 1) {
                            const int mid = (lo + hi) >> 1;
                            if (s_pref[mid] <= e) lo = mid; else hi = mid;
                        }
                        f_slot = (uint32_t)fa.list_seg[(size_t)lo * fa.list_seg_cap + (e - s_pref[lo])];
                    }
                    // lanes without an entry (the list's last tile) run the tile on the first entry's sample again: a valid position, and they store what
                    // its own lane stores
                    f_slot = f_has ? f_slot : (uint32_t)__builtin_amdgcn_readlane((int)f_slot, (int)__builtin_ctzll(__ballot(f_has)));
                }
            }
            if (!fold_round) {
                // ---- no ray in hand and none to take: wait for the first trip's last chunks / tiles (another wave is in them), or done
                int fin = 0;
                if (lane == 0) fin = (lds_ld(&s_pending) == 0 && lds_ld(&s_bhead) >= lds_ld(&s_bready)) ? 1 : 0;
                if (__builtin_amdgcn_readfirstlane(fin)) break;
                __builtin_amdgcn_s_sleep(32);
                tick(9);
                continue;
            }
        }
        unsigned long long net_mask = 0ull;   // lanes whose slots carry something for the network
        bool have_ray = false;
        if (fold_round) {
            net_mask = __ballot(f_has && half == 0);
            if (lane == 0) atomicAdd(&hist[MAXT], (int)__popcll(net_mask));   // (a.stats + 3, the emitted-sample counter, was advanced by the march launches)
            tick(0);
        } else {
        rounds++;
        tick(0);
        // ---- 2. march: one window round of 8 lanes per ray, then the rays still going with the whole wave
        have_ray = index >= 0;
        if (have_ray && sub == 0) atomicAdd(&hist[j], 1);
        const pnm::MarchParams& am = fused_karg_fresh<pnm::MarchParams, 0>(a);   // the march's uniform inputs: scalar loads here (see fused_karg_fresh)
        const pnm2::March2Tables& tbm = fused_karg_fresh<pnm2::March2Tables, 1>(tb);
        pnm3::RayConsts c;
        pnm3::frame_consts(am, c);
        c.ox = r_ox; c.oy = r_oy; c.oz = r_oz; c.dx = r_dx; c.dy = r_dy; c.dz = r_dz; c.rdx = r_rdx; c.rdy = r_rdy; c.rdz = r_rdz; c.far = r_far;
        pnm3::RayState st{0.f, 0.f, 0u};
        bool have = false;
        if (have_ray) {  // pnm3::ray_start with the ray's t from the register (noise = 0: perturb is off on this path, renderer.py:857)
            float t = r_t;
            t += pnm::clampf(t * am.dt_gamma, c.dt_min, c.dt_max) * 0.0f;
            st.last_t = t;
            st.t = t;
            have = t < c.far;
        }
        const bool done = pnm3::march_window<K, MULTI, 8, PN_FUSED_STAGE, 1>(am, tbm, c, 8u, sub, gbase, lane, stage, X, Dd, dl, st, PN_FUSED_WROUNDS, have PN_FUSED_PK);
        const bool deferred = have && !done;
        tick(1);
        unsigned long long dm = __ballot(deferred && sub == 0);
        if (deferred && sub == 0) atomicAdd(&hist[2 * MAXT + j], 1);
        while (dm) {
            const int L = (int)__builtin_ctzll(dm);
            dm &= dm - 1ull;
            pnm3::RayConsts c2;
            pnm3::frame_consts(am, c2);
            c2.ox = readlane_f(c.ox, L); c2.oy = readlane_f(c.oy, L); c2.oz = readlane_f(c.oz, L);
            c2.dx = readlane_f(c.dx, L); c2.dy = readlane_f(c.dy, L); c2.dz = readlane_f(c.dz, L);
            c2.rdx = readlane_f(c.rdx, L); c2.rdy = readlane_f(c.rdy, L); c2.rdz = readlane_f(c.rdz, L);
            c2.far = readlane_f(c.far, L);
            pnm3::RayState s2{readlane_f(st.t, L), readlane_f(st.last_t, L), (uint32_t)__builtin_amdgcn_readlane((int)st.step, L)};
            const size_t sl = (size_t)slotw + (size_t)(L >> 3) * 8;
            pnm3::march_window<K, MULTI, 64, PN_FUSED_STAGE, 1>(am, tbm, c2, 8u, lane, 0, lane, stage, LXYZ ? xyz_all + wv * 192 + (L >> 3) * 24 : fa.xyzs + sl * 3,
                                                                LXYZ ? nullptr : fa.dirs + sl * 3, fa.deltas + sl * 2, s2, 0x7fffffff, true PN_FUSED_PK);
            if (gbase == L) st.step = s2.step;
        }
        const uint32_t emitted = have_ray ? st.step : 0u;
        if (have_ray && sub == 0 && emitted) {
            atomicAdd(&hist[MAXT + j], (int)emitted);
            if (a.stats) atomicAdd(a.stats + 3, (unsigned long long)emitted);
        }
        // slots the ray did not fill end it in the composite (delta == 0); they run through the network like the dense trips' (zero position)
        if (have_ray && (uint32_t)sub >= emitted) {
            dl[2 * sub] = 0.0f; dl[2 * sub + 1] = 0.0f;
            X[3 * sub] = X[3 * sub + 1] = X[3 * sub + 2] = 0.0f;
            if (!LXYZ) Dd[3 * sub] = Dd[3 * sub + 1] = Dd[3 * sub + 2] = 0.0f;
        }
        wave_sync_mem();
        tick(2);
        net_mask = __ballot(have_ray);
        }
        // ---- 3. network on the wave's 64 slots: tile 0 = groups 0..3, tile 1 = groups 4..7 (a fold round: one tile on the slots its lanes name) — ONE
        // inlined copy of the tile for both kinds of round
        network64(net_mask, !fold_round, f_slot);
        wave_sync_mem();
        tick(fold_round ? 7 : 3);
        if (fold_round) {
            bool on = false;
            int idx = -1;
            if (f_has && half == 0) {
                idx = fa.alive[f_slot];   // n_step == 1: a ray's sample slot is its position in the alive list
                on = composite_one(idx, f_slot, 1u, fa.T_thresh, fa.rays_t, fa.sigmas, fa.rgbs, fa.deltas, fa.weights_sum, fa.depth, fa.image) && go_on;
                if (!on && fa.finalize) finalize_ray(fa, idx);
            }
            const unsigned long long om = __ballot(on);
            wave_sync_mem();
            if (om) {
                int base = 0;
                if (lane == 0) base = atomicAdd(&s_bres, (int)__popcll(om));
                base = __builtin_amdgcn_readfirstlane(base);
                if (on) blist[base + (int)__popcll(om & ((1ull << lane) - 1ull))] = idx;
                wave_sync_mem();
                if (lane == 0) q_publish(&s_bready, base, (int)__popcll(om));
            }
            if (lane == 0) atomicSub(&s_pending, 1);
            tick(8);
            continue;
        }
        // ---- 4. composite (kernel_composite_rays, raymarching.cu:827-923): one lane per ray; a ray goes on iff it used all 8 samples
        int alive = 0;
        const FusedArgs& fc = fused_args_fresh(fa);   // the composite's (and the epilogue's) array pointers: scalar loads here instead of parked registers
        if (have_ray && sub == 0)
            alive = composite_slots8<PN_FUSED_LDS_OUT != 0>(index, slot0, fc.T_thresh, r_t, fc.rays_t, fc.sigmas, fc.rgbs, fc.deltas, fc.weights_sum, fc.depth, fc.image,
                                                            stage + grp * 8) ? 1 : 0;
        alive = __shfl(alive, gbase);
        r_t = __shfl(r_t, gbase);
        if (have_ray) {
            // renderer.py:836: the loop ends when `step` reaches max_steps, whatever is still alive
            if (alive && (uint32_t)(sb0 + 8 * (j + (QUEUED ? 0 : 1))) < fa.max_steps && j + 1 < MAXT) j++;
            else {
                if (QUEUED && fc.finalize && sub == 0) finalize_ray(fc, index);   // the ray leaves the launch: its pixel is final
                index = -1;
            }
        }
        tick(4);
    }
    if (clk && lane == 0) {
        for (int k = 0; k < 5; k++) atomicAdd(fa.clocks + k, c_acc[k]);
        atomicAdd(fa.clocks + 5, rounds);
        atomicAdd(fa.clocks + 6, 1ull);
        atomicAdd(fa.clocks + 7, __builtin_amdgcn_s_memrealtime() - rt0);  // the wave's lifetime on the constant 100 MHz clock
        atomicMax(fa.clocks + 8, rounds);
        atomicMax(fa.clocks + 9, __builtin_amdgcn_s_memrealtime() - rt0);
        if (QUEUED)
            for (int k = 5; k < 10; k++) atomicAdd(fa.clocks + 5 + k, c_acc[k]);
    }

    // ---- end of the launch: per-trip counts to memory; the last workgroup writes the trip records and re-arms the control block
    __syncthreads();
    for (int i = threadIdx.x; i < 3 * MAXT; i += PN_FUSED_WAVES * 64) {
        const int v = hist[i];
        if (v) atomicAdd(fa.ctl + PN_FUSED_CTL_HIST + i, v);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the counts are at the memory side before this workgroup reports in
    __syncthreads();
    if (threadIdx.x == 0) {
        const int old = atomicAdd(fa.ctl + PN_FUSED_CTL_DONE, 1);
        s_last = (old == (int)gridDim.x - 1) ? 1 : 0;
    }
    __syncthreads();
    if (!s_last || threadIdx.x >= 64) return;
    int m = 0;
    long long n_samples_total = 0;   // what pn_render_status reports: samples listed on the first trip + emitted on the later ones
    for (int j0 = 0; j0 < MAXT; j0 += 64) {
        const int jj = j0 + lane;
        int* hp = fa.ctl + PN_FUSED_CTL_HIST + jj;
        int al = __hip_atomic_load(hp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int em = __hip_atomic_load(hp + MAXT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int tl = __hip_atomic_load(hp + 2 * MAXT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (QUEUED && jj == 0) al = (int)fa.N_rays;  // the first trip: every ray (its record was written by the frame prologue)
        m += (int)__popcll(__ballot(al > 0));
        {
            long long e2 = al > 0 ? (long long)em : 0ll;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) e2 += __shfl_xor(e2, o);
            n_samples_total += e2;
        }
        if (al > 0) {
            PnTrip* r = fa.trips + jj;
            if (jj > 0) { r->n_alive = al; r->n_step = 8; r->step_base = sb0 + 8 * (jj - (QUEUED ? 1 : 0)); r->dense = 1; r->n_samples = al * 8; }
            if (QUEUED && jj == 0) r->n_samples = em;  // a list trip: the samples listed (n_emitted stays -1)
            else r->n_emitted = em;
            if (!(FOLD && jj == 0)) fa.tail_diag[jj] = tl;  // (FOLD: the first trip's rays through the tail pass are counted below, from the march's counters)
        }
        __hip_atomic_store(hp, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(hp + MAXT, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(hp + 2 * MAXT, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (FOLD) {  // what the first trip's compaction would have folded out of the march's segment counters: rays handed to the tail pass (front + back lists)
        int tl0 = seg_count(fa.seg_tail, lane) + seg_count(fa.seg_back, lane);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) tl0 += __shfl_xor(tl0, o);
        if (lane == 0) fa.tail_diag[0] = tl0;
    }
    if (lane == 0) {
        PnTrip* r = fa.trips + m;  // the record behind the last trip that had rays: the frame is over
        r->n_alive = 0; r->n_step = 1; r->step_base = QUEUED ? (m == 0 ? 0 : 1 + 8 * (m - 1)) : sb0 + 8 * m; r->dense = 0; r->n_samples = 0; r->n_emitted = 0;
        fa.dev->fused_trips = m;
        if (QUEUED && fa.finalize) {   // the frame's books, as k_frame_finish closes them (the launch started at trip 0 and ran until no ray was left)
            fa.dev->trips_run = min(m, PN_MAX_TRIPS);   // (fused_trips stays: the blocking driver reads it; the next frame's prologue clears it)
            fa.dev->stat_trips = m;
            fa.dev->stat_samples = n_samples_total;
            fa.dev->alive_at_exit = 0;
        }
        __hip_atomic_store(fa.ctl + PN_FUSED_CTL_DONE, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

static size_t fused_lds_bytes(int nf) {
    return (size_t)(nf == 1 ? PN_NET_HALF_BYTES : (nf == 2 ? PN_NET_X_BYTES : PN_NET_SPLIT_BYTES)) + 16 * sizeof(PnFusedLevel) + (size_t)PN_FUSED_WAVES * PN_FUSED_STAGE * sizeof(float4) +
           3 * PN_FUSED_MAX_TRIPS * sizeof(int) + ((PN_FUSED_LDS_XYZ != 0 && PN_FUSED_LDS_OUT != 0 && nf != 0) ? (size_t)PN_FUSED_WAVES * 192 * sizeof(float) : 0);
}

template <int K, bool MULTI, int NF, int MODE>
static int launch_trips_fused_t(uint32_t blocks, hipStream_t st, const pnm::MarchParams& a, const pnm2::March2Tables& tb, const FusedArgs& fa) {
    const size_t lds = fused_lds_bytes(NF);
    static bool granted[PN_MAX_DEVICES] = {false};  // dynamic LDS above 64 KB is opted into per function and DEVICE
    int dev_id = 0;
    PN_HIP_CHECK(hipGetDevice(&dev_id));
    if (dev_id < 0 || dev_id >= PN_MAX_DEVICES || !granted[dev_id]) {
        PN_HIP_CHECK(hipFuncSetAttribute((const void*)k_trips_fused<K, MULTI, NF, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        if (dev_id >= 0 && dev_id < PN_MAX_DEVICES) granted[dev_id] = true;
    }
    k_trips_fused<K, MULTI, NF, MODE><<<blocks, PN_FUSED_WAVES * 64, lds, st>>>(a, tb, fa);
    PN_LAUNCH_CHECK();
    return PN_OK;
}

static int launch_trips_fused(int K, bool multi, int nf, int mode, uint32_t blocks, hipStream_t st, const pnm::MarchParams& a, const pnm2::March2Tables& tb,
                              const FusedArgs& fa) {
#define PN_FUSED_CASE3(K_, M_, N_)                                                                  \
    return mode == 1 ? launch_trips_fused_t<K_, M_, N_, 1>(blocks, st, a, tb, fa)                   \
                     : (mode == 2 ? launch_trips_fused_t<K_, M_, N_, 2>(blocks, st, a, tb, fa) : launch_trips_fused_t<K_, M_, N_, 0>(blocks, st, a, tb, fa))
#define PN_FUSED_CASE2(K_, M_)                 \
    if (nf == 1) PN_FUSED_CASE3(K_, M_, 1);    \
    if (nf == 2) PN_FUSED_CASE3(K_, M_, 2);    \
    PN_FUSED_CASE3(K_, M_, 0)
#define PN_FUSED_CASE(K_)                      \
    if (K == K_) {                             \
        if (multi) { PN_FUSED_CASE2(K_, true); } \
        PN_FUSED_CASE2(K_, false);             \
    }
    PN_FUSED_CASE(1)
    PN_FUSED_CASE(2)
    PN_FUSED_CASE(3)
#undef PN_FUSED_CASE
#undef PN_FUSED_CASE2
#undef PN_FUSED_CASE3
    return PN_ERR_ARG;
}
