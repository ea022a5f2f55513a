// Shared host/device helpers for libpienerf_hip.so (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include "../../include/pienerf_hip.h"

#define PN_WAVE 64
#define PN_MAX_DEVICES 64  // per-device caches of function attributes (hipFuncSetAttribute is per device)

extern thread_local char pn_err_buf[512];

#define PN_HIP_CHECK(expr)                                                                                   \
    do {                                                                                                     \
        hipError_t _e = (expr);                                                                              \
        if (_e != hipSuccess) {                                                                              \
            snprintf(pn_err_buf, sizeof(pn_err_buf), "%s:%d %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            return PN_ERR_HIP;                                                                               \
        }                                                                                                    \
    } while (0)

#define PN_LAUNCH_CHECK() PN_HIP_CHECK(hipGetLastError())

#define PN_REQUIRE(cond)                                                                       \
    do {                                                                                       \
        if (!(cond)) {                                                                         \
            snprintf(pn_err_buf, sizeof(pn_err_buf), "%s:%d argument check failed: %s", __FILE__, __LINE__, #cond); \
            return PN_ERR_ARG;                                                                 \
        }                                                                                      \
    } while (0)

static inline uint32_t pn_div_up(uint64_t a, uint64_t b) { return (uint32_t)((a + b - 1) / b); }
// Tuning override read from the environment (experiments only; the defaults are the shipped configuration).
static inline uint32_t pn_env_u32(const char* name, uint32_t dflt) {
    const char* v = getenv(name);
    if (!v || !*v) return dflt;
    const long x = strtol(v, nullptr, 10);
    return x > 0 ? (uint32_t)x : dflt;
}

// Per-level geometry of the multiresolution hash grid, derived on the host with the reference's formulas
// (gridencoder/src/gridencoder.cu:132-134) so that host libm — not the GPU's approximate exp2 — fixes
// `scale` and `resolution` bit-for-bit.
#define PN_MAX_LEVELS 16
struct PnGridLevels {
    uint32_t L, C;
    uint32_t offset[PN_MAX_LEVELS];        // entries (not floats) before this level
    uint32_t hashmap_size[PN_MAX_LEVELS];  // offsets[l+1] - offsets[l]
    uint32_t resolution[PN_MAX_LEVELS];    // ceil(scale) + 1
    float scale[PN_MAX_LEVELS];            // exp2f(l*S)*H - 1
    uint32_t dense[PN_MAX_LEVELS];         // 0: hashed (fast_hash), else number of dims entering the direct index (3 = fully dense)
    uint32_t mask[PN_MAX_LEVELS];          // hashmap_size - 1 when it is a power of two, else 0
    uint32_t nomod[PN_MAX_LEVELS];         // 1 when the direct index is provably < hashmap_size (no modulo needed)
};

struct PnFusedLevel { float scale; uint32_t offset, m1, m2, mask, dense, dm, xm; };  // dm = dense ? ~0 : 0, xm = dense ? ~0 : mask; see pn_nerf_forward.hip

// The same constants as the fp32 network kernels' encoder wants them (pn_net_tile.h: encode_level): everything in BYTES of the [n_entries, 2] fp32 table, the
// dense and the hashed index form side by side (a level uses one, the other is switched off by zeros):
//   off_b = 8 offset;  dense: m1d = 8 s, m2d = 8 s^2 (s = resolution + 1; both < 2^24), mb = ~0, xmb = 0;  hashed: p1b = 8 P1, p2b = 8 P2 (mod 2^32),
//   mb = xmb = 8 mask, m1d = m2d = 0.   Needs 8 n_entries < 2^32 (pn_net_create checks).
struct PnByteLevel { float scale; uint32_t off_b, p1b, p2b, m1d, m2d, mb, xmb; };
static_assert(sizeof(PnByteLevel) == sizeof(PnFusedLevel), "both level records share the kernels' LDS slot");

int pn_fill_grid_levels(PnGridLevels* g, const int* offsets_host, uint32_t L, uint32_t C, float S, uint32_t H, uint32_t gridtype,
                        int align_corners);

// The packed network context (pn_net in the C ABI).
struct pn_net {
    PnGridLevels levels;
    const float* embeddings;  // device, not owned
    void* wsplit;             // device, owned: the fp32-accurate kernel's LDS weight image, PN_NET_SPLIT_BYTES (pn_nerf_forward.hip)
    void* fused_levels;       // device, owned: PnFusedLevel[16] (pn_nerf_forward.hip): the fp16 kernel's level records
    void* byte_levels;        // device, owned: PnByteLevel[16]: the fp32 kernels' level records
    float bound;
    uint32_t n_entries;       // rows of `embeddings` (offsets[L])
    // fp16 form (the reference under torch.cuda.amp.autocast: gridencoder/grid.py:43-44, nn.Linear in half): built on first use by
    // pn_net_enable_half, refreshed by pn_net_update
    void* emb_half;           // device, owned: embeddings rounded to fp16 (RNE), [n_entries, 2] halves
    void* whalf;              // device, owned: the fp16 kernel's LDS weight image, PN_NET_HALF_BYTES
    // staging for in-place weight refreshes (pn_net_update): pinned host images + the event of the last upload that read them
    void* stage;              // host pinned, PN_NET_SPLIT_BYTES + PN_NET_HALF_BYTES + PN_NET_X_BYTES
    hipEvent_t stage_done;
    // fp16 hi/lo form of the fp32 network (round 4): every fp32 value as two fp16 pieces x = hi + lo (11 + 11 significant bits, round to nearest), a product as
    // hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_f16 with the fp32 accumulator — three products instead of the six of the three-way bf16 split, five vector
    // instructions per pair of split values instead of eleven; 2^-22 relative per value against the bf16 split's 2^-24.  fp16 holds 2^-24 .. 65504: every
    // layer's inputs are carried at a power-of-two scale that puts their interval bound (tables' largest entry x row sums of |W| through the layers) into
    // [2^13, 2^14]; the scales are folded into the weight image, the kernel multiplies the features by x_scale and the density net's outputs by x_rscale
    // (pn_nerf_forward.hip: net_choose_form).  Zero / non-finite bounds: x_ok = 0, the bf16 split runs.
    void* wx;                 // device, owned: the fp16 hi/lo LDS weight image, PN_NET_X_BYTES
    int x_ok;                 // the fp32 network runs in the fp16 hi/lo form (PN_NET_FORM=bf16 forces 0)
    float x_scale, x_rscale;  // the features' scale xs[0]; 1 / xs[2], the scale the density net's 16 outputs leave the matrix pipe at (host copies)
    // What the kernels read: the two words {x_scale, x_rscale} in DEVICE memory, in the 16-byte tail behind the image (wx + PN_NET_X_BYTES: [tables' abs-max,
    // 0, x_scale, x_rscale]), uploaded with it — so launches captured into HIP graphs follow an in-place refresh (round-4 advisor: as by-value kernel
    // arguments a replay ran the new image with the old scales).  The FORM (x_ok: which kernel template and which image) is baked into a captured launch;
    // form_epoch counts its flips and the host (network._net_handle -> harness) refuses to replay graphs captured under another epoch.
    const float* x_scales;
    int form_epoch;
};

// weight image: [20 MFMA operand groups][3 bf16 pieces hi/mid/lo][64 lanes][8 bf16], then the VALU output layer's 192 fp32 weights
#define PN_NET_GROUPS 20
#define PN_NET_SPLIT_W_BYTES (PN_NET_GROUPS * 3 * 64 * 16)
#define PN_NET_SPLIT_BYTES (PN_NET_SPLIT_W_BYTES + 192 * 4)
// fp16 weight image: [20 operand groups][64 lanes][8 fp16], then the output layer's 192 weights (fp16-rounded, stored as fp32)
#define PN_NET_HALF_W_BYTES (PN_NET_GROUPS * 64 * 16)
#define PN_NET_HALF_BYTES (PN_NET_HALF_W_BYTES + 192 * 4)
// fp16 hi/lo weight image: [20 operand groups][hi, lo][64 lanes][8 fp16], then the output layer's 192 fp32 weights
#define PN_NET_X_W_BYTES (PN_NET_GROUPS * 2 * 64 * 16)
#define PN_NET_X_BYTES (PN_NET_X_W_BYTES + 192 * 4)

// internal launcher shared by pn_nerf_forward and the frame driver: evaluates the network on the `count` samples whose
// slot ids are list[0..count) (list == NULL: slots 0..M-1); when ctl_count != NULL the count is read from device memory.
// half != 0: the fp16 form (fp16 tables, fp16 MFMA, half-rounded activations); requires pn_net_enable_half to have been called.
// blocks_cap > 0: at most that many workgroups (the waves stride over the tiles): launches that are expected to find (next to) nothing.
int pn_nerf_forward_launch(const pn_net* net, const float* xyzs, const float* dirs, const int* list, const int* ctl_count, uint32_t M_max,
                           float density_scale, float* sigmas, float* rgbs, int half, hipStream_t stream, uint32_t blocks_cap = 0);
