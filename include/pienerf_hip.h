/* pienerf_hip.h — C ABI of libpienerf_hip.so: the MI355X (gfx950) kernels behind the
 * PIE-NeRF simulate-and-render hot path.
 *
 * Every entry point replaces one reference interface, cited as file:line relative to
 * /root/reference.  Conventions (SURVEY.md §8b "What a C-ABI replacement must export"):
 *   - plain device pointers + extents + scalars; no torch types;
 *   - the CALLER allocates every output (the reference's Python wrappers do, e.g.
 *     raymarching/raymarching.py:415-417) and zero-fills where stated;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); nothing blocks the
 *     host unless stated — the reference's per-call cudaEventSynchronize
 *     (raymarching/src/raymarching.cu:1481-1482) is deliberately not reproduced;
 *   - return value: 0 on success, a PN_ERR_* code otherwise (the pybind functions return void
 *     and TORCH_CHECK-throw; the Python mirror raises RuntimeError on non-zero).
 * Pointers are DEVICE pointers unless the parameter is marked [host].
 */
#ifndef PIENERF_HIP_H
#define PIENERF_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define PN_OK 0
#define PN_ERR_ARG 1      /* unsupported D / C / degree / num_seek_IP, null pointer, ... */
#define PN_ERR_HIP 2      /* a HIP runtime call or kernel launch failed; see pn_last_error() */
#define PN_ERR_CAPACITY 3 /* a caller-provided scratch buffer is too small */

/* Library identity: "pienerf_hip <version> gfx950".  Never NULL. */
const char* pn_version(void);

/* Stream confined to a subset of the GPU's compute units (hipExtStreamCreateWithCUMask), for the pipelined harness: the
 * simulator's chain of small launches runs on `n_cu` CUs of its own, the render streams on the complement, so neither
 * waits for the other's waves to release registers.  Mask bits first_cu .. first_cu+n_cu-1 are set when invert == 0, all
 * the others (of total_cu) when invert != 0; consecutive mask bits fall on consecutive XCDs.  *stream_out is a hipStream_t
 * (wrap it with torch.cuda.ExternalStream); destroy with pn_stream_destroy.  No reference counterpart. */
int pn_stream_create_cu_mask(uint32_t total_cu, uint32_t first_cu, uint32_t n_cu, int invert, void** stream_out);
int pn_stream_destroy(void* stream);
/* Test / tuning hook: rounds (windows of 8 ray points) a ray gets in the first march launch before it is handed to the
 * wave-per-ray tail launch; 0 restores the default (4).  The samples do not depend on it (bit for bit) — tests use 1 and a
 * large value to push every ray through either launch.  Affects renders enqueued afterwards, process-wide. */
int pn_march_set_tail_rounds(int rounds);
/* Tests / experiments: 0 makes the frame driver's skip pre-pass walk hop by hop (rounds 1-2), 1 restarts the hop chain just before the first search cell
 * with candidates and hands stragglers to the windowed march, and — --cut frames — crosses the empty regions of the static background (8^3-voxel blocks of
 * the density bitfield without an occupied voxel on any level, outside the cut box) by walking the ray's t-sequence instead of visiting their voxels (the
 * default), -1 restores the default (environment PN_SKIP_DDA).  Same results bit for bit.  Takes effect for renders enqueued (or captured) afterwards,
 * process-wide. */
int pn_march_set_skip_dda(int on);
/* Number of compute units of the current device. */
int pn_device_cu_count(void);
/* Text of the last PN_ERR_HIP on the calling thread ("" if none). */
const char* pn_last_error(void);

/* ------------------------------------------------------------------ raymarching ---- */

/* raymarching/src/raymarching.h:8 sph_from_ray (kernel raymarching.cu:165-202): coords [N,2] in [-1,1] where each ray leaves the sphere. */
int pn_sph_from_ray(const float* rays_o, const float* rays_d, float radius, uint32_t N, float* coords, void* stream);
/* raymarching/src/raymarching.h:7 near_far_from_aabb (kernel raymarching.cu:91-159). */
int pn_near_far_from_aabb(const float* rays_o, const float* rays_d, const float* aabb, uint32_t N, float min_near, float* nears, float* fars,
                          void* stream);

/* raymarching/src/raymarching.h:20-36 march_rays_quadratic_bending (kernel raymarching.cu:1121-1434,
 * host :1436-1489).  Same argument order as the pybind function (p_def before p_ori).
 * xyzs [M,3], dirs [M,3], deltas [M,2] must be zero-filled, M >= n_alive*n_step.
 * num_seek_IP must be in [1,3] (get_opts.py:97,117-120).  err_flag (int[1], may be NULL) is OR-ed with
 * 1 when a sample's search cell falls outside the spatial-hash grid (reference: device printf "ERROR"). */
int pn_march_rays_quadratic_bending(const int* pig_cnt, const int* pig_bgn, const int* pig_idx, int n_vtx, int n_grid, const float* p_def,
                                    const float* p_ori, const float* F_IP, const float* dF_IP, int max_iter_num, const float* bbmin,
                                    const float* bbmax, float hgs, const int* resolution, int num_seek_IP, float IP_dx, int cut,
                                    const float* cut_bounds, uint32_t n_alive, uint32_t n_step, const int* rays_alive, const float* rays_t,
                                    const float* rays_o, const float* rays_d, float bound, float dt_gamma, uint32_t max_steps, uint32_t C,
                                    uint32_t H, const uint8_t* grid, const float* nears, const float* fars, float* xyzs, float* dirs,
                                    float* deltas, const float* noises, int* err_flag, void* stream);

/* raymarching/src/raymarching.h:18 composite_rays (kernel raymarching.cu:827-923).  In place on rays_alive,
 * rays_t, weights_sum, depth, image. */
int pn_composite_rays(uint32_t n_alive, uint32_t n_step, float T_thresh, int* rays_alive, float* rays_t, const float* sigmas, const float* rgbs,
                      const float* deltas, float* weights_sum, float* depth, float* image, void* stream);

/* nerf/renderer.py:887  rays_alive = rays_alive[rays_alive >= 0]  — stable stream compaction.
 * out [n] receives the survivors in order; *n_out (device int[1]) their count.  scratch: >= pn_compact_scratch_ints(n) ints. */
int pn_compact_rays(const int* rays_alive, uint32_t n, int* out, int* n_out, int* scratch, void* stream);
uint32_t pn_compact_scratch_ints(uint32_t n);

/* nerf/utils.py:355-443 get_pnts_in_grids (+ Warp kernels get_pig_cnt / get_pig_idx / p2g).  Slot order inside a
 * cell is ascending point id (the reference's is atomic-race order).  bbmin [3] and resolution [3] are device arrays,
 * as in the reference.  err_flag (may be NULL) is OR-ed with 2 when a point's cell id is outside [0,n_grid). */
int pn_pnts_in_grids(int n_vtx, int n_grid, const float* pnts, const float* bbmin, float hgs, const int* resolution, int* pig_cnt, int* pig_bgn,
                     int* pig_idx, int* err_flag, void* stream);

/* nerf/utils.py:54-138 get_rays, N=-1 path.  pose: device pointer to the row-major 4x4 cam2world (the reference's `poses[0]`).
 * rays_o/rays_d [H*W,3]. */
int pn_get_rays(const float* pose, float fx, float fy, float cx, float cy, int H, int W, float* rays_o, float* rays_d, void* stream);

/* ------------------------------------------------------------------ gridencoder ---- */

/* gridencoder/src/gridencoder.h:12 grid_encode_forward (kernel gridencoder.cu:87-245, D=3, C in {1,2,4,8}, fp32).
 * dy_dx: NULL, or [B, L, 3, C] (the `calc_grad_inputs` branch, :199-243; training side).  offsets_host [host]: the L+1 int32 level offsets (the reference passes a device tensor; the
 * launcher derives per-level scale / resolution / table size on the host with the reference's formulas :132-134 and
 * hands them to the kernel by value).  outputs [L,B,C] exactly like the reference kernel; with out_bl_major != 0 the
 * kernel writes [B, L*C] directly (what grid.py:57 produces by permute+reshape). */
int pn_grid_encode_forward(const float* inputs, const float* embeddings, const int* offsets_host, float* outputs, uint32_t B, uint32_t D,
                           uint32_t C, uint32_t L, float S, uint32_t H, float* dy_dx, uint32_t gridtype, int align_corners,
                           uint32_t interp, int out_bl_major, void* stream);

/* grid_encode_forward on a half table (AT_DISPATCH_FLOATING_TYPES_AND_HALF, gridencoder.cu:448-471 -> kernel_grid<at::Half,3,C>): embeddings
 * [sO,C] and outputs fp16 (bit patterns), inputs fp32; no dy_dx (inference).  What GridEncoder.forward launches under autocast. */
int pn_grid_encode_forward_half(const float* inputs, const uint16_t* embeddings, const int* offsets_host, uint16_t* outputs, uint32_t B, uint32_t D,
                                uint32_t C, uint32_t L, float S, uint32_t H, uint32_t gridtype, int align_corners, uint32_t interp,
                                int out_bl_major, void* stream);

/* ------------------------------------------------------------------ shencoder ---- */

/* shencoder/src/shencoder.h:9 sh_encode_forward (kernel shencoder.cu:27-123), D=3, degree C in [1,4]; dy_dx NULL or [B, 3, C*C]
 * (:125-355). */
int pn_sh_encode_forward(const float* inputs, float* outputs, uint32_t B, uint32_t D, uint32_t C, float* dy_dx, void* stream);

/* ------------------------------------------------------------------ static inference ops (off the hot path) */

/* march_rays (raymarching/src/raymarching.cu:703-824, wrapper raymarching.py:327-360): the undeformed march against the density
 * bitfield, up to n_step samples per alive ray.  xyzs/dirs [n_alive*n_step,3], deltas [..,2] zero-filled by the caller; noises
 * [n_alive] or NULL (= 0). */
int pn_march_rays(uint32_t n_alive, uint32_t n_step, const int* rays_alive, const float* rays_t, const float* rays_o, const float* rays_d,
                  float bound, float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H, const uint8_t* grid, const float* nears,
                  const float* fars, float* xyzs, float* dirs, float* deltas, const float* noises, void* stream);
/* packbits (raymarching.cu:270-303): bitfield[n] bit i = grid[8n+i] > density_thresh; N = number of bytes. */
int pn_packbits(const float* grid, uint32_t N, float density_thresh, uint8_t* bitfield, void* stream);
/* morton3D / morton3D_invert (raymarching.cu:217-263): coords [N,3] int32 <-> indices [N] int32. */
int pn_morton3D(const int* coords, uint32_t N, int* indices, void* stream);
int pn_morton3D_invert(const int* indices, uint32_t N, int* coords, void* stream);

/* ------------------------------------------------------------------ training ops (off the hot path) */

/* march_rays_train (raymarching/src/raymarching.h:13, kernel raymarching.cu:314-483, wrapper raymarching.py:163-236).  xyzs/dirs [M,3],
 * deltas [M,2] zero-filled by the caller; rays [N,3] = (ray id, point offset, point count); counter[2] += (points demanded, N).
 * DIFFERENCE (documented): the reference hands out offsets and ray rows with atomicAdd (race order); here row n is ray n
 * and offsets are the exclusive prefix sum of the counts (count -> single-workgroup scan -> write), so the output is reproducible.
 * Rays whose range passes M are dropped like the reference's (:415).  noises [N] or NULL. */
int pn_march_rays_train(const float* rays_o, const float* rays_d, const uint8_t* grid, float bound, float dt_gamma, uint32_t max_steps, uint32_t N,
                        uint32_t C, uint32_t H, uint32_t M, const float* nears, const float* fars, float* xyzs, float* dirs, float* deltas, int* rays,
                        int* counter, const float* noises, void* stream);
/* composite_rays_train_forward / _backward (raymarching.h:14-15, kernels raymarching.cu:503-581,604-686). */
int pn_composite_rays_train_forward(const float* sigmas, const float* rgbs, const float* deltas, const int* rays, uint32_t M, uint32_t N, float T_thresh,
                                    float* weights_sum, float* depth, float* image, void* stream);
int pn_composite_rays_train_backward(const float* grad_weights_sum, const float* grad_image, const float* sigmas, const float* rgbs,
                                     const float* deltas, const int* rays, const float* weights_sum, const float* image, uint32_t M, uint32_t N,
                                     float T_thresh, float* grad_sigmas, float* grad_rgbs, void* stream);
/* grid_encode_backward (gridencoder.h:13, kernels gridencoder.cu:248-369): grad [L,B,C]; grad_embeddings [sO,C] zero-filled by the
 * caller, accumulated with hardware fp32 atomics; dy_dx / grad_inputs [B,3] both NULL or both given. */
int pn_grid_encode_backward(const float* grad, const float* inputs, const float* embeddings, const int* offsets_host, float* grad_embeddings,
                            uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, const float* dy_dx, float* grad_inputs,
                            uint32_t gridtype, int align_corners, uint32_t interp, void* stream);
/* The same under autocast (kernel_grid_backward<at::Half>, gridencoder.cu:248-341 with the __half2 atomicAdd of :324-331; grid.py:43-44 casts the table to
 * half): grad [L,B,C] half, grad_embeddings [sO,C] HALF, zero-filled by the caller; every contribution is rounded to half and accumulated with packed
 * half atomics (global_atomic_pk_add_f16), C in {2, 4, 8}.  The sum of halves depends on the order the atomics meet in, here as in the reference: results
 * agree with the fp32 backward to the rounding of half sums.  No input gradients on this path. */
int pn_grid_encode_backward_half(const uint16_t* grad, const float* inputs, const int* offsets_host, uint16_t* grad_embeddings, uint32_t B, uint32_t D,
                                 uint32_t C, uint32_t L, float S, uint32_t H, uint32_t gridtype, int align_corners, uint32_t interp, void* stream);
/* grad_total_variation (gridencoder.h:15, kernel gridencoder.cu:506-611): grad [sO,C] += TV gradient at the cells of `inputs` [B,3] in [0,1]. */
int pn_grad_total_variation(const float* inputs, const float* embeddings, float* grad, const int* offsets_host, float weight, uint32_t B, uint32_t D,
                            uint32_t C, uint32_t L, float S, uint32_t H, uint32_t gridtype, int align_corners, void* stream);
/* sh_encode_backward (shencoder.h:10, kernel shencoder.cu:358-383): grad_inputs [B,3] += grad [B,C*C] . dy_dx [B,3,C*C]. */
int pn_sh_encode_backward(const float* grad, const float* inputs, uint32_t B, uint32_t D, uint32_t C, const float* dy_dx, float* grad_inputs,
                          void* stream);

/* ------------------------------------------------------------------ network -------- */

/* NeRFNetwork.forward (nerf/network.py:98-127): hash grid (16x2) -> 32->64->16 -> exp | SH(16)+15 -> 31->64->64->3 -> sigmoid,
 * no biases, fp32 in / fp32 out, fused in one kernel; the dense layers run on the bf16 matrix cores as a three-way split with fp32
 * accumulation (fp32-accurate: ~5e-6 relative on sigma).  Weights row-major [out,in] as in the state dict.
 * pn_net_create packs them (and the per-level table geometry) into a device-side context. */
typedef struct pn_net pn_net;
int pn_net_create(pn_net** out, const float* embeddings /*device, [n,2]*/, const int* offsets_host /*[L+1]*/, uint32_t L, uint32_t C,
                  float per_level_scale_log2, uint32_t base_resolution, float bound, const float* W0_host, const float* W1_host,
                  const float* W2_host, const float* W3_host, const float* W4_host, void* stream);
void pn_net_destroy(pn_net* net);
/* Refreshes the packed weights of an existing context in place after the parameters changed (training: every optimizer step invalidates
 * them): host-side packing into the context's pinned staging buffer + two asynchronous uploads on `stream`.  No allocation, no stream
 * synchronisation (it waits only for the previous refresh's own upload event).  `embeddings` replaces the table pointer; when the fp16
 * tables exist they are re-rounded from it.  PN_ERR_ARG while `stream` is being captured (the packing cannot be replayed). */
int pn_net_update(pn_net* net, const float* embeddings, const float* W0_host, const float* W1_host, const float* W2_host, const float* W3_host,
                  const float* W4_host, void* stream);
/* Form of the fp32 network's dense layers on `net` (chosen by pn_net_create / pn_net_update from the weights and the tables): 2 = fp16 hi/lo pieces on the
 * fp16 matrix pipe (every fp32 value as hi + lo, 22 significant bits; three products per K chunk; 4e-6 relative on sigma against the sequential fp32 oracle,
 * as the bf16 form), every layer's inputs carried at the power of two that puts their interval bound (tables' largest entry x row sums of |W|) into
 * [2^13, 2^14], the scales folded into the weight image — taken whenever those bounds are finite and positive; 0 = three bf16 pieces, six products (any
 * weights: an all-zero layer, a non-finite weight).  PN_NET_FORM=bf16 in the environment forces 0. */
int pn_net_form(const pn_net* net);
/* Number of times pn_net_update changed that form since pn_net_create.  The per-layer scales of form 2 live in device memory beside the weight image and are
 * read by the kernels, so launches captured into HIP graphs follow an in-place refresh; the FORM itself (kernel template + image) is fixed in a captured
 * launch: graphs captured under another epoch must be captured again (pienerf_amd/harness.py refuses to replay them). */
int pn_net_form_epoch(const pn_net* net);
/* Creates the fp16 copy of the hash tables (`embeddings.to(torch.half)`, gridencoder/grid.py:43-44; round to nearest even) once; the fp16
 * weight image always exists.  Call before the first *_half launch or fp16 render, outside stream capture. */
int pn_net_enable_half(pn_net* net, void* stream);
/* sigmas[M], rgbs[M,3] for xyzs[M,3] in [-bound,bound], dirs[M,3] unit.  density_scale multiplies sigma (renderer.py:875). */
int pn_nerf_forward(const pn_net* net, const float* xyzs, const float* dirs, uint32_t M, float density_scale, float* sigmas, float* rgbs,
                    void* stream);
/* The same under autocast (fp16 tables, half Linear layers on v_mfma_f32_32x32x16_f16, half activations): sigmas fp32 (trunc_exp casts to
 * float, nerf/activation.py:7), rgbs fp32 holding the half-rounded sigmoid outputs. */
int pn_nerf_forward_half(const pn_net* net, const float* xyzs, const float* dirs, uint32_t M, float density_scale, float* sigmas, float* rgbs,
                         void* stream);
/* NeRFNetwork.density (nerf/network.py:129-146): sigma = trunc_exp(h[0]) (no density_scale), geo_feat = h[1:16]; the same fused
 * kernel stopped after the sigma net.  Used off the hot path (point sampling, main_sample.py:164-168).  sigmas [M], geo_feat [M,15]. */
int pn_nerf_density(const pn_net* net, const float* xyzs, uint32_t M, float* sigmas, float* geo_feat, void* stream);
int pn_nerf_density_half(const pn_net* net, const float* xyzs, uint32_t M, float* sigmas, float* geo_feat, void* stream);
/* density_scale * sigma only (no geo_feat written): what update_extra_state needs from density() over 2-4 M cell samples
 * (nerf/renderer.py:493-495); half != 0 selects the fp16 form. */
int pn_nerf_sigma(const pn_net* net, const float* xyzs, uint32_t M, float density_scale, float* sigmas, int half, void* stream);
/* [host] fp32 -> fp16 bit patterns, round to nearest even: the rounding the host side of pn_net_create applies to the weights. */
int pn_host_float_to_half(const float* in_host, uint16_t* out_host, uint32_t n);

/* ------------------------------------------------------------------ whole frame ---- */

/* NeRFRenderer.rund_cuda (nerf/renderer.py:755-907) for one frame with no host synchronisation inside the loop:
 * bbox/resolution of the deformed IPs, spatial hash, near/far, then trips of { march, network, composite,
 * stable compaction } driven by a device-side (n_alive, n_step) record that follows renderer.py:839-846,887-891.
 * Results are those of the op-by-op loop.  Workspace comes from pn_frame_create (sized for N rays, n_vtx IPs). */
typedef struct pn_frame pn_frame;
typedef struct {
    int max_iter_num;     /* --max_iter_num */
    float hash_grid_size; /* opt.hash_grid_size = 1.2*sim_dx (get_opts.py:96) */
    int num_seek_IP;
    float IP_dx;          /* model.IP_dx = 1.05*sim_dx (main_gui.py:56) */
    int cut;
    float cut_bounds[6];
    float bound;
    float min_near;
    float dt_gamma;
    uint32_t max_steps;
    float T_thresh;
    uint32_t cascade;     /* C */
    uint32_t grid_size;   /* H = 128 */
    float density_scale;
    float bg_color;       /* scalar background (renderer.py:803-804: bg_color = 1) */
    int fp16;             /* != 0: the network runs as under torch.cuda.amp.autocast (trainer.py:561, Trainer(fp16=True)): fp16 hash tables
                             (gridencoder/grid.py:43-44), fp16 MFMA layers with half-rounded activations; needs pn_net_enable_half */
    int reuse_tables;     /* != 0: the IP state (p_def, F, dF) is the one of the previous pn_render_deformed on this workspace — keep its bounding box,
                             spatial hash, candidate lists and packed records and only start new rays.  For a frame rendered in ray batches
                             (max_ray_batch, get_opts.py:24): the reference rebuilds get_pnts_in_grids for every rund_cuda call */
    int ray_batch;        /* > 0 (>= 64): the frame is rendered as ray batches of this many rays (opt.max_ray_batch = 4096, get_opts.py:24; the
                             staging loop of nerf/renderer.py:562-576) — every batch with its OWN trip schedule n_step = max(min(N_b // n_alive_b, 8), 1)
                             and its own max_steps count, exactly as if the batches were rendered one after the other, but all batches advance
                             inside the same launches (rays are independent; the alive list stays sorted by ray id, so a batch is a contiguous
                             run of it).  0: one schedule for the whole ray set (what the reference's render_deformed does, renderer.py:587-600).
                             Deformed render only. */
    int throughput;       /* > 0: the THROUGHPUT form of the frame's first trip — its first march pass walks every ray with ONE lane (one visited point of
                             the ray's chain per round, up to this many rounds; rays that outlast them go on in the wave-per-ray windows) instead of
                             evaluating windows of 8 / 64 consecutive lattice elements of which the chain visits one in 4.6: a quarter of the vector work
                             per visited point, bit-identical samples, a longer first trip (~7 us per round).  For pipelines that keep several frames in
                             flight (harness.capture_pipelined with more than one lane: +8 % steps/s on the chair, +13 % on configs[4]); 0: the latency
                             form.  Deformed render only. */
    int throughput_trips; /* with throughput > 0: how many leading trips of the frame march in that form (0 or 1: the first trip only).  Worth it for a
                             trip that still has rays enough to fill the GPU with one lane each (>= ~128 k alive: the second trip of the trex option
                             set, 246 k rays x 3 samples: +11 % steps/s); a trip of few rays with 8 samples each is faster in the windows.  The
                             results do not depend on it.  harness.capture_pipelined picks it from the trip records of its warm-up frame. */
    int ray_tile_w;       /* > 0: the N rays are the row-major pixels of an image this wide (get_rays, nerf/utils.py:65-147 with error_map None).  The
                             frame then walks them in 16 x 4 pixel tiles instead of runs of 64 pixels of a row: rays_alive starts as that permutation
                             of arange(N) (renderer.py:828) and every later list inherits the order, so the 64 rays that share a wave march about
                             equally long and fewer waves hold a silhouette ray (chair: skip pre-pass -17 %, tail pass -6 %, one-lane pass -10 %;
                             the network kernel's gathers share fewer lines, +2 %; +4 % steps/s).  Per-ray results do not depend on the order of
                             the alive list (each ray's samples, its composite and its pixel are its own); ignored unless ray_tile_w % 16 == 0 and
                             N % (4 * ray_tile_w) == 0, with ray_batch > 0 (whose batches are contiguous runs of a sorted alive list) and by
                             pn_render_static. */
    int fused_from;       /* the loop trips from this one on run as ONE persistent launch (csrc/pn_trips_fused.h): once n_alive <= N / 8 the reference's
                             n_step = max(min(N // n_alive, 8), 1) (renderer.py:839-846) is 8 for the rest of the frame, so every ray can loop
                             { march 8 samples; network; composite } on its own until it dies — the per-ray arithmetic, the samples, the pixels and the
                             per-trip counts of the trip-by-trip loop, without its 4-6 launches and its compaction per trip.
                             0: from trip 1 (right behind the frame's first trip; the chair);
                             k >= 1: the first k trips as per-trip launches (a scene whose second trip still has more than N / 8 rays alive: the trex
                             option set — harness.capture_pipelined reads k off its warm-up frame); < 0: never.  If the launch finds n_step < 8 at
                             its first trip it does nothing and the frame is continued like one that ran out of captured trips (pn_render_continue;
                             the blocking pn_render_deformed runs one per-trip trip and tries again).  Not with ray_batch > 0 (batches keep their own
                             n_step), not for pn_render_static, not for max_steps > 1024. */
    int fused_whole;      /* != 0 (with fused_from == 0): the WHOLE frame behind the skip pre-pass in that launch, first trip included, where that
                             applies — the first trip couples the rays only through the next trip's n_step, and n_alive there cannot exceed the rays the
                             skip pre-pass left anything to march for: the launch checks on the device that those are at most N / 8 (the chair: 10 % of the
                             rays meet the bounding box of the integration points) and does nothing otherwise; the blocking pn_render_deformed then runs
                             the first trip as per-trip launches and fuses from trip 1, a fixed-trip render is left unfinished at trip 0 and finished by
                             pn_render_continue.  Inside the launch the first trip is three kinds of work items per workgroup (one lane per ray for a
                             bounded number of rounds; 64-lane windows for the rays still searching; network tile + composite + hand-over per finished
                             chunk of 64 rays) that its waves take whenever they hold no ray of a later trip.  A frame is then prologue (2 launches), skip
                             pre-pass, this launch, epilogue.  Measured on the chair (MI355X): one frame at a time 0.90 ms against 0.86 ms, three frames in
                             flight 1 730 against 1 930 steps/s (the workgroups of the launch hold a CU each while their first trip's dependency chain
                             leaves most of its waves waiting), two frames in flight 1 700-1 740 against 1 560: harness.capture_pipelined switches it on for
                             pipelines of two render lanes (what every rank of a multi-GPU job runs).  Same samples, records and pixels either way. */
    int fused_fold;       /* != 0 (with fused_from <= 1, without fused_whole): the first trip's NETWORK, COMPOSITE and COMPACTION inside the fused launch — the
                             trip's march stays what it is (skip pre-pass, one lane per ray or windows, tail pass: launches that use every CU) and leaves
                             its segmented sample list; the launch runs network tiles of 32 list entries, composites (one sample per ray) and takes the
                             survivors on through a per-workgroup list.  Four launches fewer on a frame's chain (k_list_pack, k_nerf_forward, k_composite,
                             k_compact).  Applies when at most N / 8 rays found a sample on the first trip (checked on the device: then n_step is 8 from the
                             second trip on); otherwise the launch does nothing and the frame goes on as with fused_whole.  Same results bit for bit. */
    int fused_grid;       /* workgroups of the fused launch (0: one per CU, its upper bound — 12 waves and 157 KB of LDS each, so a workgroup has its CU to
                             itself).  A pipeline with several frames in flight gives each frame's launch a part of the GPU, so that the launches of
                             different frames run side by side instead of one after the other and the rest of the frame's kernels (prologue, skip
                             pre-pass, first trip, simulator) find CUs whose LDS is free: measured on the chair with three render lanes, 64 / 96 / 128 / 160 /
                             192 / 256 workgroups: 1 798 / 1 939 / 1 937 / 1 874 / 1 799 / 1 721 steps/s (profiles/r04_fused_grid_sweep.txt);
                             harness.capture_pipelined passes half the CUs when it runs more than one lane.  The results do not depend on it. */
} pn_render_opts;
int pn_frame_create(pn_frame** out, uint32_t max_rays, uint32_t max_vtx, uint32_t max_grid_cells);
void pn_frame_destroy(pn_frame* f);
/* rays_o/rays_d [N,3]; p_def/p_ori [n_vtx,3], F_IP [n_vtx,9], dF_IP [n_vtx,27]; bitfield [C*H^3/8];
 * outputs image [N,3], depth [N], depth_0 [N], weights_sum [N].  stats_host (may be NULL) [host, int64[5]] =
 * {trips, emitted samples, error flags, rays alive at exit, rays left alive by fixed-trip renders on this workspace since
 * pn_frame_reset_unfinished}; reading it synchronises the stream. */
int pn_render_deformed(pn_frame* f, const pn_net* net, const pn_render_opts* opts, const float* rays_o, const float* rays_d, uint32_t N,
                       const float* p_def, const float* p_ori, const float* F_IP, const float* dF_IP, int n_vtx, const uint8_t* bitfield,
                       float* image, float* depth, float* depth_0, float* weights_sum, int64_t* stats_host, void* stream);

/* Non-blocking form of pn_render_deformed for HIP-graph capture / host run-ahead: enqueues exactly n_trips loop trips
 * (trips past the last alive ray cost ~20 us each), the epilogue and an asynchronous copy of the trip records to pinned host
 * memory.  Never synchronises.  The frame is complete iff pn_render_status reports 0 rays alive at exit; otherwise render again
 * with more trips. */
int pn_render_deformed_async(pn_frame* f, const pn_net* net, const pn_render_opts* opts, const float* rays_o, const float* rays_d, uint32_t N,
                             const float* p_def, const float* p_ori, const float* F_IP, const float* dF_IP, int n_vtx, const uint8_t* bitfield,
                             float* image, float* depth, float* depth_0, float* weights_sum, int n_trips, void* stream);
/* stats_host [host, int64[5]] = {trips with alive rays, emitted samples, error flags, rays alive at exit, unfinished total} of the last render on f.
 * synchronize != 0: waits for `stream` first; 0: the caller guarantees the render has completed (e.g. through an event). */
int pn_render_status(pn_frame* f, int64_t* stats_host, int synchronize, void* stream);
/* Zeroes the workspace's running total of rays left alive by fixed-trip renders (stats[4]): a frame rendered as many captured ray batches is
 * verified once, at its end, instead of once per batch. */
int pn_frame_reset_unfinished(pn_frame* f, void* stream);

/* Continues the last render on `f` where its trips stopped (a fixed-trip render that pn_render_status reports with rays alive at exit — the
 * reference's loop simply keeps going, renderer.py:836-891): n_trips more trips (0: blocking, until no ray is alive), then the epilogue again.
 * The workspace holds the frame's tables, ray state and colour accumulator, so only the rays, the bitfield and the SAME output buffers are
 * needed; results equal those of one render with enough trips, bit for bit.  is_static: the frame came from pn_render_static. */
int pn_render_continue(pn_frame* f, const pn_net* net, const pn_render_opts* opts, const float* rays_o, const float* rays_d, uint32_t N,
                       const uint8_t* bitfield, float* image, float* depth, float* depth_0, float* weights_sum, int64_t* stats_host, int n_trips,
                       int is_static, void* stream);

/* NeRFRenderer.run_cuda, eval branch (nerf/renderer.py:305-387): the undeformed render — near / far from `aabb_host` [host, 6 floats:
 * aabb_infer], trips of { march_rays, network, composite_rays, compaction } with the same device-side trip record as pn_render_deformed, then
 * image += (1 - weights_sum) * bg, depth = clamp(depth - nears, 0) / (fars - nears).  n_trips == 0: blocking form (batches of trips
 * until no ray is alive); n_trips > 0: exactly that many trips, no host synchronisation (pn_render_status afterwards).  depth_0 receives the
 * un-normalised depth (the reference discards it).  Off the simulate-and-render hot path. */
int pn_render_static(pn_frame* f, const pn_net* net, const pn_render_opts* opts, const float* rays_o, const float* rays_d, uint32_t N,
                     const float* aabb_host, const uint8_t* bitfield, float* image, float* depth, float* depth_0, float* weights_sum,
                     int64_t* stats_host, int n_trips, void* stream);

/* ------------------------------------------------------------------ frame copies to the host (nerf/trainer.py:589-592) ---- */

/* The reference's device->host boundary: image / depth / depth_0 .cpu().numpy() per frame.  A pn_copier performs such copies without a HIP
 * stream: a host thread waits for `after_event` (a hipEvent_t recorded behind the producing kernels; NULL: no wait), hands the copy to the
 * HSA runtime (SDMA engine) and waits for its completion signal, so the render streams never carry it (this part runs four hardware queues
 * concurrently; a copy on a fifth stream makes everything time-slice, a copy on a render stream holds that stream for the PCIe time).
 * dst_host: pinned host memory (hipHostMalloc / torch pin_memory); src_dev: device memory.  Copies complete in submission order.
 * pn_copier_wait blocks until the copy with that ticket (and every earlier one) has completed; returns PN_ERR_HIP if any copy failed. */
typedef struct pn_copier pn_copier;
int pn_copier_create(pn_copier** out);
void pn_copier_destroy(pn_copier* c);
int pn_copier_submit(pn_copier* c, void* dst_host, const void* src_dev, uint64_t bytes, void* after_event, uint64_t* ticket);
int pn_copier_wait(pn_copier* c, uint64_t ticket);

/* ------------------------------------------------------------------ density-grid state (off the hot path) */

/* NeRFRenderer.mark_untrained_grid (nerf/renderer.py:390-452): density_grid [cascade, H^3] (morton order) gets -1 in every cell that no
 * camera sees — cell centre (2c/(H-1) - 1)(bound_c - bound_c/H), cam = (centre - t) @ R, z > 0 and |x|, |y| inside the frustum padded by one
 * cell.  poses [B,4,4] row-major cam2world on the device; *n_unseen (device int[1]) receives the number of such cells. */
int pn_mark_untrained_grid(const float* poses, uint32_t B, float fx, float fy, float cx, float cy, uint32_t cascade, uint32_t H, float bound,
                           float* density_grid, int* n_unseen, void* stream);
/* update_extra_state, full sweep (renderer.py:466-497): xyzs [cascade*H^3, 3] = jittered centre of every cell, row = cascade*H^3 + morton
 * index; noise [cascade*H^3, 3] uniform in [0,1) (torch.rand_like).  The caller evaluates sigma there (pn_nerf_density) and hands the result
 * to pn_density_grid_update as tmp_grid. */
int pn_density_cells_full(uint32_t cascade, uint32_t H, float bound, const float* noise, float* xyzs, void* stream);
/* update_extra_state, partial sweep of one cascade (renderer.py:499-527): N cells from rand_coords [N,3] (torch.randint(0,H)) + N picks
 * occ[floor(rand_pick * n_occ)] from the cascade's occupied cells (density_grid_cas > 0, compacted on the device: no host round trip);
 * tmp_grid_cas [H^3] is set to -1; indices [2N] (-1 where the occupied list is empty) and jittered xyzs [2N,3] come back; noise [2N,3].
 * scratch: >= pn_density_partial_scratch_ints(H) ints. */
int pn_density_cells_partial(uint32_t cas, uint32_t H, float bound, uint32_t N, const int* rand_coords, const float* rand_pick, const float* noise,
                             const float* density_grid_cas, float* tmp_grid_cas, int* scratch, int* indices, float* xyzs, void* stream);
uint64_t pn_density_partial_scratch_ints(uint32_t H);
/* tmp_grid_cas[indices[m]] = sigmas[m] for indices >= 0 (renderer.py:527). */
int pn_density_scatter(uint32_t n, const int* indices, const float* sigmas, float* tmp_grid_cas, void* stream);
/* renderer.py:535-543: density_grid = where(grid >= 0 & tmp >= 0, max(grid * decay, tmp), grid) over n = cascade*H^3 cells; mean_thresh
 * (device float[2]) = { mean(clamp(grid, 0)), min(mean, density_thresh) } by a fixed-order reduction; bitfield = packbits(grid > threshold)
 * with the threshold read from the device.  partial: >= ceil(n / 256) doubles of scratch. */
int pn_density_grid_update(uint32_t n, float* density_grid, const float* tmp_grid, float decay, float density_thresh, uint8_t* bitfield,
                           double* partial, float* mean_thresh, void* stream);

/* Measurement hook (bench.py).  `enable` is a bitmask: bit 0 (1) — renders on f accumulate march work counters on the device,
 * {marching-loop iterations, candidate entries scanned, per-IP inverse warps, samples emitted}, the units behind the march
 * kernel's algorithmic-bytes figure (the counters add atomics, so never time with this bit set);
 * bit 1 (2) — blocking renders bracket each trip's march and network launches with HIP events (see pn_frame_trip_times);
 * bit 2 (4) — the fused launches of the later trips (pn_render_opts.fused_from) sum per-phase shader-clock cycles over their waves (see
 * pn_frame_fused_clocks; a drain of the memory counters at every phase boundary: never time `value` with it);
 * 0 switches all off.  counters_host (uint64[4], may be NULL): synchronises and reads the totals accumulated so far
 * (before any re-zeroing caused by enabling). */
int pn_frame_march_counters(pn_frame* f, int enable, uint64_t* counters_host, void* stream);
/* With bit 2: clocks_host (uint64[16], may be NULL; synchronises) = cycles summed over the waves of the fused launches on f since the last reset for
 * {hand-out of rays, march (8-lane window round), march (64-lane windows of the rays still going), network, composite}, then wave-rounds, waves,
 * wave lifetimes in 100 MHz ticks (sum), the largest round count and the longest lifetime of a wave; [10..14] (whole-frame form, fused_from = 0): the
 * first trip's one-lane march, its 64-lane windows, its network, its composite + hand-over, the wait at the workgroup barrier behind it; [15]: the form
 * of the last render's fused launch (0 later trips only, 1 whole frame, 2 first trip folded in);
 * *first_trip_out (may be NULL) = the trip at which the last render on f switched to the fused launch, -1 if it did not.  reset != 0: zero the sums. */
int pn_frame_fused_clocks(pn_frame* f, uint64_t* clocks_host, int* first_trip_out, int reset, void* stream);
/* With bit 1 of `enable` set: the per-trip durations (ms, HIP events on the launch stream) of the last blocking render:
 * *n_trips_out entries in each array. */
/* Diagnostics: the device's per-trip records of the last render on `f` as int[max_trips][5] = (n_alive, n_step, step_base, n_samples, n_emitted; -1 on trips whose list is compacted)
 * and the number of rays each trip handed to the wave-per-ray tail pass.  Synchronises the stream. */
int pn_frame_trip_records(pn_frame* f, int* records_host, int* tail_counts_host, int max_trips, void* stream);
int pn_frame_trip_times(pn_frame* f, float* march_ms_host, float* network_ms_host, int max_trips, int* n_trips_out, void* stream);

/* ------------------------------------------------------------------ simulator ------ */

/* Simulator.get_IP_info (simulator/solver.py:402-424) = update_F_kernel (simulator/cuda_utils.py:206-233) + the
 * permute/cast: pos [n_IP,3], F [n_IP,9] (flat c*3+r), dF [n_IP,27] (flat c*9+r*3+j), fp32.
 * topo [n_IP,8] int32; dof [10 n_k,3]; Nx [n_IP,8,10]; dNx [n_IP,8,3,10]; ddNx [n_IP,8,3,3,10] fp64. */
int pn_sim_update_F(int n_IP, const int* topo, const double* dof, const double* Nx, const double* dNx, const double* ddNx, float* pos, float* F,
                    float* dF, void* stream);

/* calc_elastic (simulator/cuda_utils.py:83-121) + volume_invariant_project (simulator/func_utils.py:21-40).
 * RF, VF [n_IP,3,3] fp64 row-major; FF may be NULL. */
int pn_sim_calc_elastic(int n_IP, const int* topo, const double* dNx, const double* dof, double* RF, double* VF, double* FF, void* stream);

/* collect_rhs_IP (simulator/cuda_utils.py:124-151) in its deterministic gather form (the reference ships the same
 * idea unused: collect_rhs_kernel :153-188, CSR built at solver.py:284-313).  csr_bg/csr_cnt [n_k], csr_buf [8 n_IP]
 * of (vid*8+dir) ascending per kernel.  rhs [10 n_k,3] is overwritten. */
int pn_sim_collect_rhs(int n_k, double dx, const int* csr_bg, const int* csr_cnt, const int* csr_buf, const double* mu, const double* lam,
                       const double* dNx, const double* RF, const double* VF, double* rhs, void* stream);

/* Y[n,3] = A[n,n] X[n,3]: the kron(A,I3) form of `global_matrix @ rhs` / `mass_matrix_invt2 @ dof_tilde`
 * (simulator/solver.py:493-496,532-538,576,600). */
int pn_sim_matvec3(int n, const double* A, const double* X, double* Y, void* stream);

/* Simulator.stepforward (simulator/solver.py:595-602) incl. compute_momentum (:574-576) and build_rhs (:541-571).
 * All vectors [10 n_k,3] fp64.  dof and dof_vel are updated in place.  work: >= pn_sim_work_doubles(n_k, n_IP) doubles.
 * prepared != 0: pn_sim_prepare has run on this `work` (the gather's chunk layout is there, and the per-IP rotations the local step's SVD is
 * warm-started from — they carry over from one local/global iteration and substep to the next); 0: the layout is rebuilt in this call and every
 * SVD starts from the identity. */
int pn_sim_stepforward(int n_k, int n_IP, int iters, double dt, double dx, const int* topo, const int* csr_bg, const int* csr_cnt,
                       const int* csr_buf, const double* mu, const double* lam, const double* dNx, const double* dNx_csr, const int* csr_pos,
                       const double* Ainv,
                       const double* Mmat, const double* dof_rest, const double* rhs_rest, const double* rhs_gravity, const double* dof_f,
                       double* dof, double* dof_vel, double* work, int prepared, void* stream);
/* dNx_csr (may be NULL): dNx rows gathered in CSR order, dNx_csr[e] = dNx[csr_buf[e]] (30 doubles each), built once at
 * initialisation; with it collect_rhs streams contiguous memory (one workgroup per kernel) instead of chasing csr_buf.
 * csr_pos (may be NULL; needs dNx_csr): inverse of csr_buf, csr_pos[csr_buf[e]] = e; calc_elastic then also writes P once per
 * neighbour slot in CSR order and the gather has no index left to follow. */
uint64_t pn_sim_work_doubles(int n_k, int n_IP);
/* Simulator.stepforward in its CELL form (csrc/pn_sim.hip: k_cells_elastic_gather): calc_elastic (cuda_utils.py:83-121) and collect_rhs_IP (:124-151)
 * of a local/global iteration as ONE launch, the dense product (solver.py:600-601) as the other — 1 + 2 iters launches per substep instead of 1 + 3 iters.
 * All integration points of one kernel-grid cell share their 8 neighbour kernels (solver.py:186-205); the caller sorts the points by cell and cuts the
 * cells into chunks of <= pn_sim_cells_chunk_ips() points:
 *   chunk_tab [n_chunks][12] int32: {points in the chunk, kernel of neighbour slot 0..7, 0, 0, 0};
 *   dNx_cell  [n_chunks][waves = chunk_ips / 8][15][64][2] fp64: lane l of wave w holds point w * 8 + l / 8, slot l % 8 of the chunk; its 30 gradients
 *             dNx[point, slot, c, x] (flat c * 10 + x) as 15 pairs; zeros for lanes behind the chunk's last point;
 *   mu_cell, lam_cell [n_chunks * chunk_ips] fp64 in chunk order;
 *   kp_bg [n_k + 1], kp_pos [8 n_chunks]: the (chunk * 8 + slot) pairs that refer to a kernel, in ascending order, are that kernel's run of partial sums
 *             [kp_bg[k], kp_bg[k + 1]); kp_pos[chunk * 8 + slot] = the pair's place in it (absolute) — where the chunk stores that partial sum.
 * Same results as pn_sim_stepforward up to the summation order of collect_rhs (1e-16 relative) AND the SVD's stopping rule: this form's warm-started
 * threshold Jacobi stops at off-diagonals <= 1e-11 of the diagonal (squares: 1e-22; pairs below 3e-12 are not rotated) where pn_sim_stepforward's stops at
 * 1e-12 — measured <= 8e-10 relative on the displacements between the forms (tests/test_gpu_persistent.py asserts 1e-8).  The warm start (each point's V of
 * the previous local/global iteration, kept in `work`) makes a substep's bits depend on the step HISTORY: bit-reproducible run to run for identical
 * histories; whoever restores dof / dof_vel to replay a trajectory bit for bit re-runs pn_sim_cells_prepare (Simulator.reset_warm_start) as well.
 * work >= pn_sim_cells_work_doubles doubles, initialised once by pn_sim_cells_prepare (identity rotations for the warm-started SVD, arrival counters). */
/* Which decomposition stands in for wp.svd3 (simulator/cuda_utils.py:107; warp-lang is third-party and absent) in calc_elastic, for every substep /
 * calc_elastic call ENQUEUED after it (process-global; a captured graph keeps the choice it was captured with):
 *   0 (default)  converged Jacobi — the contract: U, V proper rotations, the sign of det F on the last singular value;
 *   n in 1..64   the published algorithm wp.svd3 implements (McAdams et al., UW-Madison TR1690) with n fixed Jacobi sweeps, approximate Givens
 *                quaternions, negating-swap sort, Givens-quaternion QR, the paper's 10-digit constants (8 = double-precision setting, 4 = the paper's
 *                single-precision one).  pn_sim_stepforward, pn_sim_stepforward_cells and pn_sim_calc_elastic honour it; pn_sim_stepforward_coop
 *                returns PN_ERR_ARG while it is set.  oracle/sim_oracle.cpp: svd3_mcadams is the CPU restatement it is tested against. */
int pn_sim_set_svd(int mcadams_sweeps);
int pn_sim_get_svd(void);
int pn_sim_cells_chunk_ips(void);
uint64_t pn_sim_cells_work_doubles(int n_k, int n_chunks);
int pn_sim_cells_prepare(int n_k, int n_chunks, double* work, void* stream);
int pn_sim_stepforward_cells(int n_k, int n_chunks, int iters, double dt, double dx, const int* chunk_tab, const double* dNx_cell, const double* mu_cell,
                             const double* lam_cell, const int* kp_bg, const int* kp_pos, const double* Ainv, const double* Mmat, const double* dof_rest,
                             const double* rhs_rest, const double* rhs_gravity, const double* dof_f, double* dof, double* dof_vel, double* work,
                             void* stream);
/* The local/global iterations of a substep as ONE persistent kernel of n_wg workgroups (one per CU; csrc/pn_sim.hip: k_substep_coop) instead of four
 * launches per iteration: same arguments and results as pn_sim_stepforward (tolerance of the summation orders, ~1e-13 relative), `work` prepared by
 * pn_sim_prepare, `coop` >= pn_sim_coop_bytes(n_k, n_IP, n_wg) bytes prepared by pn_sim_coop_prepare, which also returns plan[3] = {pieces, entries
 * per slot, pieces per kernel at most} to pass on.  pn_sim_coop_bytes returns 0 and pn_sim_coop_prepare PN_ERR_ARG when the scene does not fit the persistent form (n_k > 204,
 * more integration points than 32 n_wg, kernel lists too long): callers keep pn_sim_stepforward then.  Every workgroup must become resident; one that
 * waits longer than ~2 s raises a flag and the launch ends with invalid results: pn_sim_coop_status (synchronous) reads the flag. */
uint64_t pn_sim_coop_bytes(int n_k, int n_IP, int n_wg);
int pn_sim_coop_prepare(int n_k, int n_IP, int n_wg, const int* csr_bg, const int* csr_cnt, void* coop, int* plan_out, void* stream);
int pn_sim_coop_status(const void* coop, int* timed_out);
/* Timing experiments only (environment PN_SIM_COOP_DBG & 4): per-phase tick sums of workgroup 0, see csrc/pn_sim.hip. */
int pn_sim_coop_clocks(const void* coop, uint64_t* ticks9);
int pn_sim_stepforward_coop(int n_k, int n_IP, int iters, double dt, double dx, const int* topo, const double* mu, const double* lam, const double* dNx,
                            const double* dNx_csr, const int* csr_pos, const double* Ainv, const double* Mmat, const double* dof_rest,
                            const double* rhs_rest, const double* rhs_gravity, const double* dof_f, double* dof, double* dof_vel, double* work,
                            void* coop, int n_wg, const int* plan, void* stream);
/* State-independent contents of `work` (once per simulator / per allocation of `work`), see pn_sim_stepforward. */
int pn_sim_prepare(int n_k, int n_IP, const int* csr_bg, const int* csr_cnt, double* work, void* stream);

/* Simulator.update_force (simulator/solver.py:578-588): dof_f [10 n_k,3] is overwritten, in ONE launch, with the pick force f3 of IP `vid`
 * (every other entry zero).  vid < 0: clear_force (:590-593), f3_host / topo / rho / Nx may then be NULL.  Enqueue it on the stream the
 * substeps run on: stream order then decides which substep sees the change. */
int pn_sim_update_force(int n_k, int vid, const double* f3_host, double dx, const int* topo, const double* rho, const double* Nx, double* dof_f,
                        void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PIENERF_HIP_H */
