// ORACLE — TEST INFRASTRUCTURE ONLY.  PINNED (rounds 3-4) against the reference's own kernels: oracle/ref_build.py compiles
// raymarching/src, gridencoder/src and shencoder/src of /root/reference for gfx950 into oracle/_ref/ (git-ignored binaries), and
// tests/test_gpu_ref.py::test_cpu_oracle_equals_the_reference_kernels_directly runs THIS restatement and those kernels on the same
// inputs: march (num_seek_IP 1/2/3, 1/3/5 Newton iterations, --cut), near/far, morton, packbits bit for bit; composite decisions
// equal and values within 2e-6; hash grid within 2e-6 of the contracting build; SH within 1e-6.  get_rays and trunc_exp are pinned by
// reference-run fixtures (tests/golden/make_golden_ref.py, tests/test_golden_ref.py).
//
// CPU restatement of the render half of the PIE-NeRF simulate-and-render hot
// path.  Only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg may load this library; the product path
// (pienerf_amd/) never does.  The reference ships no tests or golden vectors of
// its own for this path; nothing here is checked against anything but the
// reference's kernels (GPU tests) and independent maths (CPU tests).
//
// Each function cites the reference file:line (relative to /root/reference)
// whose arithmetic it restates.  Built with -ffp-contract=off so that every
// float operation rounds once, in source order; the HIP march/composite
// kernels are built the same way, which is what makes their integer decisions
// (voxel index, mip level, cell id, accepted-IP count) comparable bit for bit.
//
// Type-promotion notes that change bits and are restated literally:
//   * `0.5 * (x * mip_rbound + 1) * H` is evaluated in double (0.5 is a double
//     literal) and narrowed to float by clamp()'s parameter
//     (raymarching/src/raymarching.cu:1394-1396).
//   * `bbmax[i] - 1e-6` is a double subtraction narrowed to float (:1203-1205).
//   * `b[i] = Fk_q[i] + 0.5 * dFk_q_q[i] - q_[i]` is double arithmetic (:1296).
//   * `level * H3 + morton` is a float addition narrowed to uint32 (:1398).
#include <cmath>
#include <cfloat>
#include <cstdint>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

inline float clampf(const float x, const float lo, const float hi) { return fminf(hi, fmaxf(lo, x)); }  // raymarching.cu:35-37
inline float signf(const float x) { return copysignf(1.0f, x); }                                      // :31-33

// raymarching.cu:43-48
inline int mip_from_pos(const float x, const float y, const float z, const float max_cascade) {
    const float mx = fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z)));
    int exponent;
    frexpf(mx, &exponent);
    return (int)fminf(max_cascade - 1, fmaxf(0, (float)exponent));
}

// raymarching.cu:50-55  (H arrives as float; `* 0.5` is a double multiply, exact)
inline int mip_from_dt(const float dt, const float H, const float max_cascade) {
    const float mx = (float)((double)(dt * H) * 0.5);
    int exponent;
    frexpf(mx, &exponent);
    return (int)fminf(max_cascade - 1, fmaxf(0, (float)exponent));
}

// raymarching.cu:57-71
inline uint32_t expand_bits(uint32_t v) {
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
inline uint32_t morton3D(uint32_t x, uint32_t y, uint32_t z) {
    return expand_bits(x) | (expand_bits(y) << 1) | (expand_bits(z) << 2);
}

// raymarching.cu:940-984 — flat-index helpers, restated literally (quirk R7q-v:
// dot31 produces index a*3+j, mul31 consumes it as c*3+r).
inline void dot31(const float* T, const float* V, float* M) {
    for (int m = 0; m < 9; m++) M[m] = T[m] * V[0] + T[9 + m] * V[1] + T[18 + m] * V[2];
}
inline void mul31(const float* M, const float* V, float* R) {
    R[0] = M[0] * V[0] + M[3] * V[1] + M[6] * V[2];
    R[1] = M[1] * V[0] + M[4] * V[1] + M[7] * V[2];
    R[2] = M[2] * V[0] + M[5] * V[1] + M[8] * V[2];
}
inline float det3x3(const float* A) {
    return A[0] * (A[4] * A[8] - A[5] * A[7]) - A[1] * (A[3] * A[8] - A[5] * A[6]) + A[2] * (A[3] * A[7] - A[4] * A[6]);
}
inline int inv3x3(const float* A, float* Ai) {
    const float det = det3x3(A);
    if (det == 0) return -1;
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
    return 0;
}

// neighbour visiting order shared by both searches (raymarching.cu:1006-1016, 1083-1093)
const int NBR26[26 * 3] = {
    -1, 0, 0, 0, -1, 0, 0, 0, -1, 1, 0, 0, 0, 1, 0, 0, 0, 1,
    -1, -1, 0, -1, 0, -1, 0, -1, -1, 1, 1, 0, 1, 0, 1, 0, 1, 1,
    -1, 1, 0, -1, 0, 1, 0, -1, 1, 1, -1, 0, 1, 0, -1, 0, 1, -1,
    -1, -1, 1, -1, 1, -1, 1, -1, -1, 1, 1, -1, 1, -1, 1, -1, 1, 1,
    -1, -1, -1, 1, 1, 1};

struct Pig {
    const int *cnt, *bgn, *idx;
    const int* res;
    int n_grid;
};

inline float dist2(const float* pk_, float x, float y, float z) {
    return (pk_[0] - x) * (pk_[0] - x) + (pk_[1] - y) * (pk_[1] - y) + (pk_[2] - z) * (pk_[2] - z);
}

// raymarching.cu:986-1043.  NOTE the offset triple (f,g,h) is applied as
// (g2+f, g1+g, g0+h).
int find_closest_IP(float x, float y, float z, const float* p_def, const Pig& pg, int g0, int g1, int g2) {
    const int* r = pg.res;
    int gid = g2 * r[1] * r[0] + g1 * r[0] + g0;
    float best = 9999.9;  // double literal narrowed to float, as in the reference
    int ip = -1;
    if (gid < 0 || gid >= pg.n_grid) return -1;  // reference: device assert (:995)
    for (int i = 0; i < pg.cnt[gid]; i++) {
        const int t = pg.idx[pg.bgn[gid] + i];
        const float d = dist2(&p_def[t * 3], x, y, z);
        if (d < best) { best = d; ip = t; }
    }
    if (ip == -1) {
        for (int k = 0; k < 26; k++) {
            const int f = NBR26[3 * k], g = NBR26[3 * k + 1], h = NBR26[3 * k + 2];
            if (g2 + f >= r[2] || g2 + f < 0 || g1 + g >= r[1] || g1 + g < 0 || g0 + h >= r[0] || g0 + h < 0) continue;
            gid = (g2 + f) * r[1] * r[0] + (g1 + g) * r[0] + g0 + h;
            for (int i = 0; i < pg.cnt[gid]; i++) {
                const int t = pg.idx[pg.bgn[gid] + i];
                const float d = dist2(&p_def[t * 3], x, y, z);
                if (d < best) { best = d; ip = t; }
            }
        }
    }
    return ip;
}

// raymarching.cu:1045-1118.  Offsets applied as (g0+dx, g1+dy, g2+dz); always
// visits all 27 cells; insertion-sorted on strict '<'.
int find_closest_IPs(float x, float y, float z, const float* p_def, const Pig& pg, int g0, int g1, int g2, int* ips, int K) {
    float dists[10];
    for (int i = 0; i < K; i++) { dists[i] = FLT_MAX; ips[i] = -1; }
    auto visit = [&](int gid) {
        if (gid < 0 || gid >= pg.n_grid) return;
        for (int i = 0; i < pg.cnt[gid]; i++) {
            const int t = pg.idx[pg.bgn[gid] + i];
            const float d = dist2(&p_def[t * 3], x, y, z);
            for (int j = 0; j < K; j++) {
                if (d < dists[j]) {
                    for (int k = K - 1; k > j; k--) { dists[k] = dists[k - 1]; ips[k] = ips[k - 1]; }
                    dists[j] = d;
                    ips[j] = t;
                    break;
                }
            }
        }
    };
    const int* r = pg.res;
    visit(g2 * r[1] * r[0] + g1 * r[0] + g0);
    for (int i = 0; i < 26; i++) {
        const int n0 = g0 + NBR26[3 * i], n1 = g1 + NBR26[3 * i + 1], n2 = g2 + NBR26[3 * i + 2];
        if (n0 >= 0 && n0 < r[0] && n1 >= 0 && n1 < r[1] && n2 >= 0 && n2 < r[2]) visit(n2 * r[1] * r[0] + n1 * r[0] + n0);
    }
    int found = 0;
    for (int i = 0; i < K; i++) if (ips[i] != -1) found++;
    return found;
}

// Per-IP Newton inverse warp, raymarching.cu:1262-1324: solve F q + 1/2 (dF.q) q = q' for q = p - p_ori, q' = x - p_def,
// starting at p = p_ori, at most max_iter_num updates, stopping when |dq|^2 < 1e-12 (checked after the update).
// Returns true when the result is further than IP_dx (inf-norm) from p_ori — the caller then decrements n_IP.
bool newton_warp(const float* x3, const float* pk, const float* pk_, const float* Fk, const float* dFk, int max_iter_num, float IP_dx, float* p) {
    p[0] = pk[0]; p[1] = pk[1]; p[2] = pk[2];
    int num_itr = 0;
    const float q_[3] = {x3[0] - pk_[0], x3[1] - pk_[1], x3[2] - pk_[2]};
    while (num_itr < max_iter_num) {
        const float q[3] = {p[0] - pk[0], p[1] - pk[1], p[2] - pk[2]};
        float dFk_q[9];
        dot31(dFk, q, dFk_q);
        float A[9];
        for (int j = 0; j < 9; j++) A[j] = Fk[j] + dFk_q[j];
        float A_inv[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        inv3x3(A, A_inv);  // quirk R7q-ii: failure is never acted on; A_inv stays 0 (:1285-1287)
        float Fk_q[3], dFk_q_q[3], b[3], dq[3];
        mul31(Fk, q, Fk_q);
        mul31(dFk_q, q, dFk_q_q);
        for (int i = 0; i < 3; i++) b[i] = (float)(((double)Fk_q[i] + 0.5 * (double)dFk_q_q[i]) - (double)q_[i]);
        mul31(A_inv, b, dq);
        p[0] -= dq[0];
        p[1] -= dq[1];
        p[2] -= dq[2];
        if ((double)(dq[0] * dq[0] + dq[1] * dq[1] + dq[2] * dq[2]) < 1e-12) break;
        num_itr++;
    }
    const float e0 = p[0] - pk[0], e1 = p[1] - pk[1], e2 = p[2] - pk[2];
    return fabsf(e0) > IP_dx || fabsf(e1) > IP_dx || fabsf(e2) > IP_dx;
}

struct MarchArgs {
    Pig pg;
    int n_vtx;
    const float *p_ori, *p_def, *F_IP, *dF_IP;
    int max_iter_num;
    const float *bbmin, *bbmax;
    float hgs;
    int num_seek_IP;
    float IP_dx;
    bool cut;
    const float* cut_bounds;
    uint32_t n_step;
    const float *rays_t, *rays_o, *rays_d;
    float bound, dt_gamma;
    uint32_t max_steps, C, H;
    const uint8_t* grid;
    const float* fars;
};

// One ray of kernel_march_rays_quadratic_bending, raymarching.cu:1121-1434.
// Returns the number of samples written.  xyzs/dirs/deltas point at this
// ray-slot's first sample.
// optional instrumentation (ORC_MARCH_STATS=1): loop iterations per call / max per ray / iterations with candidates
static long long g_march_iters = 0, g_march_found = 0;
static int g_march_max = 0;
static std::vector<int> g_march_hist(16, 0);
static const bool g_march_stats = getenv("ORC_MARCH_STATS") != nullptr;

uint32_t march_one(const MarchArgs& a, int index, float noise, float* xyzs, float* dirs, float* deltas, int* oob_flag) {
    int iters = 0, iters_found = 0;
    const float* ro = a.rays_o + (size_t)index * 3;
    const float* rd = a.rays_d + (size_t)index * 3;
    const float ox = ro[0], oy = ro[1], oz = ro[2];
    const float dx = rd[0], dy = rd[1], dz = rd[2];
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
    const float* bbmin = a.bbmin;
    const float* bbmax = a.bbmax;
    const float* cb = a.cut_bounds;
    const int* res = a.pg.res;

    while (t < far && step < a.n_step) {
        iters++;
        bool found = false;
        float x, y, z;
        if (a.cut) {
            x = clampf(ox + t * dx, -a.bound, a.bound);
            y = clampf(oy + t * dy, -a.bound, a.bound);
            z = clampf(oz + t * dz, -a.bound, a.bound);
        } else {
            x = clampf(ox + t * dx, bbmin[0], (float)((double)bbmax[0] - 1e-6));
            y = clampf(oy + t * dy, bbmin[1], (float)((double)bbmax[1] - 1e-6));
            z = clampf(oz + t * dz, bbmin[2], (float)((double)bbmax[2] - 1e-6));
        }
        // quirk R7q-i: `x < cut_bounds[3]` where y is meant (:1210)
        if (!a.cut || (x > cb[0] && x < cb[1] && y > cb[2] && x < cb[3] && z > cb[4] && z < cb[5])) {
            float x_map = 0.0f, y_map = 0.0f, z_map = 0.0f;
            const int g0 = (int)floorf((x - bbmin[0]) / a.hgs);
            const int g1 = (int)floorf((y - bbmin[1]) / a.hgs);
            const int g2 = (int)floorf((z - bbmin[2]) / a.hgs);
            const bool oob = (g0 < 0 || g1 < 0 || g2 < 0 || g0 >= res[0] || g1 >= res[1] || g2 >= res[2]);
            if (oob && oob_flag) *oob_flag = 1;  // reference: printf("ERROR ...") and reads out of range (:1221-1222)
            int IPs[10] = {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1};
            int n_IP = 0;
            if (!oob) {
                if (a.num_seek_IP == 1) {
                    const int ip = find_closest_IP(x, y, z, a.p_def, a.pg, g0, g1, g2);
                    if (ip == -1) n_IP = 0; else { n_IP = 1; IPs[0] = ip; }
                } else {
                    n_IP = find_closest_IPs(x, y, z, a.p_def, a.pg, g0, g1, g2, IPs, a.num_seek_IP);
                }
            }
            found = n_IP > 0;
            if (found) iters_found++;
            if (found) {
                // quirks R7q-iii/iv: n_IP-- inside the loop it bounds; strict '<' on z only (:1246-1251)
                for (int k = 0; k < n_IP; k++) {
                    const float* pk_ = &a.p_def[IPs[k] * 3];
                    if (pk_[0] <= bbmin[0] || pk_[1] <= bbmin[1] || pk_[2] < bbmin[2] || pk_[0] >= bbmax[0] || pk_[1] >= bbmax[1] || pk_[2] >= bbmax[2]) n_IP--;
                }
            }
            if (n_IP <= 0) found = false;
            if (found) {
                const float p_[3] = {x, y, z};
                float ps[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
                for (int k = 0; k < n_IP; k++) {
                    float p[3];
                    const bool reject = newton_warp(p_, &a.p_ori[IPs[k] * 3], &a.p_def[IPs[k] * 3], &a.F_IP[IPs[k] * 9], &a.dF_IP[IPs[k] * 27],
                                                    a.max_iter_num, a.IP_dx, p);
                    if (reject) n_IP--;  // quirk R7q-iii (:1316-1319)
                    ps[3 * k] = p[0];
                    ps[3 * k + 1] = p[1];
                    ps[3 * k + 2] = p[2];
                }
                if (n_IP == 1) {
                    x_map = ps[0]; y_map = ps[1]; z_map = ps[2];
                } else if (n_IP == 2) {
                    float dist[2];
                    for (int k = 0; k < 2; k++) {
                        const float* pk = &a.p_ori[IPs[k] * 3];
                        dist[k] = sqrtf((pk[0] - x) * (pk[0] - x) + (pk[1] - y) * (pk[1] - y) + (pk[2] - z) * (pk[2] - z));
                    }
                    const float dist_sum = dist[0] + dist[1];
                    const float w0 = dist[1] / dist_sum, w1 = dist[0] / dist_sum;
                    x_map = w0 * ps[0] + w1 * ps[3];
                    y_map = w0 * ps[1] + w1 * ps[4];
                    z_map = w0 * ps[2] + w1 * ps[5];
                } else if (n_IP == 3) {
                    float dist[3];
                    for (int k = 0; k < 3; k++) {
                        const float* pk = &a.p_ori[IPs[k] * 3];
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
                // n_IP == 0 after rejection: the sample is mapped to the origin (:1372-1374)
                x = x_map; y = y_map; z = z_map;
            }
        } else {
            found = true;  // cut mode, outside the cut box: static background sample (:1380-1383)
        }

        const float dt = clampf(t * a.dt_gamma, dt_min, dt_max);
        const int level = std::max(mip_from_pos(x, y, z, (float)C), mip_from_dt(dt, (float)H, (float)C));
        const float mip_bound = fminf(scalbnf(1, level), a.bound);
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
            // quirk R7q-vii: when !found this uses the un-warped point (:1422-1431)
            const float tx = (((nx + 0.5f + 0.5f * signf(dx)) * rH * 2 - 1) * mip_bound - x) * rdx;
            const float ty = (((ny + 0.5f + 0.5f * signf(dy)) * rH * 2 - 1) * mip_bound - y) * rdy;
            const float tz = (((nz + 0.5f + 0.5f * signf(dz)) * rH * 2 - 1) * mip_bound - z) * rdz;
            const float tt = t + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
            do { t += clampf(t * a.dt_gamma, dt_min, dt_max); } while (t < tt);
        }
    }
    if (g_march_stats) {
#pragma omp critical
        {
            g_march_iters += iters; g_march_found += iters_found;
            if (iters > g_march_max) g_march_max = iters;
            int b = 0; while ((1 << b) <= iters && b < 15) b++;
            g_march_hist[b]++;
        }
    }
    return step;
}

// kernel_composite_rays for one alive slot n, raymarching.cu:827-923
void composite_one(uint32_t n, uint32_t n_step, float T_thresh, int* rays_alive, float* rays_t, const float* sigmas, const float* rgbs,
                   const float* deltas, float* weights_sum, float* depth, float* image) {
    const int index = rays_alive[n];
    sigmas += (size_t)n * n_step;
    rgbs += (size_t)n * n_step * 3;
    deltas += (size_t)n * n_step * 2;
    float t = rays_t[index];
    float ws = weights_sum[index], d = depth[index];
    float r = image[index * 3], g = image[index * 3 + 1], b = image[index * 3 + 2];
    uint32_t step = 0;
    while (step < n_step) {
        if (deltas[0] == 0) break;
        const float alpha = 1.0f - expf(-sigmas[0] * deltas[0]);  // reference: __expf (fast approx)
        const float T = 1 - ws;
        const float w = alpha * T;
        ws += w;
        t += deltas[1];
        d += w * t;
        r += w * rgbs[0];
        g += w * rgbs[1];
        b += w * rgbs[2];
        if (T < T_thresh) break;
        sigmas++; rgbs += 3; deltas += 2; step++;
    }
    if (step < n_step) rays_alive[n] = -1; else rays_t[index] = t;
    weights_sum[index] = ws;
    depth[index] = d;
    image[index * 3] = r; image[index * 3 + 1] = g; image[index * 3 + 2] = b;
}

// gridencoder/src/gridencoder.cu:50-84
inline uint32_t fast_hash3(const uint32_t p[3]) { return (p[0] * 1u) ^ (p[1] * 2654435761u) ^ (p[2] * 805459861u); }
inline uint32_t grid_index(uint32_t gridtype, bool align_corners, uint32_t C, uint32_t hashmap_size, uint32_t resolution, const uint32_t p[3]) {
    uint32_t stride = 1, index = 0;
    for (uint32_t d = 0; d < 3 && stride <= hashmap_size; d++) {
        index += p[d] * stride;
        stride *= align_corners ? resolution : (resolution + 1);
    }
    if (gridtype == 0 && stride > hashmap_size) index = fast_hash3(p);
    return (index % hashmap_size) * C;
}

// ---- fp16 (the reference under torch.cuda.amp.autocast, trainer.py:561 with Trainer(fp16=True); BASELINE configs[4]) ----------------------
// hround(x): x rounded to the nearest fp16 value (ties to even), returned as float — what `tensor.to(torch.half)`, c10::Half(float) and
// __float2half do.  Software, so that the oracle does not depend on compiler / CPU fp16 support.
static inline float hround(float f) {
    uint32_t x;
    memcpy(&x, &f, 4);
    const uint32_t sign = x & 0x80000000u;
    x &= 0x7fffffffu;
    if (x >= 0x7f800000u) return f;                                                   // inf / nan
    if (x >= 0x477ff000u) { const uint32_t inf = sign | 0x7f800000u; float r; memcpy(&r, &inf, 4); return r; }  // >= 65520 -> inf
    if (x < 0x33000001u) { float r; memcpy(&r, &sign, 4); return r; }                 // <= 2^-25 -> (signed) zero
    const int e = (int)(x >> 23) - 127;
    const int drop = (e < -14) ? (13 + (-14 - e)) : 13;                               // low significand bits that do not fit a half
    const uint32_t m = (x & 0x7fffffu) | 0x800000u;
    uint32_t q = m >> drop;
    const uint32_t rem = m & ((1u << drop) - 1u), halfway = 1u << (drop - 1);
    if (rem > halfway || (rem == halfway && (q & 1u))) q++;
    const float r = ldexpf((float)q, e - 23 + drop);                                   // exact: q < 2^12
    return sign ? -r : r;
}
static int g_half = 0;  // orc_set_half: nerf_one / the render drivers restate the autocast arithmetic

// kernel_grid<scalar_t,3,C> for one (sample, level), gridencoder.cu:87-197 (dy_dx == nullptr path).  half == false: scalar_t = float.
// half == true: scalar_t = at::Half (gridencoder/grid.py:43-44 casts the table; `table` holds the half values as floats): positions and
// weights stay float (:137-139), and `results[ch] += w * grid[index + ch]` (:184) is, with c10::Half's operators, Half(float * float(Half))
// followed by Half + Half = Half(float + float): two roundings to half per corner.
void grid_one(const float* in3, const float* table /*level base*/, uint32_t hashmap_size, float scale, uint32_t resolution, uint32_t C, uint32_t gridtype,
              bool align_corners, uint32_t interp, float* out, bool half = false) {
    bool oob = false;
    for (int d = 0; d < 3; d++) if (in3[d] < 0 || in3[d] > 1) oob = true;
    if (oob) { for (uint32_t c = 0; c < C; c++) out[c] = 0; return; }
    float pos[3];
    uint32_t pg[3];
    for (int d = 0; d < 3; d++) {
        // single-rounding multiply-add: nvcc's default -fmad=true contracts `inputs[d] * scale + 0.5f` (gridencoder.cu:143)
        // and at fine levels (pos ~ 2000, ulp 1.2e-4) the choice moves the interpolation weights, so it is made explicit
        // here and in the HIP kernels.
        pos[d] = fmaf(in3[d], scale, align_corners ? 0.0f : 0.5f);
        pg[d] = (uint32_t)floorf(pos[d]);
        pos[d] -= (float)pg[d];
        if (interp == 1) pos[d] = pos[d] * pos[d] * (3.0f - 2.0f * pos[d]);
    }
    float res[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (uint32_t idx = 0; idx < 8; idx++) {
        float w = 1;
        uint32_t pl[3];
        for (int d = 0; d < 3; d++) {
            if ((idx & (1u << d)) == 0) { w *= 1 - pos[d]; pl[d] = pg[d]; }
            else { w *= pos[d]; pl[d] = pg[d] + 1; }
        }
        const uint32_t index = grid_index(gridtype, align_corners, C, hashmap_size, resolution, pl);
        if (half) for (uint32_t c = 0; c < C; c++) res[c] = hround(res[c] + hround(w * table[index + c]));
        else for (uint32_t c = 0; c < C; c++) res[c] += w * table[index + c];
    }
    for (uint32_t c = 0; c < C; c++) out[c] = res[c];
}

inline void level_params(uint32_t level, float S, uint32_t H, float* scale, uint32_t* resolution) {
    *scale = exp2f(level * S) * H - 1.0f;            // gridencoder.cu:133
    *resolution = (uint32_t)ceilf(*scale) + 1;       // :134
}

// shencoder/src/shencoder.cu:42-68 (degree <= 4).  Constants are the closed
// forms quoted in the reference's comments, evaluated in double then narrowed.
void sh_high_bands(const float* in, uint32_t C, float* out, float* gx, float* gy, float* gz);  // bands 4..7, defined with the training-side code below
void sh_one(const float* in, uint32_t C, float* out) {
    static const double PI_ = 3.14159265358979323846;
    static const float c0 = (float)(1.0 / (2.0 * std::sqrt(PI_)));
    static const float c1 = (float)(std::sqrt(3.0) / (2.0 * std::sqrt(PI_)));
    static const float c2a = (float)(std::sqrt(15.0) / (2.0 * std::sqrt(PI_)));
    static const float c2b = (float)(3.0 * std::sqrt(5.0) / (4.0 * std::sqrt(PI_)));
    static const float c2c = (float)(std::sqrt(5.0) / (4.0 * std::sqrt(PI_)));
    static const float c2d = (float)(std::sqrt(15.0) / (4.0 * std::sqrt(PI_)));
    static const float c3a = (float)(std::sqrt(70.0) / (8.0 * std::sqrt(PI_)));
    static const float c3b = (float)(std::sqrt(105.0) / (2.0 * std::sqrt(PI_)));
    static const float c3c = (float)(std::sqrt(42.0) / (8.0 * std::sqrt(PI_)));
    static const float c3d = (float)(std::sqrt(7.0) / (4.0 * std::sqrt(PI_)));
    static const float c3e = (float)(std::sqrt(105.0) / (4.0 * std::sqrt(PI_)));
    const float x = in[0], y = in[1], z = in[2];
    const float xy = x * y, xz = x * z, yz = y * z, x2 = x * x, y2 = y * y, z2 = z * z;
    out[0] = c0;
    if (C <= 1) return;
    out[1] = -c1 * y;
    out[2] = c1 * z;
    out[3] = -c1 * x;
    if (C <= 2) return;
    out[4] = c2a * xy;
    out[5] = -c2a * yz;
    out[6] = c2b * z2 - c2c;
    out[7] = -c2a * xz;
    out[8] = c2d * x2 - c2d * y2;
    if (C <= 3) return;
    out[9] = c3a * y * (-3.0f * x2 + y2);
    out[10] = c3b * xy * z;
    out[11] = c3c * y * (1.0f - 5.0f * z2);
    out[12] = c3d * z * (5.0f * z2 - 3.0f);
    out[13] = c3c * x * (1.0f - 5.0f * z2);
    out[14] = c3e * z * (x2 - y2);
    out[15] = c3a * x * (-x2 + 3.0f * y2);
}

struct Net {
    const float* embeddings;
    const int* offsets;
    uint32_t L, Cf, Hbase;
    float S;
    float bound;
    const float *W0, *W1, *W2, *W3, *W4;  // [64,32] [16,64] [64,31] [64,64] [3,64] row-major (out,in), nerf/network.py:36-71
    bool half = false;                    // autocast: embeddings / W* below are the half-rounded copies (HalfNet)
};

// The half-rounded copies autocast makes: embeddings.to(torch.half) (gridencoder/grid.py:44) and, inside every nn.Linear, weight.to(half).
struct HalfNet {
    std::vector<float> emb, W[5];
    Net net;
    HalfNet(const Net& src, size_t n_emb_floats) {
        emb.resize(n_emb_floats);
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < (int64_t)n_emb_floats; i++) emb[i] = hround(src.embeddings[i]);
        const float* w[5] = {src.W0, src.W1, src.W2, src.W3, src.W4};
        const size_t n[5] = {64 * 32, 16 * 64, 64 * 31, 64 * 64, 3 * 64};
        for (int k = 0; k < 5; k++) { W[k].resize(n[k]); for (size_t i = 0; i < n[k]; i++) W[k][i] = hround(w[k][i]); }
        net = src;
        net.embeddings = emb.data();
        net.W0 = W[0].data(); net.W1 = W[1].data(); net.W2 = W[2].data(); net.W3 = W[3].data(); net.W4 = W[4].data();
        net.half = true;
    }
};

// NeRFNetwork.forward for one sample, nerf/network.py:98-127 + gridencoder/grid.py:145-161
// (u = (x+bound)/(2*bound)) + shencoder/sphere_harmonics.py:75-87 (size = 1).
void nerf_one(const Net& nt, const float* xyz, const float* dir, float* sigma, float* rgb) {
    // nt.half (autocast, fp16): every nn.Linear takes half inputs and half weights, accumulates in float and returns half (one rounding per
    // output; the accumulation ORDER inside cuBLAS is not knowable — sequential here); ReLU and the slices are exact in half; trunc_exp
    // casts its input to float (activation.py:7); SHEncoder returns float (sphere_harmonics.py:16) and torch.cat([d, geo_feat]) promotes to
    // float, so the SH values are rounded to half by the first colour Linear's input cast; torch.sigmoid of a half tensor computes in float
    // and rounds to half.  H(x) is that rounding; the identity otherwise.
    const bool hf = nt.half;
    auto H = [hf](float v) { return hf ? hround(v) : v; };
    float u[3];
    for (int d = 0; d < 3; d++) u[d] = (xyz[d] + nt.bound) / (2 * nt.bound);
    float enc[64];
    for (uint32_t l = 0; l < nt.L; l++) {
        float scale; uint32_t res;
        level_params(l, nt.S, nt.Hbase, &scale, &res);
        const uint32_t hs = (uint32_t)(nt.offsets[l + 1] - nt.offsets[l]);
        grid_one(u, nt.embeddings + (size_t)(uint32_t)nt.offsets[l] * nt.Cf, hs, scale, res, nt.Cf, 0, false, 0, enc + l * nt.Cf, hf);
    }
    const uint32_t in_dim = nt.L * nt.Cf;  // 32
    float h1[64], h2[16];
    for (int j = 0; j < 64; j++) {
        float s = 0;
        for (uint32_t k = 0; k < in_dim; k++) s += nt.W0[j * in_dim + k] * enc[k];
        s = H(s);
        h1[j] = s > 0 ? s : 0;
    }
    for (int j = 0; j < 16; j++) {
        float s = 0;
        for (int k = 0; k < 64; k++) s += nt.W1[j * 64 + k] * h1[k];
        h2[j] = H(s);
    }
    *sigma = expf(h2[0]);  // trunc_exp forward, nerf/activation.py:8-10
    float cin[31];
    sh_one(dir, 4, cin);
    for (int k = 0; k < 16; k++) cin[k] = H(cin[k]);
    for (int k = 0; k < 15; k++) cin[16 + k] = h2[1 + k];
    float c1[64], c2[64];
    for (int j = 0; j < 64; j++) {
        float s = 0;
        for (int k = 0; k < 31; k++) s += nt.W2[j * 31 + k] * cin[k];
        s = H(s);
        c1[j] = s > 0 ? s : 0;
    }
    for (int j = 0; j < 64; j++) {
        float s = 0;
        for (int k = 0; k < 64; k++) s += nt.W3[j * 64 + k] * c1[k];
        s = H(s);
        c2[j] = s > 0 ? s : 0;
    }
    for (int j = 0; j < 3; j++) {
        float s = 0;
        for (int k = 0; k < 64; k++) s += nt.W4[j * 64 + k] * c2[k];
        s = H(s);
        rgb[j] = H(1.0f / (1.0f + expf(-s)));  // torch.sigmoid
    }
}

}  // namespace

extern "C" {

// raymarching.cu:165-202
void orc_sph_from_ray(const float* rays_o, const float* rays_d, float radius, uint32_t N, float* coords) {
    for (int64_t n = 0; n < (int64_t)N; n++) {
        const float ox = rays_o[n * 3], oy = rays_o[n * 3 + 1], oz = rays_o[n * 3 + 2];
        const float dx = rays_d[n * 3], dy = rays_d[n * 3 + 1], dz = rays_d[n * 3 + 2];
        const float A = dx * dx + dy * dy + dz * dz;
        const float B = ox * dx + oy * dy + oz * dz;
        const float Cq = ox * ox + oy * oy + oz * oz - radius * radius;
        const float t = (-B + sqrtf(B * B - A * Cq)) / A;
        const float x = ox + t * dx, y = oy + t * dy, z = oz + t * dz;
        const float theta = atan2f(sqrtf(x * x + z * z), y);
        const float phi = atan2f(z, x);
        const float rpi = 0.3183098861837907f;
        coords[n * 2] = 2 * theta * rpi - 1;
        coords[n * 2 + 1] = phi * rpi;
    }
}

// raymarching.cu:91-159
void orc_near_far_from_aabb(const float* rays_o, const float* rays_d, const float* aabb, uint32_t N, float min_near, float* nears, float* fars) {
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < (int64_t)N; n++) {
        const float ox = rays_o[n * 3], oy = rays_o[n * 3 + 1], oz = rays_o[n * 3 + 2];
        const float dx = rays_d[n * 3], dy = rays_d[n * 3 + 1], dz = rays_d[n * 3 + 2];
        const float rdx = 1 / dx, rdy = 1 / dy, rdz = 1 / dz;
        float near = (aabb[0] - ox) * rdx, far = (aabb[3] - ox) * rdx;
        if (near > far) std::swap(near, far);
        float near_y = (aabb[1] - oy) * rdy, far_y = (aabb[4] - oy) * rdy;
        if (near_y > far_y) std::swap(near_y, far_y);
        if (near > far_y || near_y > far) { nears[n] = fars[n] = FLT_MAX; continue; }
        if (near_y > near) near = near_y;
        if (far_y < far) far = far_y;
        float near_z = (aabb[2] - oz) * rdz, far_z = (aabb[5] - oz) * rdz;
        if (near_z > far_z) std::swap(near_z, far_z);
        if (near > far_z || near_z > far) { nears[n] = fars[n] = FLT_MAX; continue; }
        if (near_z > near) near = near_z;
        if (far_z < far) far = far_z;
        if (near < min_near) near = min_near;
        nears[n] = near;
        fars[n] = far;
    }
}

// nerf/utils.py:355-443 (get_pnts_in_grids + p2g).  The reference fills slots
// in atomic-race order; this restatement (and the HIP kernel) use ascending IP
// id inside a cell — the deterministic member of the reference's outcome set.
// Returns the number of points whose cell id fell outside [0, n_grid).
int orc_pnts_in_grids(int n_vtx, int n_grid, const float* pnts, const float* bbmin, float hgs, const int* resolution, int* pig_cnt, int* pig_bgn,
                      int* pig_idx) {
    std::vector<int> gid(n_vtx);
    int bad = 0;
    std::fill(pig_cnt, pig_cnt + n_grid, 0);
    for (int p = 0; p < n_vtx; p++) {
        const int g0 = (int)floorf((pnts[p * 3] - bbmin[0]) / hgs);
        const int g1 = (int)floorf((pnts[p * 3 + 1] - bbmin[1]) / hgs);
        const int g2 = (int)floorf((pnts[p * 3 + 2] - bbmin[2]) / hgs);
        int g = g2 * resolution[1] * resolution[0] + g1 * resolution[0] + g0;
        if (g < 0 || g >= n_grid) { bad++; g = -1; }
        gid[p] = g;
        if (g >= 0) pig_cnt[g]++;
    }
    int run = 0;
    for (int g = 0; g < n_grid; g++) { pig_bgn[g] = run; run += pig_cnt[g]; }
    std::vector<int> fill(n_grid, 0);
    for (int p = 0; p < n_vtx; p++) if (gid[p] >= 0) pig_idx[pig_bgn[gid[p]] + fill[gid[p]]++] = p;
    return bad;
}

// raymarching.cu:1436-1489 (host) + 1121-1434 (kernel).  Argument order is the
// pybind order (p_def, p_ori).  xyzs/dirs/deltas must be zero-filled by the
// caller (raymarching/raymarching.py:415-417).  Returns 1 if any sample's
// search cell fell outside the grid (reference prints "ERROR").
int orc_march_rays_quadratic_bending(const int* pig_cnt, const int* pig_bgn, const int* pig_idx, int n_vtx, int n_grid, const float* p_def,
                                     const float* p_ori, const float* F_IP, const float* dF_IP, int max_iter_num, const float* bbmin,
                                     const float* bbmax, float hgs, const int* resolution, int num_seek_IP, float IP_dx, int cut,
                                     const float* cut_bounds, uint32_t n_alive, uint32_t n_step, const int* rays_alive, const float* rays_t,
                                     const float* rays_o, const float* rays_d, float bound, float dt_gamma, uint32_t max_steps, uint32_t C,
                                     uint32_t H, const uint8_t* grid, const float* nears, const float* fars, float* xyzs, float* dirs, float* deltas,
                                     const float* noises) {
    (void)nears;
    MarchArgs a;
    a.pg = Pig{pig_cnt, pig_bgn, pig_idx, resolution, n_grid};
    a.n_vtx = n_vtx;
    a.p_ori = p_ori; a.p_def = p_def; a.F_IP = F_IP; a.dF_IP = dF_IP;
    a.max_iter_num = max_iter_num;
    a.bbmin = bbmin; a.bbmax = bbmax; a.hgs = hgs;
    a.num_seek_IP = num_seek_IP; a.IP_dx = IP_dx;
    a.cut = cut != 0; a.cut_bounds = cut_bounds;
    a.n_step = n_step; a.rays_t = rays_t; a.rays_o = rays_o; a.rays_d = rays_d;
    a.bound = bound; a.dt_gamma = dt_gamma; a.max_steps = max_steps; a.C = C; a.H = H;
    a.grid = grid; a.fars = fars;
    int any_oob = 0;
#pragma omp parallel for schedule(dynamic, 256) reduction(| : any_oob)
    for (int64_t n = 0; n < (int64_t)n_alive; n++) {
        int oob = 0;
        march_one(a, rays_alive[n], noises[n], xyzs + (size_t)n * n_step * 3, dirs + (size_t)n * n_step * 3, deltas + (size_t)n * n_step * 2, &oob);
        any_oob |= oob;
    }
    if (g_march_stats) {
        fprintf(stderr, "[march] n_alive=%u n_step=%u iters=%lld found=%lld max/ray=%d hist(log2):", n_alive, n_step, g_march_iters, g_march_found, g_march_max);
        for (int b = 0; b < 12; b++) fprintf(stderr, " %d", g_march_hist[b]);
        fprintf(stderr, "\n");
        g_march_iters = g_march_found = 0; g_march_max = 0; std::fill(g_march_hist.begin(), g_march_hist.end(), 0);
    }
    return any_oob;
}

// raymarching.cu:925-932 (host) + 827-923 (kernel)
// march_rays (raymarching.cu:703-824): the undeformed march.  Same loop as march_one without the search / warp.
void orc_march_rays(uint32_t n_alive, uint32_t n_step, const int* rays_alive, const float* rays_t, const float* rays_o, const float* rays_d,
                    float bound, float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H, const uint8_t* grid, const float* fars, float* xyzs_,
                    float* dirs_, float* deltas_, const float* noises) {
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t n = 0; n < (int64_t)n_alive; n++) {
        const int index = rays_alive[n];
        const float noise = noises ? noises[n] : 0.0f;
        const float* ro = rays_o + (size_t)index * 3;
        const float* rd = rays_d + (size_t)index * 3;
        float* xyzs = xyzs_ + (size_t)n * n_step * 3;
        float* dirs = dirs_ + (size_t)n * n_step * 3;
        float* deltas = deltas_ + (size_t)n * n_step * 2;
        const float ox = ro[0], oy = ro[1], oz = ro[2];
        const float dx = rd[0], dy = rd[1], dz = rd[2];
        const float rdx = 1 / dx, rdy = 1 / dy, rdz = 1 / dz;
        const float rH = 1 / (float)H;
        const float H3 = (float)(H * H * H);
        float t = rays_t[index];
        const float far = fars[index];
        const float dt_min = 2 * 1.73205080757f / max_steps;
        const float dt_max = 2 * 1.73205080757f * (1 << (C - 1)) / H;
        uint32_t step = 0;
        t += clampf(t * dt_gamma, dt_min, dt_max) * noise;
        float last_t = t;
        while (t < far && step < n_step) {
            const float x = clampf(ox + t * dx, -bound, bound);
            const float y = clampf(oy + t * dy, -bound, bound);
            const float z = clampf(oz + t * dz, -bound, bound);
            const float dt = clampf(t * dt_gamma, dt_min, dt_max);
            const int level = std::max(mip_from_pos(x, y, z, (float)C), mip_from_dt(dt, (float)H, (float)C));
            const float mip_bound = fminf(scalbnf(1, level), bound);
            const float mip_rbound = 1 / mip_bound;
            const int nx = (int)clampf((float)(0.5 * (double)(x * mip_rbound + 1) * (double)H), 0.0f, (float)(H - 1));
            const int ny = (int)clampf((float)(0.5 * (double)(y * mip_rbound + 1) * (double)H), 0.0f, (float)(H - 1));
            const int nz = (int)clampf((float)(0.5 * (double)(z * mip_rbound + 1) * (double)H), 0.0f, (float)(H - 1));
            const uint32_t vox = (uint32_t)(level * H3 + (float)morton3D(nx, ny, nz));
            const bool occ = grid[vox / 8] & (1 << (vox % 8));
            if (occ) {
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
                do { t += clampf(t * dt_gamma, dt_min, dt_max); } while (t < tt);
            }
        }
    }
}

// packbits (raymarching.cu:270-303), morton3D / morton3D_invert (:60-81,217-263)
void orc_packbits(const float* grid, uint32_t N, float density_thresh, uint8_t* bitfield) {
    for (uint32_t n = 0; n < N; n++) {
        uint8_t bits = 0;
        for (int i = 0; i < 8; i++) bits |= (grid[(size_t)n * 8 + i] > density_thresh) ? (uint8_t)(1u << i) : 0;
        bitfield[n] = bits;
    }
}
void orc_morton3D(const int* coords, uint32_t N, int* indices) {
    for (uint32_t n = 0; n < N; n++) indices[n] = (int)morton3D((uint32_t)coords[n * 3], (uint32_t)coords[n * 3 + 1], (uint32_t)coords[n * 3 + 2]);
}
void orc_morton3D_invert(const int* indices, uint32_t N, int* coords) {
    for (uint32_t n = 0; n < N; n++)
        for (int a = 0; a < 3; a++) {
            uint32_t x = (uint32_t)(indices[n] >> a);
            x = x & 0x49249249u;
            x = (x | (x >> 2)) & 0xc30c30c3u;
            x = (x | (x >> 4)) & 0x0f00f00fu;
            x = (x | (x >> 8)) & 0xff0000ffu;
            x = (x | (x >> 16)) & 0x0000ffffu;
            coords[n * 3 + a] = (int)x;
        }
}

void orc_composite_rays(uint32_t n_alive, uint32_t n_step, float T_thresh, int* rays_alive, float* rays_t, const float* sigmas, const float* rgbs,
                        const float* deltas, float* weights_sum, float* depth, float* image) {
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < (int64_t)n_alive; n++)
        composite_one((uint32_t)n, n_step, T_thresh, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image);
}

// nerf/renderer.py:887  rays_alive = rays_alive[rays_alive >= 0]  (stable filter)
int orc_compact_rays(const int* rays_alive, int n, int* out) {
    int m = 0;
    for (int i = 0; i < n; i++) if (rays_alive[i] >= 0) out[m++] = rays_alive[i];
    return m;
}

// gridencoder.cu:371-400 launch shape; outputs [L,B,C] like the reference kernel.
void orc_grid_encode_forward(const float* inputs, const float* embeddings, const int* offsets, float* outputs, uint32_t B, uint32_t D, uint32_t C,
                             uint32_t L, float S, uint32_t H, uint32_t gridtype, int align_corners, uint32_t interp) {
    if (D != 3 || C > 8) return;
    for (uint32_t l = 0; l < L; l++) {
        float scale; uint32_t res;
        level_params(l, S, H, &scale, &res);
        if (align_corners) {}
        const uint32_t hs = (uint32_t)(offsets[l + 1] - offsets[l]);
        const float* table = embeddings + (size_t)(uint32_t)offsets[l] * C;
#pragma omp parallel for schedule(static)
        for (int64_t b = 0; b < (int64_t)B; b++)
            grid_one(inputs + b * 3, table, hs, scale, res, C, gridtype, align_corners != 0, interp, outputs + ((size_t)l * B + b) * C);
    }
}

// per-level scale/resolution table (gridencoder.cu:133-134), exposed so tests can
// pin what the HIP launcher precomputes on the host.
void orc_grid_level_params(uint32_t L, float S, uint32_t H, float* scales, uint32_t* resolutions) {
    for (uint32_t l = 0; l < L; l++) level_params(l, S, H, scales + l, resolutions + l);
}

// shencoder.cu:384-390
void orc_sh_encode_forward(const float* inputs, float* outputs, uint32_t B, uint32_t D, uint32_t C) {
    if (D != 3 || C > 8) return;
    for (int64_t b = 0; b < (int64_t)B; b++) {
        sh_one(inputs + b * 3, C < 4 ? C : 4, outputs + b * C * C);
        if (C > 4) sh_high_bands(inputs + b * 3, C, outputs + b * C * C, nullptr, nullptr, nullptr);
    }
}

// NeRFNetwork.forward over M samples (nerf/network.py:98-127), density_scale applied by the caller.
void orc_nerf_forward(const float* xyzs, const float* dirs, uint32_t M, float bound, const float* embeddings, const int* offsets, uint32_t L,
                      uint32_t Cf, float S, uint32_t Hbase, const float* W0, const float* W1, const float* W2, const float* W3, const float* W4,
                      float* sigmas, float* rgbs) {
    Net base{embeddings, offsets, L, Cf, Hbase, S, bound, W0, W1, W2, W3, W4};
    HalfNet* hn = g_half ? new HalfNet(base, (size_t)(uint32_t)offsets[L] * Cf) : nullptr;
    const Net& nt = hn ? hn->net : base;
#pragma omp parallel for schedule(static)
    for (int64_t m = 0; m < (int64_t)M; m++) nerf_one(nt, xyzs + m * 3, dirs + m * 3, sigmas + m, rgbs + m * 3);
    delete hn;
}

// orc_set_half(1): orc_nerf_forward / orc_render_deformed restate the network as the reference runs it under autocast with fp16 (see
// nerf_one / grid_one); returns the previous setting.
int orc_set_half(int on) { const int prev = g_half; g_half = on ? 1 : 0; return prev; }

// [host] test hook: out[i] = in[i] rounded to the nearest fp16 value
void orc_hround(const float* in, float* out, uint32_t n) { for (uint32_t i = 0; i < n; i++) out[i] = hround(in[i]); }

// kernel_grid<at::Half,3,C> over B samples: `embeddings` fp32 master (rounded to half here, grid.py:44), outputs [L,B,C] = half values as floats
void orc_grid_encode_forward_half(const float* inputs, const float* embeddings, const int* offsets, float* outputs, uint32_t B, uint32_t C, uint32_t L,
                                  float S, uint32_t H, uint32_t gridtype, int align_corners, uint32_t interp) {
    if (C > 8) return;
    std::vector<float> tab((size_t)(uint32_t)offsets[L] * C);
    for (size_t i = 0; i < tab.size(); i++) tab[i] = hround(embeddings[i]);
    for (uint32_t l = 0; l < L; l++) {
        float scale; uint32_t res;
        level_params(l, S, H, &scale, &res);
        const uint32_t hs = (uint32_t)(offsets[l + 1] - offsets[l]);
        const float* table = tab.data() + (size_t)(uint32_t)offsets[l] * C;
#pragma omp parallel for schedule(static)
        for (int64_t b = 0; b < (int64_t)B; b++)
            grid_one(inputs + b * 3, table, hs, scale, res, C, gridtype, align_corners != 0, interp, outputs + ((size_t)l * B + b) * C, true);
    }
}

// nerf/utils.py:54-138 (N = -1 path): pixel centres, row-major; d = normalize(...) @ R^T; o = t.
// pose: row-major 4x4 cam2world.
void orc_get_rays(const float* pose, float fx, float fy, float cx, float cy, int H, int W, float* rays_o, float* rays_d) {
#pragma omp parallel for schedule(static)
    for (int64_t p = 0; p < (int64_t)H * W; p++) {
        const float i = (float)(p % W) + 0.5f, j = (float)(p / W) + 0.5f;
        const float xs = (i - cx) / fx, ys = (j - cy) / fy, zs = 1.0f;
        const float nrm = sqrtf(xs * xs + ys * ys + zs * zs);
        const float d0 = xs / nrm, d1 = ys / nrm, d2 = zs / nrm;
        for (int c = 0; c < 3; c++) {
            rays_d[p * 3 + c] = d0 * pose[c * 4 + 0] + d1 * pose[c * 4 + 1] + d2 * pose[c * 4 + 2];
            rays_o[p * 3 + c] = pose[c * 4 + 3];
        }
    }
}

// NeRFRenderer.rund_cuda, nerf/renderer.py:755-907, given bbmin/bbmax/resolution
// (computed by the caller as :782-791 does) and perturb=False.  bg_color is a
// scalar (reference default 1).  Returns the number of loop trips; per-trip
// (n_alive, n_step) and total samples are reported through stats[] =
// {trips, total_emitted_samples, total_mlp_slots}.
int orc_render_deformed(const float* rays_o, const float* rays_d, uint32_t N, const float* p_def, const float* p_ori, const float* F_IP,
                        const float* dF_IP, int n_vtx, const float* bbmin, const float* bbmax, const int* resolution, float hgs,
                        int max_iter_num, int num_seek_IP, float IP_dx, int cut, const float* cut_bounds, float bound, float min_near,
                        float dt_gamma, uint32_t max_steps, float T_thresh, uint32_t C, uint32_t Hgrid, const uint8_t* bitfield,
                        float density_scale, float bg_color, const float* embeddings, const int* offsets, uint32_t L, uint32_t Cf, float S,
                        uint32_t Hbase, const float* W0, const float* W1, const float* W2, const float* W3, const float* W4, float* image,
                        float* depth, float* depth_0, float* weights_sum, int64_t* stats) {
    const int n_grid = resolution[2] * resolution[1] * resolution[0];
    std::vector<int> pig_cnt(n_grid), pig_bgn(n_grid), pig_idx(n_vtx);
    orc_pnts_in_grids(n_vtx, n_grid, p_def, bbmin, hgs, resolution, pig_cnt.data(), pig_bgn.data(), pig_idx.data());
    float aabb[6] = {bbmin[0], bbmin[1], bbmin[2], bbmax[0], bbmax[1], bbmax[2]};
    std::vector<float> nears(N), fars(N), rays_t(N);
    orc_near_far_from_aabb(rays_o, rays_d, aabb, N, min_near, nears.data(), fars.data());
    std::fill(weights_sum, weights_sum + N, 0.0f);
    std::fill(depth_0, depth_0 + N, 0.0f);
    std::fill(image, image + (size_t)N * 3, 0.0f);
    std::vector<int> alive(N), alive2(N);
    for (uint32_t i = 0; i < N; i++) alive[i] = (int)i;
    rays_t = nears;
    Net base{embeddings, offsets, L, Cf, Hbase, S, bound, W0, W1, W2, W3, W4};
    HalfNet* hn = g_half ? new HalfNet(base, (size_t)(uint32_t)offsets[L] * Cf) : nullptr;  // autocast: half tables / weights
    const Net& nt = hn ? hn->net : base;
    uint32_t step = 0;
    int trips = 0;
    int64_t emitted = 0, slots = 0;
    int n_alive = (int)N;
    std::vector<float> xyzs, dirs, deltas, sigmas, rgbs, noises;
    while (step < max_steps) {
        if (n_alive <= 0) break;
        const uint32_t n_step = (uint32_t)std::max(std::min((int)(N / (uint32_t)n_alive), 8), 1);
        size_t M = (size_t)n_alive * n_step;
        M += 128 - (M % 128);  // raymarching/raymarching.py:410-413
        xyzs.assign(M * 3, 0.0f); dirs.assign(M * 3, 0.0f); deltas.assign(M * 2, 0.0f);
        sigmas.resize(M); rgbs.resize(M * 3);
        noises.assign(n_alive, 0.0f);
        orc_march_rays_quadratic_bending(pig_cnt.data(), pig_bgn.data(), pig_idx.data(), n_vtx, n_grid, p_def, p_ori, F_IP, dF_IP, max_iter_num, bbmin,
                                         bbmax, hgs, resolution, num_seek_IP, IP_dx, cut, cut_bounds, (uint32_t)n_alive, n_step, alive.data(),
                                         rays_t.data(), rays_o, rays_d, bound, dt_gamma, max_steps, C, Hgrid, bitfield, nears.data(), fars.data(),
                                         xyzs.data(), dirs.data(), deltas.data(), noises.data());
        // The reference evaluates the network on all M slots (renderer.py:874).  Slots whose
        // delta is 0 are never read by composite (cu:867), so the oracle skips them; their
        // sigma/rgb are left at 0.
        int64_t em = 0;
#pragma omp parallel for schedule(dynamic, 64) reduction(+ : em)
        for (int64_t m = 0; m < (int64_t)M; m++) {
            if (deltas[m * 2] == 0) { sigmas[m] = 0; rgbs[m * 3] = rgbs[m * 3 + 1] = rgbs[m * 3 + 2] = 0; continue; }
            nerf_one(nt, &xyzs[m * 3], &dirs[m * 3], &sigmas[m], &rgbs[m * 3]);
            sigmas[m] = density_scale * sigmas[m];
            em++;
        }
        emitted += em;
        slots += (int64_t)M;
        orc_composite_rays((uint32_t)n_alive, n_step, T_thresh, alive.data(), rays_t.data(), sigmas.data(), rgbs.data(), deltas.data(), weights_sum,
                           depth_0, image);
        n_alive = orc_compact_rays(alive.data(), n_alive, alive2.data());
        alive.swap(alive2);
        step += n_step;
        trips++;
    }
    // renderer.py:896-899
    for (uint32_t i = 0; i < N; i++) {
        for (int c = 0; c < 3; c++) image[i * 3 + c] = image[i * 3 + c] + (1 - weights_sum[i]) * bg_color;
        depth[i] = fmaxf(depth_0[i] - nears[i], 0.0f) / (fars[i] - nears[i]);
    }
    if (stats) { stats[0] = trips; stats[1] = emitted; stats[2] = slots; }
    delete hn;
    return trips;
}

// Test hook: the per-IP Newton inverse warp used by the march (same function), for one deformed point and one IP.
// out[0..2] = rest-space point; returns 1 when the IP would be rejected (|p - p_ori|_inf > IP_dx).
int orc_warp_point(const float* x3, const float* p_ori, const float* p_def, const float* F9, const float* dF27, int max_iter_num, float IP_dx,
                   float* out) {
    return newton_warp(x3, p_ori, p_def, F9, dF27, max_iter_num, IP_dx, out) ? 1 : 0;
}

int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

}  // extern "C"

// =====================================================================================================================
// Training-side ops — same status as everything above: test infrastructure, parity unpinned.
// =====================================================================================================================
namespace {

// The loop shared by kernel_march_rays_train's two passes (raymarching.cu:357-401 counts, :420-480 writes).  WRITE = false:
// returns the number of occupied steps up to `limit`; WRITE = true: also stores xyzs / dirs / deltas.
template <bool WRITE>
uint32_t train_march_pass(const float* ro, const float* rd, float t0, float far, uint32_t limit, float bound, float dt_gamma, uint32_t max_steps,
                          uint32_t C, uint32_t H, const uint8_t* grid, float* xyzs, float* dirs, float* deltas) {
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
        const int level = std::max(mip_from_pos(x, y, z, (float)C), mip_from_dt(dt, (float)H, (float)C));
        const float mip_bound = fminf(scalbnf(1.0f, level), bound);
        const float mip_rbound = 1 / mip_bound;
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

// Bands l = 4..7 (degree 5-8; shencoder.cu:69-123 lists them as expanded polynomials, :125-355 their derivatives).  Restated from the
// definition instead of the list, in double, narrowed at the end:
//   Y_l^m = (-1)^m sqrt2 K_l^|m| T_l^|m|(z) * { A_|m|(x, y) for m > 0, B_|m|(x, y) for m < 0 },   Y_l^0 = K_l^0 T_l^0(z),
//   K_l^m = sqrt((2l + 1) / (4 pi) * (l - m)! / (l + m)!),  A_m + i B_m = (x + i y)^m,  T_l^m = d^m/dz^m P_l(z)  (no Condon-Shortley
//   factor inside T; the basis as a whole KEEPS that phase: Y_1^{-1} = -c y, which is what the reference's list has),
// with the closed sums  P_l(z) = 2^-l sum_k (-1)^k C(l, k) C(2l - 2k, l) z^(l - 2k)  and the binomial expansion of (x + i y)^m — a formulation
// independent of the recurrences the HIP kernel uses.  As polynomials in (x, y, z) these ARE the reference's expressions (T in z only, A / B in
// x, y only), so they agree off the unit sphere too, and so do their partial derivatives.
static double binom(int n, int k) {
    if (k < 0 || k > n) return 0.0;
    double r = 1.0;
    for (int i = 1; i <= k; i++) r = r * (double)(n - k + i) / (double)i;
    return std::round(r);
}
static double fact(int n) { double r = 1.0; for (int i = 2; i <= n; i++) r *= i; return r; }
// T_l^m(z) and its z-derivative
static void legendre_deriv(int l, int m, double z, double* T, double* dT) {
    double t = 0.0, dt = 0.0;
    for (int k = 0; 2 * k <= l; k++) {
        const int n = l - 2 * k;           // power of z in P_l
        if (n < m) break;
        const double c = ((k & 1) ? -1.0 : 1.0) * binom(l, k) * binom(2 * l - 2 * k, l) / std::ldexp(1.0, l) * fact(n) / fact(n - m);
        t += c * std::pow(z, n - m);
        if (n - m >= 1) dt += c * (n - m) * std::pow(z, n - m - 1);
    }
    *T = t; *dT = dt;
}
// A_m, B_m and their x / y derivatives
static void azimuth_poly(int m, double x, double y, double* A, double* B, double* Ax, double* Ay, double* Bx, double* By) {
    double a = 0, b = 0, ax = 0, ay = 0, bx = 0, by = 0;
    for (int j = 0; j <= m; j++) {
        const double c = binom(m, j) * ((j / 2) & 1 ? -1.0 : 1.0);     // i^j = (+1, +i, -1, -i)
        const double mono = std::pow(x, m - j) * std::pow(y, j);
        const double dmx = (m - j >= 1) ? (m - j) * std::pow(x, m - j - 1) * std::pow(y, j) : 0.0;
        const double dmy = (j >= 1) ? j * std::pow(x, m - j) * std::pow(y, j - 1) : 0.0;
        if (j & 1) { b += c * mono; bx += c * dmx; by += c * dmy; }
        else { a += c * mono; ax += c * dmx; ay += c * dmy; }
    }
    *A = a; *B = b; *Ax = ax; *Ay = ay; *Bx = bx; *By = by;
}
void sh_high_bands(const float* in, uint32_t C, float* out, float* gx, float* gy, float* gz) {
    static const double PI_ = 3.14159265358979323846;
    const double x = in[0], y = in[1], z = in[2];
    for (int l = 4; l < (int)C; l++)
        for (int m = 0; m <= l; m++) {
            double T, dT, A, B, Ax, Ay, Bx, By;
            legendre_deriv(l, m, z, &T, &dT);
            azimuth_poly(m, x, y, &A, &B, &Ax, &Ay, &Bx, &By);
            const double K = std::sqrt((2.0 * l + 1.0) / (4.0 * PI_) * fact(l - m) / fact(l + m)) * (m ? std::sqrt(2.0) * ((m & 1) ? -1.0 : 1.0) : 1.0);
            const int ip = l * l + l + m, im = l * l + l - m;
            if (out) { out[ip] = (float)(K * T * A); if (m) out[im] = (float)(K * T * B); }
            if (gx) {
                gx[ip] = (float)(K * T * Ax); gy[ip] = (float)(K * T * Ay); gz[ip] = (float)(K * dT * A);
                if (m) { gx[im] = (float)(K * T * Bx); gy[im] = (float)(K * T * By); gz[im] = (float)(K * dT * B); }
            }
        }
}

// first-order terms of sh_one (d/dx, d/dy, d/dz of the same polynomials; shencoder.cu:125-355 tabulates the same derivatives)
void sh_one_grad(const float* in, uint32_t C, float* gx, float* gy, float* gz) {
    static const double PI_ = 3.14159265358979323846;  // the same closed forms as sh_one
    static const float c1 = (float)(std::sqrt(3.0) / (2.0 * std::sqrt(PI_)));
    static const float c2a = (float)(std::sqrt(15.0) / (2.0 * std::sqrt(PI_)));
    static const float c2b = (float)(3.0 * std::sqrt(5.0) / (4.0 * std::sqrt(PI_)));
    static const float c2d = (float)(std::sqrt(15.0) / (4.0 * std::sqrt(PI_)));
    static const float c3a = (float)(std::sqrt(70.0) / (8.0 * std::sqrt(PI_)));
    static const float c3b = (float)(std::sqrt(105.0) / (2.0 * std::sqrt(PI_)));
    static const float c3c = (float)(std::sqrt(42.0) / (8.0 * std::sqrt(PI_)));
    static const float c3d = (float)(std::sqrt(7.0) / (4.0 * std::sqrt(PI_)));
    static const float c3e = (float)(std::sqrt(105.0) / (4.0 * std::sqrt(PI_)));
    const float x = in[0], y = in[1], z = in[2];
    const float xy = x * y, xz = x * z, yz = y * z, x2 = x * x, y2 = y * y, z2 = z * z;
    const uint32_t C2 = C * C;
    for (uint32_t i = 0; i < C2; i++) gx[i] = gy[i] = gz[i] = 0.0f;
    if (C <= 1) return;
    gy[1] = -c1; gz[2] = c1; gx[3] = -c1;
    if (C <= 2) return;
    gx[4] = c2a * y; gy[4] = c2a * x;
    gy[5] = -c2a * z; gz[5] = -c2a * y;
    gz[6] = 2.0f * c2b * z;
    gx[7] = -c2a * z; gz[7] = -c2a * x;
    gx[8] = 2.0f * c2d * x; gy[8] = -2.0f * c2d * y;
    if (C <= 3) return;
    gx[9] = -6.0f * c3a * xy; gy[9] = c3a * (-3.0f * x2 + 3.0f * y2);
    gx[10] = c3b * yz; gy[10] = c3b * xz; gz[10] = c3b * xy;
    gy[11] = c3c * (1.0f - 5.0f * z2); gz[11] = -10.0f * c3c * yz;
    gz[12] = c3d * (15.0f * z2 - 3.0f);
    gx[13] = c3c * (1.0f - 5.0f * z2); gz[13] = -10.0f * c3c * xz;
    gx[14] = 2.0f * c3e * xz; gy[14] = -2.0f * c3e * yz; gz[14] = c3e * (x2 - y2);
    gx[15] = c3a * (-3.0f * x2 + 3.0f * y2); gy[15] = 6.0f * c3a * xy;
}

}  // namespace

extern "C" {

// march_rays_train (raymarching.cu:314-497).  The reference hands out point ranges and ray rows with two atomicAdd counters, so
// its ray order is a race; here (and in the HIP path) rays keep their own order: rays[n] = (n, exclusive prefix of the counts, count),
// counter += (total points, N).  Rays whose range would pass M are dropped exactly like the reference's (:415).
void orc_march_rays_train(const float* rays_o, const float* rays_d, const uint8_t* grid, float bound, float dt_gamma, uint32_t max_steps, uint32_t N,
                          uint32_t C, uint32_t H, uint32_t M, const float* nears, const float* fars, float* xyzs, float* dirs, float* deltas,
                          int* rays, int* counter, const float* noises) {
    const float dt_min = 2 * 1.73205080757f / max_steps;
    const float dt_max = 2 * 1.73205080757f * (1 << (C - 1)) / H;
    std::vector<uint32_t> cnt(N);
    std::vector<float> t0s(N);
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t n = 0; n < (int64_t)N; n++) {
        float t0 = nears[n];
        t0 += clampf(t0 * dt_gamma, dt_min, dt_max) * (noises ? noises[n] : 0.0f);
        t0s[n] = t0;
        cnt[n] = train_march_pass<false>(rays_o + n * 3, rays_d + n * 3, t0, fars[n], max_steps, bound, dt_gamma, max_steps, C, H, grid, nullptr, nullptr,
                                         nullptr);
    }
    uint32_t point = (uint32_t)counter[0];
    const uint32_t ray0 = (uint32_t)counter[1];
    for (uint32_t n = 0; n < N; n++) {
        rays[(size_t)(ray0 + n) * 3] = (int)n;
        rays[(size_t)(ray0 + n) * 3 + 1] = (int)point;
        rays[(size_t)(ray0 + n) * 3 + 2] = (int)cnt[n];
        point += cnt[n];
    }
    counter[0] = (int)point;
    counter[1] = (int)(ray0 + N);
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t n = 0; n < (int64_t)N; n++) {
        const uint32_t off = (uint32_t)rays[(size_t)(ray0 + n) * 3 + 1];
        if (cnt[n] == 0 || off + cnt[n] > M) continue;
        train_march_pass<true>(rays_o + n * 3, rays_d + n * 3, t0s[n], fars[n], cnt[n], bound, dt_gamma, max_steps, C, H, grid, xyzs + (size_t)off * 3,
                               dirs + (size_t)off * 3, deltas + (size_t)off * 2);
    }
}

// composite_rays_train_forward (raymarching.cu:503-581)
void orc_composite_rays_train_forward(const float* sigmas, const float* rgbs, const float* deltas, const int* rays, uint32_t M, uint32_t N,
                                      float T_thresh, float* weights_sum, float* depth, float* image) {
    for (uint32_t n = 0; n < N; n++) {
        const uint32_t index = rays[n * 3], offset = rays[n * 3 + 1], num_steps = rays[n * 3 + 2];
        if (num_steps == 0 || offset + num_steps > M) {
            weights_sum[index] = 0; depth[index] = 0;
            image[index * 3] = image[index * 3 + 1] = image[index * 3 + 2] = 0;
            continue;
        }
        const float* s = sigmas + offset; const float* c = rgbs + (size_t)offset * 3; const float* dl = deltas + (size_t)offset * 2;
        uint32_t step = 0;
        float T = 1.0f, r = 0, g = 0, b = 0, ws = 0, t = 0, d = 0;
        while (step < num_steps) {
            const float alpha = 1.0f - expf(-s[0] * dl[0]);
            const float weight = alpha * T;
            r += weight * c[0]; g += weight * c[1]; b += weight * c[2];
            t += dl[1];
            d += weight * t;
            ws += weight;
            T *= 1.0f - alpha;
            if (T < T_thresh) break;
            s++; c += 3; dl += 2; step++;
        }
        weights_sum[index] = ws; depth[index] = d;
        image[index * 3] = r; image[index * 3 + 1] = g; image[index * 3 + 2] = b;
    }
}

// composite_rays_train_backward (raymarching.cu:604-686); grad_sigmas / grad_rgbs zero-filled by the caller
void orc_composite_rays_train_backward(const float* grad_weights_sum, const float* grad_image, const float* sigmas, const float* rgbs,
                                       const float* deltas, const int* rays, const float* weights_sum, const float* image, uint32_t M, uint32_t N,
                                       float T_thresh, float* grad_sigmas, float* grad_rgbs) {
    for (uint32_t n = 0; n < N; n++) {
        const uint32_t index = rays[n * 3], offset = rays[n * 3 + 1], num_steps = rays[n * 3 + 2];
        if (num_steps == 0 || offset + num_steps > M) continue;
        const float gws = grad_weights_sum[index];
        const float* gi = grad_image + (size_t)index * 3;
        const float r_final = image[index * 3], g_final = image[index * 3 + 1], b_final = image[index * 3 + 2], ws_final = weights_sum[index];
        const float* s = sigmas + offset; const float* c = rgbs + (size_t)offset * 3; const float* dl = deltas + (size_t)offset * 2;
        float* gs = grad_sigmas + offset; float* gc = grad_rgbs + (size_t)offset * 3;
        uint32_t step = 0;
        float T = 1.0f, r = 0, g = 0, b = 0, ws = 0;
        while (step < num_steps) {
            const float alpha = 1.0f - expf(-s[0] * dl[0]);
            const float weight = alpha * T;
            r += weight * c[0]; g += weight * c[1]; b += weight * c[2];
            ws += weight;
            T *= 1.0f - alpha;
            gc[0] = gi[0] * weight; gc[1] = gi[1] * weight; gc[2] = gi[2] * weight;
            gs[0] = dl[0] * (gi[0] * (T * c[0] - (r_final - r)) + gi[1] * (T * c[1] - (g_final - g)) + gi[2] * (T * c[2] - (b_final - b)) +
                             gws * (1 - ws_final));
            if (T < T_thresh) break;
            s++; c += 3; dl += 2; gs++; gc += 3; step++;
        }
    }
}

// kernel_grid's dy_dx branch (gridencoder.cu:199-243): dy_dx [B, L, 3, C]
void orc_grid_encode_dy_dx(const float* inputs, const float* embeddings, const int* offsets, float* dy_dx, uint32_t B, uint32_t C, uint32_t L, float S,
                           uint32_t H, uint32_t gridtype, int align_corners, uint32_t interp) {
    for (uint32_t l = 0; l < L; l++) {
        float scale; uint32_t res;
        level_params(l, S, H, &scale, &res);
        const uint32_t hs = (uint32_t)(offsets[l + 1] - offsets[l]);
        const float* table = embeddings + (size_t)(uint32_t)offsets[l] * C;
        for (uint32_t b = 0; b < B; b++) {
            const float* in3 = inputs + (size_t)b * 3;
            float* out = dy_dx + ((size_t)b * L + l) * 3 * C;
            bool oob = false;
            for (int d = 0; d < 3; d++) if (in3[d] < 0 || in3[d] > 1) oob = true;
            if (oob) { for (uint32_t i = 0; i < 3 * C; i++) out[i] = 0; continue; }  // gridencoder.cu:115-125
            float pos[3], deriv[3];
            uint32_t pg[3];
            for (int d = 0; d < 3; d++) {
                pos[d] = fmaf(in3[d], scale, align_corners ? 0.0f : 0.5f);
                pg[d] = (uint32_t)floorf(pos[d]);
                pos[d] -= (float)pg[d];
                if (interp == 1) { deriv[d] = 6 * pos[d] * (1 - pos[d]); pos[d] = pos[d] * pos[d] * (3.0f - 2.0f * pos[d]); }
                else deriv[d] = 1.0f;
            }
            for (uint32_t gd = 0; gd < 3; gd++) {
                float rg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                for (uint32_t idx = 0; idx < 4; idx++) {
                    float w = scale;
                    uint32_t pl[3];
                    for (uint32_t nd = 0; nd < 2; nd++) {
                        const uint32_t d = (nd >= gd) ? (nd + 1) : nd;
                        if ((idx & (1u << nd)) == 0) { w *= 1 - pos[d]; pl[d] = pg[d]; }
                        else { w *= pos[d]; pl[d] = pg[d] + 1; }
                    }
                    pl[gd] = pg[gd];
                    const uint32_t il = grid_index(gridtype, align_corners != 0, C, hs, res, pl);
                    pl[gd] = pg[gd] + 1;
                    const uint32_t ir = grid_index(gridtype, align_corners != 0, C, hs, res, pl);
                    for (uint32_t c = 0; c < C; c++) rg[c] += w * (table[ir + c] - table[il + c]) * deriv[gd];
                }
                for (uint32_t c = 0; c < C; c++) out[gd * C + c] = rg[c];
            }
        }
    }
}

// kernel_grid_backward (gridencoder.cu:248-340) + kernel_input_backward (:343-369).  grad [L,B,C]; grad_embeddings zero-filled by the
// caller and accumulated here in sample order (the reference's atomicAdd order is a race); grad_inputs may be NULL.
void orc_grid_encode_backward(const float* grad, const float* inputs, const int* offsets, float* grad_embeddings, uint32_t B, uint32_t C, uint32_t L,
                              float S, uint32_t H, const float* dy_dx, float* grad_inputs, uint32_t gridtype, int align_corners, uint32_t interp) {
    for (uint32_t l = 0; l < L; l++) {
        float scale; uint32_t res;
        level_params(l, S, H, &scale, &res);
        const uint32_t hs = (uint32_t)(offsets[l + 1] - offsets[l]);
        float* gt = grad_embeddings + (size_t)(uint32_t)offsets[l] * C;
        for (uint32_t b = 0; b < B; b++) {
            const float* in3 = inputs + (size_t)b * 3;
            bool oob = false;
            for (int d = 0; d < 3; d++) if (in3[d] < 0 || in3[d] > 1) oob = true;
            if (oob) continue;
            float pos[3];
            uint32_t pg[3];
            for (int d = 0; d < 3; d++) {
                pos[d] = fmaf(in3[d], scale, align_corners ? 0.0f : 0.5f);
                pg[d] = (uint32_t)floorf(pos[d]);
                pos[d] -= (float)pg[d];
                if (interp == 1) pos[d] = pos[d] * pos[d] * (3.0f - 2.0f * pos[d]);
            }
            const float* g = grad + ((size_t)l * B + b) * C;
            for (uint32_t idx = 0; idx < 8; idx++) {
                float w = 1;
                uint32_t pl[3];
                for (int d = 0; d < 3; d++) {
                    if ((idx & (1u << d)) == 0) { w *= 1 - pos[d]; pl[d] = pg[d]; }
                    else { w *= pos[d]; pl[d] = pg[d] + 1; }
                }
                const uint32_t index = grid_index(gridtype, align_corners != 0, C, hs, res, pl);
                for (uint32_t c = 0; c < C; c++) gt[index + c] += w * g[c];
            }
        }
    }
    if (dy_dx && grad_inputs)
        for (uint32_t b = 0; b < B; b++)
            for (uint32_t d = 0; d < 3; d++) {
                float result = 0;
                for (uint32_t l = 0; l < L; l++)
                    for (uint32_t c = 0; c < C; c++) result += grad[((size_t)l * B + b) * C + c] * dy_dx[(((size_t)b * L + l) * 3 + d) * C + c];
                grad_inputs[(size_t)b * 3 + d] = result;
            }
}

// kernel_grad_tv (gridencoder.cu:506-611): accumulates into grad (sample order instead of the reference's atomic race order)
void orc_grad_total_variation(const float* inputs, const float* embeddings, float* grad, const int* offsets, float weight, uint32_t B, uint32_t C,
                              uint32_t L, float S, uint32_t H, uint32_t gridtype, int align_corners) {
    for (uint32_t l = 0; l < L; l++) {
        float scale; uint32_t res;
        level_params(l, S, H, &scale, &res);
        const uint32_t hs = (uint32_t)(offsets[l + 1] - offsets[l]);
        const float* table = embeddings + (size_t)(uint32_t)offsets[l] * C;
        float* gt = grad + (size_t)(uint32_t)offsets[l] * C;
        for (uint32_t b = 0; b < B; b++) {
            const float* in3 = inputs + (size_t)b * 3;
            bool oob = false;
            for (int d = 0; d < 3; d++) if (in3[d] < 0 || in3[d] > 1) oob = true;
            if (oob) continue;
            uint32_t pg[3];
            for (int d = 0; d < 3; d++) pg[d] = (uint32_t)floorf(fmaf(in3[d], scale, align_corners ? 0.0f : 0.5f));
            float results[8] = {0, 0, 0, 0, 0, 0, 0, 0}, idelta[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            const uint32_t index = grid_index(gridtype, align_corners != 0, C, hs, res, pg);
            const float w = weight / (2 * 3);
            for (int d = 0; d < 3; d++) {
                const uint32_t cur = pg[d];
                if (cur < res) {
                    pg[d] = cur + 1;
                    const uint32_t ir = grid_index(gridtype, align_corners != 0, C, hs, res, pg);
                    for (uint32_t c = 0; c < C; c++) { const float gv = table[index + c] - table[ir + c]; results[c] += gv; idelta[c] += gv * gv; }
                }
                if (cur > 0) {
                    pg[d] = cur - 1;
                    const uint32_t il = grid_index(gridtype, align_corners != 0, C, hs, res, pg);
                    for (uint32_t c = 0; c < C; c++) { const float gv = table[index + c] - table[il + c]; results[c] += gv; idelta[c] += gv * gv; }
                }
                pg[d] = cur;
            }
            for (uint32_t c = 0; c < C; c++) gt[index + c] += w * results[c] * (1.0f / sqrtf(idelta[c] + 1e-9f));
        }
    }
}

// kernel_sh's dy_dx branch (shencoder.cu:125-355): dy_dx [B, 3, C*C]; kernel_sh_backward (:358-383): grad_inputs +=
void orc_sh_encode_dy_dx(const float* inputs, float* dy_dx, uint32_t B, uint32_t C) {
    const uint32_t C2 = C * C;
    for (uint32_t b = 0; b < B; b++) {
        float* g = dy_dx + (size_t)b * 3 * C2;
        for (uint32_t i = 0; i < 3 * C2; i++) g[i] = 0.0f;
        if (C <= 4) { sh_one_grad(inputs + (size_t)b * 3, C, g, g + C2, g + 2 * C2); continue; }
        float lo[3][16];
        sh_one_grad(inputs + (size_t)b * 3, 4, lo[0], lo[1], lo[2]);
        for (int d = 0; d < 3; d++) for (int i = 0; i < 16; i++) g[d * C2 + i] = lo[d][i];
        sh_high_bands(inputs + (size_t)b * 3, C, nullptr, g, g + C2, g + 2 * C2);
    }
}
void orc_sh_encode_backward(const float* grad, uint32_t B, uint32_t C, const float* dy_dx, float* grad_inputs) {
    const uint32_t C2 = C * C;
    for (uint32_t t = 0; t < B * 3; t++) {
        const uint32_t b = t / 3, d = t - b * 3;
        for (uint32_t ch = 0; ch < C2; ch++) grad_inputs[t] += grad[(size_t)b * C2 + ch] * dy_dx[((size_t)b * 3 + d) * C2 + ch];
    }
}

}  // extern "C"
