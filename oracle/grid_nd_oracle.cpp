// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// CPU restatement of the hash-grid encoder for input dimensions D = 2, 4, 5 (and 3, as a cross-check of render_oracle.cpp's D = 3 code):
// gridencoder/src/gridencoder.cu:50-84 (fast_hash<D>, get_grid_index<D,C>), :87-245 (kernel_grid<float,D,C> with its dy_dx branch), :248-340
// (kernel_grid_backward), :343-369 (kernel_input_backward), dispatched over D at :386-399 and :430-444.  The render path uses D = 3
// (render_oracle.cpp); this file exists so that the drop-in's stand-alone op covers the extension's whole interface.  fp32, sequential sums in the
// reference's corner order; `inputs * scale + offset` with one rounding (nvcc's default contraction, as in render_oracle.cpp: grid_one).
// Pinned by tests/test_gpu_ref.py against the reference's own kernel_grid<float, D, C> compiled for gfx950 (oracle/_ref).
#include <cmath>
#include <cstdint>
#include <cstddef>

namespace {

constexpr uint32_t PRIMES[7] = {1u, 2654435761u, 805459861u, 3674653429u, 2097192037u, 1434869437u, 2165219737u};  // gridencoder.cu:54

// gridencoder.cu:66-84
inline uint32_t index_nd(uint32_t D, uint32_t gridtype, bool align_corners, uint32_t C, uint32_t hashmap_size, uint32_t resolution, const uint32_t* p) {
    uint32_t stride = 1, index = 0;
    for (uint32_t d = 0; d < D && stride <= hashmap_size; d++) {
        index += p[d] * stride;
        stride *= align_corners ? resolution : (resolution + 1);
    }
    if (gridtype == 0 && stride > hashmap_size) {
        index = 0;
        for (uint32_t d = 0; d < D; d++) index ^= p[d] * PRIMES[d];
    }
    return (index % hashmap_size) * C;
}

struct Cell {
    float pos[5], deriv[5];
    uint32_t pg[5];
};

inline bool locate(const float* in, uint32_t D, float scale, bool align_corners, uint32_t interp, Cell& c) {
    for (uint32_t d = 0; d < D; d++) if (in[d] < 0 || in[d] > 1) return false;      // :113-133
    for (uint32_t d = 0; d < D; d++) {
        float p = std::fmaf(in[d], scale, align_corners ? 0.0f : 0.5f);            // :143
        c.pg[d] = (uint32_t)std::floor(p);
        p -= (float)c.pg[d];
        if (interp == 1) { c.deriv[d] = 6 * p * (1.0f - p); p = p * p * (3.0f - 2.0f * p); }   // :40-47,147-152
        else c.deriv[d] = 1.0f;
        c.pos[d] = p;
    }
    return true;
}

}  // namespace

extern "C" {

// outputs [L, B, C]; dy_dx (may be NULL) [B, L, D, C]
void orc_grid_nd_forward(const float* inputs, const float* embeddings, const int* offsets, float* outputs, float* dy_dx, uint32_t B, uint32_t D, uint32_t C,
                         uint32_t L, float S, uint32_t H, uint32_t gridtype, int align_corners, uint32_t interp) {
    if (D < 2 || D > 5 || C > 8) return;
    for (uint32_t l = 0; l < L; l++) {
        const float scale = exp2f(l * S) * H - 1.0f;
        const uint32_t res = (uint32_t)std::ceil(scale) + 1, hs = (uint32_t)(offsets[l + 1] - offsets[l]);
        const float* table = embeddings + (size_t)(uint32_t)offsets[l] * C;
        for (uint32_t b = 0; b < B; b++) {
            float* out = outputs + ((size_t)l * B + b) * C;
            float* dd = dy_dx ? dy_dx + ((size_t)b * L + l) * D * C : nullptr;
            Cell c;
            if (!locate(inputs + (size_t)b * D, D, scale, align_corners != 0, interp, c)) {
                for (uint32_t ch = 0; ch < C; ch++) out[ch] = 0;
                if (dd) for (uint32_t i = 0; i < D * C; i++) dd[i] = 0;
                continue;
            }
            float r[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (uint32_t idx = 0; idx < (1u << D); idx++) {                          // :160-186
                float w = 1;
                uint32_t pl[5];
                for (uint32_t d = 0; d < D; d++) {
                    if ((idx & (1u << d)) == 0) { w *= 1 - c.pos[d]; pl[d] = c.pg[d]; }
                    else { w *= c.pos[d]; pl[d] = c.pg[d] + 1; }
                }
                const uint32_t index = index_nd(D, gridtype, align_corners != 0, C, hs, res, pl);
                for (uint32_t ch = 0; ch < C; ch++) r[ch] += w * table[index + ch];
            }
            for (uint32_t ch = 0; ch < C; ch++) out[ch] = r[ch];
            if (!dd) continue;
            for (uint32_t gd = 0; gd < D; gd++) {                                     // :204-243
                float g[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                for (uint32_t idx = 0; idx < (1u << (D - 1)); idx++) {
                    float w = scale;
                    uint32_t pl[5];
                    for (uint32_t nd = 0; nd < D - 1; nd++) {
                        const uint32_t d = (nd >= gd) ? (nd + 1) : nd;
                        if ((idx & (1u << nd)) == 0) { w *= 1 - c.pos[d]; pl[d] = c.pg[d]; }
                        else { w *= c.pos[d]; pl[d] = c.pg[d] + 1; }
                    }
                    pl[gd] = c.pg[gd];
                    const uint32_t il = index_nd(D, gridtype, align_corners != 0, C, hs, res, pl);
                    pl[gd] = c.pg[gd] + 1;
                    const uint32_t ir = index_nd(D, gridtype, align_corners != 0, C, hs, res, pl);
                    for (uint32_t ch = 0; ch < C; ch++) g[ch] += w * (table[ir + ch] - table[il + ch]) * c.deriv[gd];
                }
                for (uint32_t ch = 0; ch < C; ch++) dd[gd * C + ch] = g[ch];
            }
        }
    }
}

// grad [L, B, C]; grad_embeddings (zeroed by the caller) [offsets[L], C], accumulated in ascending (level, sample, corner) order — one member of the
// reference's atomicAdd outcome set; grad_inputs [B, D] = sum over (level, channel) of grad * dy_dx when dy_dx is given (:343-369)
void orc_grid_nd_backward(const float* grad, const float* inputs, const int* offsets, float* grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L,
                          float S, uint32_t H, const float* dy_dx, float* grad_inputs, uint32_t gridtype, int align_corners, uint32_t interp) {
    if (D < 2 || D > 5 || C > 8) return;
    for (uint32_t l = 0; l < L; l++) {
        const float scale = exp2f(l * S) * H - 1.0f;
        const uint32_t res = (uint32_t)std::ceil(scale) + 1, hs = (uint32_t)(offsets[l + 1] - offsets[l]);
        float* gt = grad_embeddings + (size_t)(uint32_t)offsets[l] * C;
        for (uint32_t b = 0; b < B; b++) {
            Cell c;
            if (!locate(inputs + (size_t)b * D, D, scale, align_corners != 0, interp, c)) continue;   // :276-281
            const float* g = grad + ((size_t)l * B + b) * C;
            for (uint32_t idx = 0; idx < (1u << D); idx++) {
                float w = 1;
                uint32_t pl[5];
                for (uint32_t d = 0; d < D; d++) {
                    if ((idx & (1u << d)) == 0) { w *= 1 - c.pos[d]; pl[d] = c.pg[d]; }
                    else { w *= c.pos[d]; pl[d] = c.pg[d] + 1; }
                }
                const uint32_t index = index_nd(D, gridtype, align_corners != 0, C, hs, res, pl);
                for (uint32_t ch = 0; ch < C; ch++) gt[index + ch] += w * g[ch];
            }
        }
    }
    if (!dy_dx || !grad_inputs) return;
    for (uint32_t t = 0; t < B * D; t++) {
        const uint32_t b = t / D, d = t - b * D;
        float r = 0;
        for (uint32_t l = 0; l < L; l++)
            for (uint32_t ch = 0; ch < C; ch++) r += grad[((size_t)l * B + b) * C + ch] * dy_dx[(((size_t)b * L + l) * D + d) * C + ch];
        grad_inputs[t] = r;
    }
}

// kernel_grad_tv<float, D, C> (gridencoder.cu:506-611): grad (accumulated into; the caller's tensor) [offsets[L], C]; sequential in (level, sample) order
void orc_grid_nd_grad_tv(const float* inputs, const float* embeddings, float* grad, const int* offsets, float weight, uint32_t B, uint32_t D, uint32_t C,
                         uint32_t L, float S, uint32_t H, uint32_t gridtype, int align_corners) {
    if (D < 2 || D > 5 || C > 8) return;
    for (uint32_t l = 0; l < L; l++) {
        const float scale = exp2f(l * S) * H - 1.0f;
        const uint32_t res = (uint32_t)std::ceil(scale) + 1, hs = (uint32_t)(offsets[l + 1] - offsets[l]);
        const float* table = embeddings + (size_t)(uint32_t)offsets[l] * C;
        float* gt = grad + (size_t)(uint32_t)offsets[l] * C;
        for (uint32_t b = 0; b < B; b++) {
            const float* in = inputs + (size_t)b * D;
            bool oob = false;
            for (uint32_t d = 0; d < D; d++) if (in[d] < 0 || in[d] > 1) oob = true;
            if (oob) continue;
            uint32_t pg[5];
            for (uint32_t d = 0; d < D; d++) pg[d] = (uint32_t)std::floor(std::fmaf(in[d], scale, align_corners ? 0.0f : 0.5f));
            float results[8] = {0, 0, 0, 0, 0, 0, 0, 0}, idelta[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            const uint32_t index = index_nd(D, gridtype, align_corners != 0, C, hs, res, pg);
            const float w = weight / (2 * D);
            for (uint32_t d = 0; d < D; d++) {
                const uint32_t cur = pg[d];
                if (cur < res) {
                    pg[d] = cur + 1;
                    const uint32_t ir = index_nd(D, gridtype, align_corners != 0, C, hs, res, pg);
                    for (uint32_t ch = 0; ch < C; ch++) { const float gv = table[index + ch] - table[ir + ch]; results[ch] += gv; idelta[ch] += gv * gv; }
                }
                if (cur > 0) {
                    pg[d] = cur - 1;
                    const uint32_t il = index_nd(D, gridtype, align_corners != 0, C, hs, res, pg);
                    for (uint32_t ch = 0; ch < C; ch++) { const float gv = table[index + ch] - table[il + ch]; results[ch] += gv; idelta[ch] += gv * gv; }
                }
                pg[d] = cur;
            }
            for (uint32_t ch = 0; ch < C; ch++) gt[index + ch] += w * results[ch] * (1.0f / std::sqrt(idelta[ch] + 1e-9f));
        }
    }
}

}  // extern "C"
