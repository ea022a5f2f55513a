// Real spherical-harmonics bands l = 4..7 (SHEncoder degree 5-8: shencoder/src/shencoder.cu:69-123 forward, :125-355 derivatives) for the op-level
// API.  The render path uses degree 4 (pn_net_tile.h: sh16); these bands exist so that the drop-in covers the extension's whole interface
// (`SHEncoder` asserts degree <= 8, shencoder/sphere_harmonics.py:70).
//
// Evaluated by recurrence in double and narrowed once, instead of 48 expanded polynomials:
//   a_m + i b_m = (x + i y)^m                                  (a_m = x a_{m-1} - y b_{m-1},  b_m = x b_{m-1} + y a_{m-1})
//   t_m^m = (2m - 1)!!,  t_{m+1}^m = (2m + 1) z t_m^m,  (l - m) t_l^m = (2l - 1) z t_{l-1}^m - (l + m - 1) t_{l-2}^m     (t_l^m = d^m P_l / dz^m)
//   Y_l^{+-m} = (-1)^m sqrt2 K_l^m t_l^m {a_m, b_m},  Y_l^0 = K_l^0 t_l^0,  K_l^m = sqrt((2l + 1) / (4 pi) (l - m)! / (l + m)!)
// — the basis that keeps the Condon-Shortley phase, as the reference's list does (Y_1^{-1} = -c y).  As polynomials in (x, y, z) these are the
// reference's expressions (t in z only, a / b in x, y only), so values and partial derivatives agree off the unit sphere as well:
//   d/dx: m a_{m-1} | m b_{m-1};  d/dy: -m b_{m-1} | m a_{m-1};  d/dz: t_l^{m+1}.
#pragma once

namespace pnsh {

// out / gx / gy / gz: arrays of C * C floats (any may be null); entries [16, C * C) are written.
__device__ inline void high_bands(float xf, float yf, float zf, int C, float* out, float* gx, float* gy, float* gz) {
    const double x = xf, y = yf, z = zf;
    double a[9], b[9];
    a[0] = 1.0; b[0] = 0.0;
    for (int m = 1; m <= 8; m++) { a[m] = x * a[m - 1] - y * b[m - 1]; b[m] = x * b[m - 1] + y * a[m - 1]; }
    double fact[16];
    fact[0] = 1.0;
    for (int i = 1; i < 16; i++) fact[i] = fact[i - 1] * i;
    const double inv4pi = 0.07957747154594767;  // 1 / (4 pi)
    // t[m][l], m <= 8, l <= 7 (t_l^m = 0 for m > l)
    double t[9][8];
    double dfact = 1.0;                          // (2m - 1)!!
    for (int m = 0; m <= 8; m++) {
        if (m > 0) dfact *= (2 * m - 1);
        for (int l = 0; l < 8; l++) t[m][l] = 0.0;
        if (m < 8) t[m][m] = dfact;
        if (m + 1 < 8) t[m][m + 1] = (2 * m + 1) * z * dfact;
        for (int l = m + 2; l < 8; l++) t[m][l] = ((2 * l - 1) * z * t[m][l - 1] - (l + m - 1) * t[m][l - 2]) / (double)(l - m);
    }
    for (int l = 4; l < C; l++)
        for (int m = 0; m <= l; m++) {
            double K = sqrt((2 * l + 1) * inv4pi * fact[l - m] / fact[l + m]);
            if (m) K *= ((m & 1) ? -1.4142135623730951 : 1.4142135623730951);
            const int ip = l * l + l + m, im = l * l + l - m;
            const double T = K * t[m][l], dT = K * t[m + 1][l];
            if (out) { out[ip] = (float)(T * a[m]); if (m) out[im] = (float)(T * b[m]); }
            if (gx) {
                const double am = m ? m * a[m - 1] : 0.0, bm = m ? m * b[m - 1] : 0.0;
                gx[ip] = (float)(T * am); gy[ip] = (float)(-T * bm); gz[ip] = (float)(dT * a[m]);
                if (m) { gx[im] = (float)(T * bm); gy[im] = (float)(T * am); gz[im] = (float)(dT * b[m]); }
            }
        }
}

}  // namespace pnsh
