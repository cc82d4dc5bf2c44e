/*
 * recfilter_oracle.c -- CPU restatement of the reference scan operator.
 * TEST INFRASTRUCTURE ONLY (see recfilter_oracle.h for the rules and the
 * reference file:line each function follows).
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp).
 * -ffp-contract=off keeps the f32 path a plain mul/add chain in the order the
 * reference expression is written (feedforward term first, then feedback taps
 * j = 0..order-1, lib/recfilter.cpp:324-340).
 */
#include "recfilter_oracle.h"

#include <complex.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

int orc_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------------- */
/* The scan operator, lib/recfilter.cpp:302-343.
 *
 * A "panel" is `w` adjacent lines that are contiguous in memory (w = 1 for a
 * scan along x; up to PANEL lines for scans along y/z where the x axis is
 * contiguous).  Lines never interact, so processing them side by side is the
 * same computation as one line at a time; it only makes the memory walk
 * cache friendly.  Position r runs 0..n-1; the element touched is i = r for a
 * causal scan and n-1-r for an anticausal one (:311-315).
 *
 *   zero border   (:337-340)  tap j contributes only when r > j
 *   clamp border  (:330-336)  tap index is clamped into [0, n-1] and read from
 *                             the partially updated buffer
 *
 * ACC is the type the arithmetic is carried in: the pixel type for floats, and
 * uint32 for the integer pixel types (wrap-around arithmetic is a ring
 * homomorphism, so truncating once at the store equals truncating per op as
 * the reference's int16/int32 expressions do).
 */
#define PANEL 256

#define DEFINE_SCAN_PANEL(NAME, T, ACC)                                                      \
    static void NAME(T *base, int64_t n, int64_t stride, int64_t w, int causal, int order,   \
                     const ACC *c, int clamp) {                                              \
        ACC tmp[PANEL];                                                                      \
        int64_t r_end = n;                                                                   \
        if (w == 1 && n > order) r_end = order;   /* single line: only the border samples need the general form */ \
        for (int64_t r = 0; r < r_end; r++) {                                                \
            int64_t i = causal ? r : n - 1 - r;                                              \
            T *cur = base + i * stride;                                                      \
            for (int64_t v = 0; v < w; v++) tmp[v] = (ACC)(c[0] * (ACC)cur[v]);              \
            for (int j = 0; j < order; j++) {                                                \
                int64_t t = causal ? i - (j + 1) : i + (j + 1);                              \
                if (clamp) {                                                                 \
                    if (t < 0) t = 0;                                                        \
                    if (t > n - 1) t = n - 1;                                                \
                } else if (!(r > j)) {                                                       \
                    continue; /* select(rx>j, f(..), 0) */                                   \
                }                                                                            \
                const T *tap = base + t * stride;                                            \
                ACC a = c[1 + j];                                                            \
                for (int64_t v = 0; v < w; v++) tmp[v] = (ACC)(tmp[v] + (ACC)(a * (ACC)tap[v])); \
            }                                                                                \
            for (int64_t v = 0; v < w; v++) cur[v] = (T)tmp[v];                              \
        }                                                                                    \
        /* same arithmetic in the same order for the samples whose taps are all inside the line (r >= order) */ \
        const int64_t step = causal ? stride : -stride;                                      \
        T *cur = base + (causal ? r_end : n - 1 - r_end) * stride;                           \
        for (int64_t r = r_end; r < n; r++, cur += step) {                                   \
            ACC acc = (ACC)(c[0] * (ACC)cur[0]);                                             \
            for (int j = 0; j < order; j++) acc = (ACC)(acc + (ACC)(c[1 + j] * (ACC)cur[-(j + 1) * step])); \
            cur[0] = (T)acc;                                                                 \
        }                                                                                    \
    }

DEFINE_SCAN_PANEL(scan_panel_f32, float, float)
DEFINE_SCAN_PANEL(scan_panel_f64, double, double)
DEFINE_SCAN_PANEL(scan_panel_i32, int32_t, uint32_t)
DEFINE_SCAN_PANEL(scan_panel_i16, int16_t, uint32_t)

static size_t dtype_size(int dtype) {
    switch (dtype) {
        case ORC_F32: return 4;
        case ORC_F64: return 8;
        case ORC_I32: return 4;
        case ORC_I16: return 2;
        default: return 0;
    }
}

int orc_apply_scan(void *data, int dtype, int ndim, const int64_t *extent,
                   const orc_scan *scan, int border, int threads) {
    if (!data || !extent || !scan) return -1;
    if (ndim < 1 || ndim > ORC_MAX_DIMS) return -2;
    if (scan->dim < 0 || scan->dim >= ndim) return -3;
    if (scan->order < 1 || scan->order > ORC_MAX_ORDER) return -4; /* needs feedfwd + >=1 feedback, :274 */
    size_t esz = dtype_size(dtype);
    if (!esz) return -5;

    int64_t n = extent[scan->dim];
    int64_t inner = 1, outer = 1;
    for (int d = 0; d < scan->dim; d++) inner *= extent[d];
    for (int d = scan->dim + 1; d < ndim; d++) outer *= extent[d];
    if (n <= 0 || inner <= 0 || outer <= 0) return 0;

    /* coefficients cast to the pixel type, lib/recfilter.cpp:324,335,338 */
    float  cf[ORC_MAX_ORDER + 1];
    double cd[ORC_MAX_ORDER + 1];
    uint32_t cu[ORC_MAX_ORDER + 1];
    for (int j = 0; j <= scan->order; j++) {
        cf[j] = scan->coeff[j];
        cd[j] = (double)scan->coeff[j];
        if (dtype == ORC_I16) cu[j] = (uint32_t)(int32_t)(int16_t)scan->coeff[j];
        else                  cu[j] = (uint32_t)(int32_t)scan->coeff[j];
    }

    /* panel width: PANEL lines when there are plenty of panels, narrower (>= 16 lines, one cache line of floats)
     * when the threads would otherwise idle -- e.g. a 4096-wide image gives only 16 panels of 256 columns */
    int64_t pw = PANEL;
    {
        int nthr = threads > 1 ? threads : 1;
        int64_t want = inner / (4 * (int64_t)nthr);
        want &= ~(int64_t)15;
        if (want < 16) want = 16;
        if (want < pw) pw = want;
    }
    int64_t panels_per_outer = (inner + pw - 1) / pw;
    int64_t n_tasks = outer * panels_per_outer;
    int clamp = (border == ORC_BORDER_CLAMP);
    int causal = scan->causal ? 1 : 0;
    int order = scan->order;
    char *bytes = (char *)data;
    (void)threads;

#ifdef _OPENMP
    int nt = threads > 1 ? threads : 1;
#pragma omp parallel for schedule(static) num_threads(nt) if (nt > 1)
#endif
    for (int64_t task = 0; task < n_tasks; task++) {
        int64_t o = task / panels_per_outer;
        int64_t p = task % panels_per_outer;
        int64_t x0 = p * pw;
        int64_t w = inner - x0 < pw ? inner - x0 : pw;
        size_t off = (size_t)(o * n * inner + x0);
        switch (dtype) {
            case ORC_F32: scan_panel_f32((float *)bytes + off, n, inner, w, causal, order, cf, clamp); break;
            case ORC_F64: scan_panel_f64((double *)bytes + off, n, inner, w, causal, order, cd, clamp); break;
            case ORC_I32: scan_panel_i32((int32_t *)bytes + off, n, inner, w, causal, order, cu, clamp); break;
            case ORC_I16: scan_panel_i16((int16_t *)bytes + off, n, inner, w, causal, order, cu, clamp); break;
        }
    }
    return 0;
}

int orc_apply_filter(void *data, int dtype, int ndim, const int64_t *extent,
                     const orc_scan *scans, int n_scans, int border, int threads) {
    for (int s = 0; s < n_scans; s++) {
        int rc = orc_apply_scan(data, dtype, ndim, extent, &scans[s], border, threads);
        if (rc) return rc;
    }
    return 0;
}

/* ------------------------------------------------------------------------- */
/* Coefficient design, lib/iir_coeff.cpp.
 * van Vliet-Young-Verbeek recursive Gaussian via pole rescaling; the float /
 * double mix below follows the reference's declared types because the quoted
 * values depend on where it rounds to float. */

static float orc_qs(float s) { return (float)(0.00399341 + 0.4715161 * s); }   /* :38-40 */

static double complex orc_ds_c(double complex d, float s) {                    /* :60-63 */
    double q = orc_qs(s);
    double mag = pow(cabs(d), 1.0 / q);
    double ang = carg(d) / q;
    return mag * cos(ang) + I * (mag * sin(ang));
}

static float orc_ds_r(float d, float s) { return (float)pow(d, 1.0 / orc_qs(s)); } /* :83-85 */

static void orc_weights1(float s, float *b0, float *a1) {                      /* :103-108 */
    const float d3 = 1.86543f;
    float d = orc_ds_r(d3, s);
    *b0 = (float)(-(1.0 - d) / d);
    *a1 = (float)(-1.0 / d);
}

static void orc_weights2(float s, float *b0, float *a1, float *a2) {           /* :127-136 */
    double complex d1 = 1.41650 + I * 1.00829;
    double complex d = orc_ds_c(d1, s);
    float n2 = (float)cabs(d);
    n2 *= n2;
    float re = (float)creal(d);
    *b0 = (float)((1.0 - 2.0 * re + n2) / n2);
    *a1 = (float)(-2.0 * re / n2);
    *a2 = (float)(1.0 / n2);
}

static void orc_weights3(float s, float *b0, float *a1, float *a2, float *a3) { /* :150-159 */
    float b10, b20, a11, a21, a22;
    orc_weights1(s, &b10, &a11);
    orc_weights2(s, &b20, &a21, &a22);
    *a1 = a11 + a21;
    *a2 = a11 * a21 + a22;
    *a3 = a11 * a22;
    *b0 = b10 * b20;
}

void orc_gaussian_weights(float sigma, int order, float *out) {                /* :162-177 */
    /* the reference allocates order+1 slots but its default branch always
     * writes four; order > 3 is therefore out of contract and treated as 3 */
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    switch (order) {
        case 1: orc_weights1(sigma, &a[0], &a[1]); break;
        case 2: orc_weights2(sigma, &a[0], &a[1], &a[2]); break;
        default: orc_weights3(sigma, &a[0], &a[1], &a[2], &a[3]); order = 3; break;
    }
    out[0] = a[0];
    for (int i = 1; i <= order; i++) out[i] = -a[i];
}

static int orc_factorial(int k) { int r = 1; for (int i = 1; i <= k; i++) r *= i; return r; }

void orc_integral_image_coeff(int n, float *out) {                             /* :222-234 */
    out[0] = 1.0f;
    for (int i = 1; i <= n; i++) {
        int n_choose_i = orc_factorial(n) / (orc_factorial(i) * orc_factorial(n - i));
        float binom = (float)(pow(-1.0f, i) * (float)n_choose_i);              /* :18-21, r = 1 */
        out[i] = -1.0f * binom;
    }
}

void orc_overlap_feedback_coeff(const float *a, int na, const float *b, int nb, float *c) { /* :236-263 */
    float pa[32], pb[32], pc[64];
    pa[0] = 1.0f; for (int i = 0; i < na; i++) pa[i + 1] = -a[i];
    pb[0] = 1.0f; for (int i = 0; i < nb; i++) pb[i + 1] = -b[i];
    int la = na + 1, lb = nb + 1, lc = la + lb - 1;
    for (int i = 0; i < lc; i++) {
        pc[i] = 0.0f;
        for (int j = 0; j <= i; j++)
            if (j < la && i - j < lb) pc[i] += pa[j] * pb[i - j];
    }
    for (int i = 1; i < lc; i++) c[i - 1] = -pc[i];
}

int orc_gaussian_box_filter(int k, float sigma) {                              /* :205-220 */
    float sum = 0.0f;
    float alpha = 0.005f;
    int sum_limit = (int)floorf(((float)k - 1.0f) / 2.0f);
    for (int i = 0; i <= sum_limit; i++) {
        int f_k = orc_factorial(k), f_i = orc_factorial(i);
        int f_k_i = orc_factorial(k - i), f_k_1 = orc_factorial(k - 1);
        float f = (float)(f_k / (f_i * f_k_i));
        float p = (float)(pow(-1.0, i) / (float)f_k_1);
        sum += (float)(p * f * pow(((float)k / 2.0 - i), k - 1));
    }
    sum = (float)(sqrt(2.0 * M_PI) * (sum + alpha) * sigma);
    return (int)ceilf(sum);
}

/* ------------------------------------------------------------------------- */
/* Tile matrices, lib/coefficients.cpp (float arithmetic like the reference). */

void orc_matrix_B(float feedfwd, const float *feedback, int order, int tile,
                  int clamp_border, float *B) {                                /* :8-49 */
    /* B[row][col]: output position `row` <- input position `col` */
    for (int row = 0; row < tile; row++)
        for (int col = 0; col < tile; col++)
            B[row * tile + col] = (row == col) ? feedfwd : 0.0f;
    for (int row = 0; row < tile; row++) {
        for (int col = 0; col < tile; col++) {
            for (int j = 0; j < order; j++) {
                float a;
                if (row - j - 1 >= 0) a = B[(row - j - 1) * tile + col] * feedback[j];
                else if (clamp_border) a = (col == 0) ? feedback[j] : 0.0f;    /* :38-39 */
                else a = 0.0f;
                B[row * tile + col] += a;
            }
        }
    }
}

void orc_matrix_R(const float *feedback, int order, int tile, float *R) {      /* :51-83 */
    for (int p = 0; p < tile; p++)
        for (int j = 0; j < order; j++) R[p * order + j] = 0.0f;
    for (int p = 0; p < tile; p++) {
        for (int j = 0; j < order; j++) {
            if (p < order) R[p * order + j] = (p + j < order) ? feedback[p + j] : 0.0f;
            for (int q = 0; p - q - 1 >= 0 && q < order; q++)
                R[p * order + j] += R[(p - q - 1) * order + j] * feedback[q];
        }
    }
}

/* ------------------------------------------------------------------------- */

double orc_check_result_f32(const float *ref, const float *out, size_t n, double *mean_pct) { /* recfilter.h:818-821 */
    double mx = 0.0, mean = 0.0;
    for (size_t i = 0; i < n; i++) {
        double diff = (double)ref[i] - (double)out[i];
        double re = 100.0 * fabs(diff) / ((double)ref[i] + 1e-9);
        mean += re;
        if (re > mx) mx = re;
    }
    if (mean_pct) *mean_pct = n ? mean / (double)n : 0.0;
    return mx;
}
