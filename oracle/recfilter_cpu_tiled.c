/*
 * recfilter_cpu_tiled.c -- CPU counterpart of the reference's TILED CPU schedule, for the bench's baseline.
 * TEST / BENCH INFRASTRUCTURE ONLY (same rules as recfilter_oracle.c): never linked into the product.
 *
 * What it stands for: RecFilter::cpu_auto_schedule on a filter that was split()
 *     lib/recfilter.cpp:610-678   cpu_auto_intra_schedule / cpu_auto_inter_schedule:
 *                                 intra-tile stages computed per tile with the scan loop innermost, tiles in
 *                                 parallel over `outer`, inner(0) vectorised; inter-tile carry stages serial
 *                                 along the tile axis, parallel over the remaining axes
 * running the tiled algorithm of lib/split.cpp in the form SURVEY.md Appendix A restates it:
 *     pass 1  per tile: every scan of the dimension with zero entering state, k-sample tail kept per scan
 *                                                                  (split.cpp:503-665, 256-499)
 *     carry   per line, per scan in application order: same-dimension chaining with the earlier scans' completed
 *             carries, then the recurrence over the tiles            (split.cpp:912-1004, 743-867)
 *     pass 2  per tile: every scan again, entering with the completed carry of the previous tile
 *                                                                  (split.cpp:1008-1130, 1647-1780)
 * one dimension after the other (the scans of different dimensions commute, split.cpp:215-242).  The weights of
 * the carry stage (matrix_R / tail_weights, coefficients.cpp:51-83, split.cpp:152-203) are built the way the
 * product builds them: by running the tile scan on unit carries, which makes them consistent with the in-place
 * clamped border of lib/recfilter.cpp:330-336.
 *
 * f32 pixels only (the bench's type).  Lines adjacent in memory (scans along y / z) are processed side by side in
 * panels, which is what Halide's vectorize(inner(0), 8) amounts to; scans along x run one row per task.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "recfilter_oracle.h"

#ifdef _OPENMP
#include <omp.h>
#endif

#define TP_MAXK 8
#define TP_MAXS 16
#define TP_PANEL 64
#define TP_MAXT 256

typedef struct {
    int causal, order;
    float b, a[TP_MAXK];
} tp_scan;

/* one scan over a tile of T samples for `w` side-by-side lines; v[p * w + lane], p = MEMORY position inside the tile.
 * h[j * w + lane] = j-th previous output entering the tile (j = 0 adjacent), all zero for a tile-local scan.
 * clamp_first: the tile is the image-border tile of this scan and the border is clamped (recfilter.cpp:330-336:
 * at r = 0 every tap reads the not yet updated sample, afterwards out-of-range taps read the updated first one). */
static void tp_scan_tile(float *v, int T, int w, const tp_scan *s, int k, float *h, int clamp_first) {
    float y[TP_PANEL];
    for (int p = 0; p < T; p++) {
        float *cur = v + (size_t)(s->causal ? p : T - 1 - p) * w;
        if (clamp_first && p == 0)
            for (int j = 0; j < k; j++)
                for (int l = 0; l < w; l++) h[j * w + l] = cur[l];
        for (int l = 0; l < w; l++) y[l] = s->b * cur[l];
        for (int j = 0; j < s->order; j++) {
            const float a = s->a[j];
            const float *hj = h + (size_t)j * w;
            for (int l = 0; l < w; l++) y[l] = y[l] + a * hj[l];
        }
        if (clamp_first && p == 0) {
            for (int j = 0; j < k; j++)
                for (int l = 0; l < w; l++) h[j * w + l] = y[l];
        } else {
            for (int j = k - 1; j > 0; j--) memcpy(h + (size_t)j * w, h + (size_t)(j - 1) * w, sizeof(float) * w);
            memcpy(h, y, sizeof(float) * w);
        }
        memcpy(cur, y, sizeof(float) * w);
    }
}

typedef struct {
    int n_scans, k, T, M, clamp;
    tp_scan sc[TP_MAXS];
    /* A[s][r][o]; W[variant][q][s][r][o] (variant bit 0: first tile of the line, bit 1: last tile) */
    float A[TP_MAXS][TP_MAXK][TP_MAXK];
    float *W;
} tp_dim;

static float *tp_W(const tp_dim *d, int v, int q, int s) {
    return d->W + ((((size_t)v * d->n_scans + q) * d->n_scans + s) * d->k) * d->k;
}

/* tail r of scan s inside a tile (memory index): r-th sample from the end in scan direction */
static int tp_tail_pos(const tp_scan *s, int T, int r) { return s->causal ? T - 1 - r : r; }

static int tp_is_first(const tp_scan *s, int t, int M) { return s->causal ? (t == 0) : (t == M - 1); }

static int tp_build_tables(tp_dim *d) {
    const int k = d->k, T = d->T, n = d->n_scans;
    d->W = (float *)calloc((size_t)4 * n * n * k * k, sizeof(float));
    if (!d->W) return -1;
    float v[TP_MAXT], h[TP_MAXK];
    for (int variant = 0; variant < 4; variant++) {
        for (int q = 0; q < n; q++) {
            for (int o = 0; o < k; o++) {
                /* response of the tile to a unit carry component o entering scan q, then pushed through the later
                 * scans of the dimension tile-locally */
                memset(v, 0, sizeof(float) * T);
                for (int j = 0; j < k; j++) h[j] = (j == o) ? 1.0f : 0.0f;
                tp_scan zero_input = d->sc[q];
                tp_scan_tile(v, T, 1, &zero_input, k, h, 0);
                if (variant == 0)
                    for (int r = 0; r < k; r++) d->A[q][r][o] = v[tp_tail_pos(&d->sc[q], T, r)];
                for (int s = q + 1; s < n; s++) {
                    const int first_tile = variant & 1, last_tile = (variant >> 1) & 1;
                    const int border = d->sc[s].causal ? first_tile : last_tile;
                    float hz[TP_MAXK] = {0};
                    tp_scan_tile(v, T, 1, &d->sc[s], k, hz, d->clamp && border);
                    float *Wm = tp_W(d, variant, q, s);
                    for (int r = 0; r < k; r++) Wm[r * k + o] = v[tp_tail_pos(&d->sc[s], T, r)];
                }
            }
        }
    }
    return 0;
}

/* Tiled filtering along ONE dimension: n = extent, `inner` = stride of the dimension (lines adjacent in memory when
 * inner > 1), `outer` = number of slabs.  tails: [scan][tile][r][line]. */
static int tp_run_dim(float *data, int64_t n, int64_t inner, int64_t outer, tp_dim *d, int threads) {
    const int T = d->T, k = d->k, ns = d->n_scans;
    const int M = (int)(n / T);
    d->M = M;
    const int64_t lines = inner * outer;
    const int pw = inner >= TP_PANEL ? TP_PANEL : (int)inner;
    const int64_t panels = (inner + pw - 1) / pw;
    float *tails = (float *)malloc(sizeof(float) * (size_t)ns * M * k * lines);
    if (!tails) return -1;
    const int nt = threads > 1 ? threads : 1;
    (void)nt;
#define TAIL(s, t, r, line) tails[(((size_t)(s) * M + (t)) * k + (r)) * lines + (line)]

    /* ---- pass 1: tiles in parallel, all scans of the dimension tile-locally, tails kept ---- */
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nt) if (nt > 1)
#endif
    for (int64_t task = 0; task < outer * M * panels; task++) {
        const int64_t o = task / (M * panels), rem = task % (M * panels);
        const int t = (int)(rem / panels);
        const int64_t x0 = (rem % panels) * pw;
        const int w = (int)(inner - x0 < pw ? inner - x0 : pw);
        float buf[TP_MAXT * TP_PANEL], h[TP_MAXK * TP_PANEL];
        const float *src = data + (size_t)o * n * inner + (size_t)t * T * inner + x0;
        for (int p = 0; p < T; p++) memcpy(buf + (size_t)p * w, src + (size_t)p * inner, sizeof(float) * w);
        for (int s = 0; s < ns; s++) {
            memset(h, 0, sizeof(float) * k * w);
            tp_scan_tile(buf, T, w, &d->sc[s], k, h, d->clamp && tp_is_first(&d->sc[s], t, M));
            for (int r = 0; r < k; r++)
                memcpy(&TAIL(s, t, r, o * inner + x0), buf + (size_t)tp_tail_pos(&d->sc[s], T, r) * w, sizeof(float) * w);
        }
    }

    /* ---- carry: lines in parallel, serial over the tiles ---- */
    const int64_t cw = 256;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nt) if (nt > 1)
#endif
    for (int64_t c0 = 0; c0 < lines; c0 += cw) {
        const int64_t c1 = c0 + cw < lines ? c0 + cw : lines;
        for (int s = 0; s < ns; s++) {
            const tp_scan *sc = &d->sc[s];
            /* same-dimension chaining with the completed carries of the earlier scans */
            for (int q = 0; q < s; q++) {
                const tp_scan *sq = &d->sc[q];
                for (int t = 0; t < M; t++) {
                    if (tp_is_first(sq, t, M)) continue;
                    const int tp = sq->causal ? t - 1 : t + 1;
                    const int variant = (t == 0 ? 1 : 0) | (t == M - 1 ? 2 : 0);
                    const float *Wm = tp_W(d, variant, q, s);
                    for (int r = 0; r < k; r++)
                        for (int o = 0; o < k; o++) {
                            const float wv = Wm[r * k + o];
                            if (wv == 0.0f) continue;
                            float *dst = &TAIL(s, t, r, 0);
                            const float *cq = &TAIL(q, tp, o, 0);
                            for (int64_t l = c0; l < c1; l++) dst[l] += wv * cq[l];
                        }
                }
            }
            for (int i = 1; i < M; i++) {
                const int t = sc->causal ? i : M - 1 - i, tp = sc->causal ? t - 1 : t + 1;
                for (int r = 0; r < k; r++) {
                    float *dst = &TAIL(s, t, r, 0);
                    for (int o = 0; o < k; o++) {
                        const float av = d->A[s][r][o];
                        const float *cp = &TAIL(s, tp, o, 0);
                        for (int64_t l = c0; l < c1; l++) dst[l] += av * cp[l];
                    }
                }
            }
        }
    }

    /* ---- pass 2: tiles in parallel, every scan entering with the previous tile's completed carry ---- */
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nt) if (nt > 1)
#endif
    for (int64_t task = 0; task < outer * M * panels; task++) {
        const int64_t o = task / (M * panels), rem = task % (M * panels);
        const int t = (int)(rem / panels);
        const int64_t x0 = (rem % panels) * pw;
        const int w = (int)(inner - x0 < pw ? inner - x0 : pw);
        float buf[TP_MAXT * TP_PANEL], h[TP_MAXK * TP_PANEL];
        float *img = data + (size_t)o * n * inner + (size_t)t * T * inner + x0;
        for (int p = 0; p < T; p++) memcpy(buf + (size_t)p * w, img + (size_t)p * inner, sizeof(float) * w);
        for (int s = 0; s < ns; s++) {
            const tp_scan *sc = &d->sc[s];
            const int first = tp_is_first(sc, t, M);
            if (first) memset(h, 0, sizeof(float) * k * w);
            else {
                const int tp = sc->causal ? t - 1 : t + 1;
                for (int j = 0; j < k; j++) memcpy(h + (size_t)j * w, &TAIL(s, tp, j, o * inner + x0), sizeof(float) * w);
            }
            tp_scan_tile(buf, T, w, sc, k, h, d->clamp && first);
        }
        for (int p = 0; p < T; p++) memcpy(img + (size_t)p * inner, buf + (size_t)p * w, sizeof(float) * w);
    }
#undef TAIL
    free(tails);
    return 0;
}

/* Tiled counterpart of orc_apply_filter for f32 pixels.  tile[d] = tile width along dimension d (must divide the
 * extent, <= 256; 0 = that dimension is filtered untiled with orc_apply_scan). */
int orc_apply_filter_tiled_f32(float *data, int ndim, const int64_t *extent, const orc_scan *scans, int n_scans,
                               int border, const int *tile, int threads) {
    if (!data || !extent || !scans || !tile) return -1;
    if (ndim < 1 || ndim > ORC_MAX_DIMS) return -2;
    for (int dm = 0; dm < ndim; dm++) {
        tp_dim d;
        memset(&d, 0, sizeof(d));
        d.clamp = border == ORC_BORDER_CLAMP;
        for (int i = 0; i < n_scans; i++) {
            if (scans[i].dim != dm) continue;
            if (scans[i].order < 1 || scans[i].order > TP_MAXK || d.n_scans >= TP_MAXS) return -4;
            tp_scan *s = &d.sc[d.n_scans++];
            s->causal = scans[i].causal ? 1 : 0;
            s->order = scans[i].order;
            s->b = scans[i].coeff[0];
            for (int j = 0; j < s->order; j++) s->a[j] = scans[i].coeff[1 + j];
            if (s->order > d.k) d.k = s->order;
        }
        if (d.n_scans == 0) continue;
        int64_t inner = 1, outer = 1;
        for (int e = 0; e < dm; e++) inner *= extent[e];
        for (int e = dm + 1; e < ndim; e++) outer *= extent[e];
        const int T = tile[dm];
        if (T == 0) {
            for (int i = 0; i < n_scans; i++)
                if (scans[i].dim == dm) {
                    int rc = orc_apply_scan(data, ORC_F32, ndim, extent, &scans[i], border, threads);
                    if (rc) return rc;
                }
            continue;
        }
        if (T < d.k || T > TP_MAXT || extent[dm] % T != 0) return -6;
        d.T = T;
        if (tp_build_tables(&d)) return -7;
        int rc = tp_run_dim(data, extent[dm], inner, outer, &d, threads);
        free(d.W);
        if (rc) return rc;
    }
    return 0;
}
