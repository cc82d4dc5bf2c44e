/*
 * recfilter_oracle.h -- CPU restatement of the mit-gfx/recfilter scan operator.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under recfilter_amd/ (the product) may
 * include, link or call this file.  It is used by tests/, by
 * __graft_entry__.smoke() as the checker and by bench.py's cpu_baseline leg.
 *
 * What it restates (citations are into /root/reference):
 *   - the scan operator appended by RecFilter::add_filter
 *         lib/recfilter.cpp:302-343   (causal / anticausal index, zero border via
 *                                      select(rx>j, ..), clamped border via
 *                                      max(..,0) / min(..,width-1), coefficients
 *                                      cast to the pixel type)
 *   - the coefficient helpers
 *         lib/iir_coeff.cpp:38-63,83-85,103-159,162-177 (gaussian_weights)
 *         lib/iir_coeff.cpp:222-234   (integral_image_coeff)
 *         lib/iir_coeff.cpp:236-263   (overlap_feedback_coeff)
 *         lib/iir_coeff.cpp:205-220   (gaussian_box_filter)
 *   - the tile propagation matrices (used to pin the product's plan tables)
 *         lib/coefficients.cpp:8-49   (matrix_B)
 *         lib/coefficients.cpp:51-83  (matrix_R)
 *   - the error metric of the reference's own tests
 *         lib/recfilter.h:818-821     (CheckResult, percent relative error)
 *
 * Pinning: the reference cannot be built here (every translation unit includes
 * <Halide.h>; the Halide fork is an empty, unpinned submodule), so oracle/_ref
 * does not exist.  The oracle is pinned against the known answers of the
 * reference's own tests (the hand-written loops in tests/test_*.cpp, restated
 * independently in tests/ref_loops.py) and the coefficient values quoted in
 * SURVEY.md section 8 (a-3, a-14), see tests/test_oracle_golden.py.
 */
#ifndef RECFILTER_ORACLE_H
#define RECFILTER_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_F32 = 0, ORC_F64 = 1, ORC_I32 = 2, ORC_I16 = 3 };
enum { ORC_BORDER_ZERO = 0, ORC_BORDER_CLAMP = 1 };
enum { ORC_MAX_DIMS = 4 };
enum { ORC_MAX_ORDER = 32 };   /* the reference takes any order (lib/recfilter.cpp:260-343); its own sweep stops at 29
                               * (apps/audio/audio_filter_high_order.cpp:14,38) */

/* One scan = one RecFilter::add_filter call.
 * coeff[0] = feedforward, coeff[1..order] = feedback (added, not subtracted). */
typedef struct {
    int    dim;        /* 0 = x (fastest varying), 1 = y, 2 = z, 3 = w            */
    int    causal;     /* 1: +dim, 0: -dim                                          */
    int    order;      /* number of feedback coefficients                           */
    float  coeff[ORC_MAX_ORDER + 1];  /* feedforward + feedback, as floats like the reference */
} orc_scan;

/* Apply ONE scan in place to a dense x-fastest array (lib/recfilter.cpp:302-343).
 * threads <= 1: serial; otherwise OpenMP over independent lines (results are
 * identical: lines never interact). Returns 0 on success. */
int orc_apply_scan(void *data, int dtype, int ndim, const int64_t *extent,
                   const orc_scan *scan, int border, int threads);

/* Apply a list of scans in call order (successive in-place passes). */
int orc_apply_filter(void *data, int dtype, int ndim, const int64_t *extent,
                     const orc_scan *scans, int n_scans, int border, int threads);

/* Tiled CPU counterpart (recfilter_cpu_tiled.c; lib/recfilter.cpp:610-678 over the algorithm of lib/split.cpp): f32
 * pixels, tile[d] = tile width along dimension d (divides the extent, <= 256; 0 = untiled).  Same result as
 * orc_apply_filter up to f32 rounding. */
int orc_apply_filter_tiled_f32(float *data, int ndim, const int64_t *extent, const orc_scan *scans, int n_scans,
                               int border, const int *tile, int threads);

/* Coefficient helpers (lib/iir_coeff.cpp). out must hold order+1 floats. */
void orc_gaussian_weights(float sigma, int order, float *out);
void orc_integral_image_coeff(int n, float *out);
/* c has na+nb entries on return */
void orc_overlap_feedback_coeff(const float *a, int na, const float *b, int nb, float *c);
int  orc_gaussian_box_filter(int k, float sigma);

/* Tile matrices (lib/coefficients.cpp). feedback has `order` entries.
 * B: tile x tile, B[row*tile + col], row = output position, col = input position.
 * R: tile x order, R[row*order + j]. */
void orc_matrix_B(float feedfwd, const float *feedback, int order, int tile,
                  int clamp_border, float *B);
void orc_matrix_R(const float *feedback, int order, int tile, float *R);

/* Reference test metric (lib/recfilter.h:818-821): percent relative error,
 * 100*|ref-out|/(ref+1e-9); returns max, stores mean. f32 inputs. */
double orc_check_result_f32(const float *ref, const float *out, size_t n, double *mean_pct);

/* Number of OpenMP threads the build can use (1 if built without OpenMP). */
int orc_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
