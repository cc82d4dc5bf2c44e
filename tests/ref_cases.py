"""Filter configurations of the reference's own tests and apps, as data.

Every entry restates the *configuration* (shape, tile, scans, border, dtype) of
one file under /root/reference (cited per entry); the loop references those
files print their results against are restated independently in ref_loops.py.

A scan is (dim, causal, [feedfwd, fb1..fbk]) with dim 0 = x (fastest axis).
Shapes are numpy shapes, i.e. (..., z, y, x).
"""
from __future__ import annotations

import math

import numpy as np

X, Y, Z = 0, 1, 2
C, A = True, False  # causal (+dim) / anticausal (-dim)


def _w(rows):
    return [list(map(float, r)) for r in rows]


REFERENCE_TESTS = {
    # tests/test_trivial.cpp:8-22 -- 20x20, tile 4, summed-area table
    "test_trivial": dict(shape=(20, 20), tile=4, dtype=np.float32, clamped=False,
                         scans=[(X, C, [1.0, 1.0]), (Y, C, [1.0, 1.0])]),
    # tests/test_type_invariance.cpp:13-34 -- int16, second-order integral
    "test_type_invariance": dict(shape=(20, 20), tile=4, dtype=np.int16, clamped=False,
                                 scans=[(X, C, [1.0, 1.0, -1.0]), (Y, C, [1.0, 1.0, -1.0])]),
    # tests/test_repeated_causal.cpp:12-41 -- 20x1, four causal order-3 scans
    "test_repeated_causal": dict(shape=(1, 20), tile=4, dtype=np.float32, clamped=False,
                                 scans=[(X, C, [1.0, 1.00, 0.250, 0.0625]),
                                        (X, C, [1.0, 0.75, 0.500, 0.0625]),
                                        (X, C, [1.0, 0.50, 0.250, 0.0625]),
                                        (X, C, [1.0, 0.25, 0.125, 0.0625])]),
    # tests/test_repeated_anticausal.cpp:12-40 -- 20x1, four anticausal order-2 scans
    "test_repeated_anticausal": dict(shape=(1, 20), tile=4, dtype=np.float32, clamped=False,
                                     scans=[(X, A, [1.0, 0.700, 0.5000]),
                                            (X, A, [1.0, 0.500, 0.5000]),
                                            (X, A, [1.0, 0.250, 0.1250]),
                                            (X, A, [1.0, 0.125, 0.0625])]),
    # tests/test_causal_anticausal.cpp:12-40 -- the W(i,1) double assignment at :24-27
    # leaves W(i,1)=0.0625 and W(i,2)=0 (Image<float> is zero-initialised)
    "test_causal_anticausal": dict(shape=(1, 20), tile=4, dtype=np.float32, clamped=False,
                                   scans=[(X, C, [1.0, 0.5, 0.0625, 0.0]),
                                          (X, A, [1.0, 0.5, 0.0625, 0.0]),
                                          (X, C, [1.0, 0.5, 0.0625, 0.0]),
                                          (X, A, [1.0, 0.5, 0.0625, 0.0])]),
    # tests/test_causal_xy.cpp:12-42
    "test_causal_xy": dict(shape=(16, 16), tile=4, dtype=np.float32, clamped=False,
                           scans=[(X, C, [1.0, 0.5, 0.500, 0.12500]),
                                  (X, C, [1.0, 0.5, 0.250, 0.12500]),
                                  (Y, C, [1.0, 0.5, 0.125, 0.06250]),
                                  (Y, C, [1.0, 0.5, 0.125, 0.03125])]),
    # tests/test_causal_anticausal_xy.cpp:12-42
    "test_causal_anticausal_xy": dict(shape=(16, 16), tile=4, dtype=np.float32, clamped=False,
                                      scans=[(X, C, [1.0, 0.5, 0.500, 0.12500]),
                                             (X, A, [1.0, 0.5, 0.250, 0.12500]),
                                             (Y, C, [1.0, 0.5, 0.125, 0.06250]),
                                             (Y, A, [1.0, 0.5, 0.125, 0.03125])]),
    # tests/test_generic_xy.cpp:12-45 -- seven order-2 scans, uneven per dimension
    "test_generic_xy": dict(shape=(16, 16), tile=4, dtype=np.float32, clamped=False,
                            scans=[(X, C, [1.0, 0.5, 0.2500]),
                                   (X, A, [1.0, 0.5, 0.1250]),
                                   (X, C, [1.0, 0.5, 0.0625]),
                                   (X, A, [1.0, 0.5, 0.1250]),
                                   (Y, C, [1.0, 0.5, 0.2500]),
                                   (Y, A, [1.0, 0.5, 0.0625]),
                                   (Y, A, [1.0, 0.5, 0.1250])]),
    # tests/test_generic_xyz.cpp:12-47 -- 16^3, six order-2 scans
    "test_generic_xyz": dict(shape=(16, 16, 16), tile=4, dtype=np.float32, clamped=False,
                             scans=[(X, C, [1.0, 0.5, 0.2500]),
                                    (X, A, [1.0, 0.5, 0.1250]),
                                    (Y, C, [1.0, 0.5, 0.0625]),
                                    (Y, A, [1.0, 0.5, 0.1250]),
                                    (Z, C, [1.0, 0.5, 0.2500]),
                                    (Z, A, [1.0, 0.5, 0.0625])]),
}

# Known answers on the reference's all-ones input (lib/recfilter.h:695-696 makes
# generate_random_image return T(1) everywhere), computed from the tests' own loop
# semantics -- SURVEY.md section 4.  (first, last, centre, sum); "first" is index 0
# in every dim, "last" the max index, "centre" size/2 in every dim.
ANCHORS_ALL_ONES = {
    "test_trivial": (1.0, 400.0, 121.0, 44100.0),
    "test_generic_xy": (347.6587, 50.4491, 991.6123, 158557.147),
    "test_generic_xyz": (71.32104, 34.26548, 420.9293, 1014701.97),
    "test_causal_xy": (1.0, 554.205, 151.628, 37562.4724),
    "test_causal_anticausal_xy": (177.4516, 56.49136, 378.6183, 67917.1479),
    "test_repeated_causal": (1.0, 7502.124, 469.789, 29077.8507),
    "test_repeated_anticausal": (511.1182, 1.0, 72.27883, 2866.10571),
    "test_causal_anticausal": (13.24756, 7.571615, 26.47968, 436.836525),
}

# The apps' own check loops (apps/summed_table/summed_table.cpp:66-84; apps/bspline/bicubic_filter.cpp:109-158 and the
# identical loops of biquintic_cascaded_filter.cpp:143-190 / biquintic_overlapped_filter.cpp with the apps' coefficients
# {1+a, -a} and {1+a, -a, 0.1}, a = 2 - sqrt 3), evaluated in float32 as the apps do on a 24 x 32 image: all ones (what the
# reference's generate_random_image returns) and the exactly representable ramp (x % 7) + 2 (y % 5).  (first, last,
# centre, sum).  The apps apply +x, +y, -x, -y; the filter is +x, -x, +y, -y -- x and y commute up to rounding.
APP_ANCHORS = {
    ("summed_table", "ones"): (1.0, 768.0, 221.0, 158400.0),
    ("summed_table", "ramp"): (0.0, 5104.0, 1367.0, 1016400.0),
    ("bicubic", "ones"): (1.0, 1.0, 1.0, 768.0),
    ("bicubic", "ramp"): (-0.8462355136871338, 9.72153091430664, 5.829268932342529, 5098.764333099127),
    ("biquintic", "ones"): (1.435629963874817, 1.426071047782898, 1.3890289068222046, 1068.92553794384),
    ("biquintic", "ramp"): (-0.17553919553756714, 13.785110473632812, 7.76296329498291, 7105.855566650629),
}

# Coefficient known answers, values of the reference's iir_coeff.cpp quoted in
# SURVEY.md section 8 (a-14).
GAUSS_SIGMA5 = {
    1: [0.2320382, 0.7679618],
    2: [0.0975842401, 1.5283848, -0.625968993],
    3: [0.0226432718, 2.29634666, -1.79971004, 0.480720282],
}
INTEGRAL_COEFF = {1: [1, 1], 2: [1, 2, -1], 3: [1, 3, -3, 1]}
# matrix_R example of SURVEY a-3: a = {.5, .25}, T = 4
MATRIX_R_EXAMPLE = dict(feedback=[0.5, 0.25], tile=4,
                        rows=[[0.5, 0.25], [0.5, 0.125], [0.375, 0.125], [0.3125, 0.09375]])
# untiled f32 result of the cfg-3 filter on 64x64 default_rng(1234).random(float32)
# (SURVEY.md section 8c): first, last, centre, sum
CFG3_RANDOM64 = (0.6977313, 0.5212882, 0.4841482, 2070.12479)

_a = 2.0 - math.sqrt(3.0)
BICUBIC_COEFF = [1.0 + _a, -_a]                     # apps/bspline/bicubic_filter.cpp:36-37
GAUSS2 = [0.0975842401, 1.5283848, -0.625968993]    # gaussian_weights(5, 2)
GAUSS3 = [0.0226432718, 2.29634666, -1.79971004, 0.480720282]


def xy_pm(coeff):
    """+x, -x, +y, -y with one coefficient vector (the shape of the Gaussian / B-spline apps)."""
    return [(X, C, list(coeff)), (X, A, list(coeff)), (Y, C, list(coeff)), (Y, A, list(coeff))]


# BASELINE.json configs in their concrete form.  Sizes here are the
# full ones; tests shrink them and the bench uses them as they are.
BASELINE_CONFIGS = {
    "cfg1_prefix_sum_1d": dict(shape=(4096,), dtype=np.float32, clamped=False,
                               scans=[(X, C, [1.0, 1.0])]),
    "cfg2_summed_table": dict(shape=(8192, 8192), dtype=np.float32, clamped=False,
                              scans=[(X, C, [1.0, 1.0]), (Y, C, [1.0, 1.0])]),
    "cfg3_gaussian2_xy": dict(shape=(16384, 16384), dtype=np.float32, clamped=True,
                              scans=xy_pm(GAUSS2)),
    "cfg4a_bicubic_rgb": dict(shape=(16384, 16384), planes=3, dtype=np.float32, clamped=True,
                              scans=xy_pm(BICUBIC_COEFF)),
    "cfg4b_gaussian3_rgb": dict(shape=(16384, 16384), planes=3, dtype=np.float32, clamped=True,
                                scans=xy_pm(GAUSS3)),
    "cfg5_generic_xyz": dict(shape=(2048, 2048, 2048), dtype=np.float32, clamped=False,
                             scans=REFERENCE_TESTS["test_generic_xyz"]["scans"]),
}


# shapes the fused x/y path accepts (width % 256 == 0, height % 32 == 0), small enough for the CPU emulator
FUSED_CASES = {
    "gauss2_clamped": dict(shape=(128, 512), scans=xy_pm(GAUSS2), clamped=True),
    "gauss3_clamped": dict(shape=(64, 768), scans=xy_pm(GAUSS3), clamped=True),
    "bicubic_clamped_ty32": dict(shape=(96, 512), scans=xy_pm(BICUBIC_COEFF), clamped=True),
    "generic_xy_zero": dict(shape=(128, 512), scans=REFERENCE_TESTS["test_generic_xy"]["scans"], clamped=False),
    "sat": dict(shape=(128, 256), scans=[(0, True, [1.0, 1.0]), (1, True, [1.0, 1.0])], clamped=False),
    "x_only": dict(shape=(64, 512), scans=[(0, True, [0.5, 0.4, -0.1]), (0, False, [0.5, 0.4, -0.1])], clamped=True),
    "y_only": dict(shape=(128, 256), scans=[(1, False, [0.5, 0.4, -0.1]), (1, True, [0.5, 0.4])], clamped=True),
    "single_tile": dict(shape=(64, 256), scans=xy_pm(GAUSS2), clamped=True),
    # widths that are not multiples of 256: the last tile of every row is partial
    "partial_gauss2_clamped": dict(shape=(64, 320), scans=xy_pm(GAUSS2), clamped=True),
    "partial_gauss3_clamped": dict(shape=(96, 464), scans=xy_pm(GAUSS3), clamped=True),
    "partial_mixed_zero": dict(shape=(64, 528), scans=REFERENCE_TESTS["test_generic_xy"]["scans"], clamped=False),
    "partial_single_tile": dict(shape=(32, 48), scans=xy_pm(GAUSS2), clamped=True),
    "partial_x_only_causal_then_anti": dict(shape=(64, 272), scans=[(0, True, [0.5, 0.4, -0.1]), (0, False, [0.6, 0.3]),
                                                                    (0, True, [0.9, 0.05]), (0, False, [0.5, 0.4, -0.1])], clamped=True),
    "partial_sat": dict(shape=(64, 1936), scans=[(0, True, [1.0, 1.0]), (1, True, [1.0, 1.0])], clamped=False),
    # widths that are multiples of 4 only: the scan enters inside a 16-sample segment
    "partial_w4_gauss2_clamped": dict(shape=(64, 300), scans=xy_pm(GAUSS2), clamped=True),
    "partial_w4_gauss3_clamped": dict(shape=(50, 1000), scans=xy_pm(GAUSS3), clamped=True),
    "partial_w4_mixed_zero": dict(shape=(33, 20), scans=REFERENCE_TESTS["test_generic_xy"]["scans"], clamped=False),
    "partial_w4_x_scans": dict(shape=(32, 268), scans=[(0, True, [0.5, 0.4, -0.1]), (0, False, [0.6, 0.3]),
                                                       (0, True, [0.9, 0.05]), (0, False, [0.5, 0.4, -0.1])], clamped=True),
    # widths that are not multiples of 4: rows are only element-aligned and end in a partial 16-byte chunk
    "odd_w_gauss2_clamped": dict(shape=(64, 301), scans=xy_pm(GAUSS2), clamped=True),
    "odd_w_gauss3_clamped": dict(shape=(50, 1001), scans=xy_pm(GAUSS3), clamped=True),
    "odd_w_mixed_zero": dict(shape=(33, 21), scans=REFERENCE_TESTS["test_generic_xy"]["scans"], clamped=False),
    "odd_w_x_scans": dict(shape=(32, 270), scans=[(0, True, [0.5, 0.4, -0.1]), (0, False, [0.6, 0.3]),
                                                  (0, True, [0.9, 0.05]), (0, False, [0.5, 0.4, -0.1])], clamped=True),
    "odd_w_one_past_a_tile": dict(shape=(70, 257), scans=xy_pm(GAUSS2), clamped=True),
    "odd_w_tiny": dict(shape=(3, 1), scans=xy_pm(GAUSS2), clamped=True),
    # heights that are not multiples of 32: the last tile row is partial
    "partial_y_gauss2_clamped": dict(shape=(100, 512), scans=xy_pm(GAUSS2), clamped=True),
    "partial_xy_gauss3_clamped": dict(shape=(135, 240), scans=xy_pm(GAUSS3), clamped=True),
    "partial_y_mixed_zero": dict(shape=(77, 256), scans=REFERENCE_TESTS["test_generic_xy"]["scans"], clamped=False),
    "partial_y_tiny": dict(shape=(5, 32), scans=xy_pm(GAUSS2), clamped=True),
    "partial_y_only_scans": dict(shape=(70, 272), scans=[(1, True, [0.5, 0.4, -0.1]), (1, False, [0.6, 0.3]),
                                                         (1, True, [0.9, 0.05]), (1, False, [0.5, 0.4, -0.1])], clamped=True),
}



def random_image(shape, dtype=np.float32, seed=1234):
    """Generator (ii) of SURVEY 8d: fixed-seed uniform [0,1) (ints: [0,255])."""
    rng = np.random.default_rng(seed)
    if np.issubdtype(np.dtype(dtype), np.integer):
        return rng.integers(0, 256, size=shape).astype(dtype)
    return rng.random(size=shape, dtype=np.float32).astype(dtype)


def ones_image(shape, dtype=np.float32):
    """Generator (i): what the reference's generate_random_image really returns."""
    return np.ones(shape, dtype=dtype)


def rel_err_strict(out, ref):
    """SURVEY 8d's parity metric, literally: max over pixels of |out-ref| / max(|ref|, 1e-6)
    (the comparison of lib/recfilter.h:818-821 made relative)."""
    out = np.asarray(out, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    return float(np.max(np.abs(out - ref) / np.maximum(np.abs(ref), 1e-6)))


def rel_err_highpass_floor(out, ref):
    """Metric for HIGH-PASS results only: |out-ref| / max(|ref|, 1 % of the image's peak magnitude).

    A high-pass filter such as the B-spline prefilter (apps/bspline) produces zero crossings, where a pointwise
    relative error is ill-conditioned for ANY f32 implementation (the reference's own f32 loops included) and would
    report rounding noise of 1e-7 absolute as 1e-3 "relative".  Not used where the result stays away from zero."""
    out = np.asarray(out, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    floor = max(1e-2 * float(np.max(np.abs(ref))), 1e-30)
    return float(np.max(np.abs(out - ref) / np.maximum(np.abs(ref), floor)))


def has_zero_crossings(ref):
    """True when the reference result changes sign (or touches zero): the result of a high-pass filter, or of random
    coefficients with negative lobes, where only the floored metric is meaningful.  A large dynamic range alone (a
    summed-area table runs from 0.8 to 3e7) is NOT a reason to leave the strict metric."""
    r = np.asarray(ref, dtype=np.float64)
    lo, hi = float(r.min()), float(r.max())
    return (lo < 0.0 < hi) or float(np.abs(r).min()) < 1e-6


def rel_err(out, ref):
    """The parity metric the tests assert against 1e-4: STRICT pointwise (rel_err_strict) wherever the reference
    result stays away from zero -- every low-pass BASELINE config (summed-area tables, Gaussians, the 3-D filter) on
    the positive synthetic images -- and the floored high-pass metric only for results with zero crossings."""
    return rel_err_highpass_floor(out, ref) if has_zero_crossings(ref) else rel_err_strict(out, ref)
