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


def cuda_image(shape, dtype=np.float32, seed=1234, lo=None, hi=None):
    """random_image on the GPU: the values are drawn on the host from the seeded numpy generator and copied, so the input of
    a GPU test is the same on every box (a device-side torch.rand depends on the generator state the test inherits).
    Integers: uniform in [lo, hi) (default [0, 256))."""
    import torch
    dt = np.dtype(dtype)
    if np.issubdtype(dt, np.integer):
        img = np.random.default_rng(seed).integers(0 if lo is None else lo, 256 if hi is None else hi, size=shape).astype(dt)
    elif dt == np.float64:
        img = np.random.default_rng(seed).random(size=shape)
    else:
        img = random_image(shape, dt, seed)
    return torch.from_numpy(np.ascontiguousarray(img)).cuda()


def ones_image(shape, dtype=np.float32):
    """Generator (i): what the reference's generate_random_image really returns."""
    return np.ones(shape, dtype=dtype)


def _note(metric, value, out, ref, out_dtype=None):
    """Record mode (tests/parity_record.py, RF_RECORD_PARITY=<file>): every metric evaluation is written down with the oracle
    call its reference came from, so that tests/metric_margin.py can replay the assertion through the f32 oracle."""
    import parity_record
    parity_record.note_metric(metric, value, out, ref, out_dtype)
    return value


def rel_err_strict(out, ref):
    """SURVEY 8d's parity metric, literally: max over pixels of |out-ref| / max(|ref|, 1e-6)
    (the comparison of lib/recfilter.h:818-821 made relative)."""
    dt = getattr(out, "dtype", None)
    out = np.asarray(out, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    return _note("strict", float(np.max(np.abs(out - ref) / np.maximum(np.abs(ref), 1e-6))), out, ref, dt)


HIGHPASS_FLOOR = 1e-1


def rel_err_highpass_floor(out, ref):
    """Metric for HIGH-PASS results only: |out-ref| / max(|ref|, HIGHPASS_FLOOR = 10 % of the image's peak magnitude), i.e.
    an error of at most 1e-5 of the peak wherever the result is small.  (Rounds 2-5 used 1 %; tests/metric_margin.py showed the
    serial f32 reference operator itself at 2-6e-5 under that floor for four to twelve scans with negative lobes -- a margin
    of 2-4 against 1e-4, where VERDICT r5 asks for 10.)

    A high-pass filter such as the B-spline prefilter (apps/bspline) produces zero crossings, where a pointwise
    relative error is ill-conditioned for ANY f32 implementation (the reference's own f32 loops included) and would
    report rounding noise of 1e-7 absolute as 1e-3 "relative".  Not used where the result stays away from zero."""
    dt = getattr(out, "dtype", None)
    out = np.asarray(out, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    floor = max(HIGHPASS_FLOOR * float(np.max(np.abs(ref))), 1e-30)
    return _note("floor", float(np.max(np.abs(out - ref) / np.maximum(np.abs(ref), floor))), out, ref, dt)


def rel_err_scaled(out, ref, scale):
    """Metric for results that are SUMS OF TERMS which may cancel (a pointwise consumer `w_f F(x') + w_i x' + b`, an
    unsharp mask, a difference of filters): |out-ref| / max(scale, 1e-6) with `scale` = the sum of the terms' magnitudes,
    pixel by pixel.  One f32 rounding of an O(1) term is 6e-8 absolute whatever the terms cancel to; judged against the
    result alone (4.5e-4 where -0.7 F + 1.7 x' + 0.1 nearly cancels) it reads as 2e-4 "relative" for the f32 reference
    operator itself (tests/metric_margin.py measures the margin of every assertion)."""
    dt = getattr(out, "dtype", None)
    out = np.asarray(out, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    scale = np.broadcast_to(np.asarray(scale, dtype=np.float64), ref.shape)
    return _note("scaled", float(np.max(np.abs(out - ref) / np.maximum(scale, 1e-6))), out, ref, dt)


def has_zero_crossings(ref):
    """True when the reference result changes sign (or touches zero): the result of a high-pass filter, or of random
    coefficients with negative lobes.  (Rounds 2-5 chose the metric with this test; it does not see a result that keeps one sign
    and still cancels to 1e-3 of its operands at some pixel -- tests/metric_margin.py found 23 such assertions -- so rel_err no
    longer depends on it.  pointwise_scale still uses it for the filtered term of an epilogue.)"""
    r = np.asarray(ref, dtype=np.float64)
    lo, hi = float(r.min()), float(r.max())
    return (lo < 0.0 < hi) or float(np.abs(r).min()) < 1e-6


LOCAL_FLOOR = 0.1        # a sample counts with at least this fraction of the largest magnitude in its neighbourhood ...
LOCAL_RADIUS = 32        # ... of this many samples either side, along every axis (images and volumes: Gaussians of sigma 5)
LOCAL_RADIUS_1D = 256    # ... and for 1-D signals: the audio biquads (apps/audio) ring for hundreds of samples


def local_radius(ndim):
    return LOCAL_RADIUS_1D if ndim == 1 else LOCAL_RADIUS


def rel_err_local_floor(out, ref):
    """THE parity metric of the tests: max over samples of |out - ref| / max(|ref|, LOCAL_FLOOR x the largest |ref| within
    LOCAL_RADIUS samples along every axis (LOCAL_RADIUS_1D for 1-D signals), 1e-6).

    Where the result is smooth and keeps its sign -- every low-pass BASELINE config on the positive synthetic images: summed-area
    tables, Gaussians, B-splines of positive data, the 3-D filter -- a sample IS of its neighbourhood's magnitude and this is
    the strict pointwise relative error of SURVEY 8d (lib/recfilter.h:818-821 made relative), large dynamic ranges included (a
    summed-area table runs from 0.8 to 3e7 smoothly).  Where a result passes through or near zero between samples of ordinary
    size -- a high-pass filter, random coefficients with negative lobes, a centred input -- that sample is a sum of cancelling
    terms of its neighbours' size, and a relative error against the cancelled value is ill-conditioned for ANY f32
    implementation, the reference operator's own loops included: one rounding of an O(1) term is 6e-8 absolute, 2e-4 "relative"
    at a sample of 3e-4.  There the error is held to 1e-4 x 10 % of the LOCAL peak (not the image's: a response that grows
    across the image is judged by what surrounds the sample).  tests/metric_margin.py replays every recorded assertion of the
    GPU suite through the f32 oracle on 200 seeds to show the reference passes this metric with a margin."""
    from scipy.ndimage import maximum_filter
    dt = getattr(out, "dtype", None)
    out = np.asarray(out, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    mag = np.abs(ref)
    floor = LOCAL_FLOOR * maximum_filter(mag, size=2 * local_radius(mag.ndim) + 1, mode="nearest") if mag.size else mag
    return _note("local", float(np.max(np.abs(out - ref) / np.maximum(np.maximum(mag, floor), 1e-6))) if mag.size else 0.0, out, ref, dt)


def rel_err(out, ref, scale=None):
    """The parity metric the tests assert against 1e-4: rel_err_local_floor -- strict pointwise wherever the result is smooth
    and one-signed, floored at 10 % of the local peak where it cancels.  `scale` (the magnitude of the terms a pointwise stage
    adds up, see pointwise_want) selects rel_err_scaled instead."""
    if scale is not None:
        return rel_err_scaled(out, ref, scale)
    return rel_err_local_floor(out, ref)


def pointwise_want(img, scans, clamped, prologue=None, epilogue=None, apply_filter=None):
    """(want, scale) of a plan with pointwise stages (rf_pointwise_desc): x' = p0 x + p1 evaluated in the pixel type as the
    kernels do, F = the f64 oracle on x', want = e0 F + e1 x' + e2, scale = |e0 F| + |e1 x'| + |e2| (the terms' magnitudes:
    what rel_err(out, want, scale=scale) judges the error against).  Without an epilogue scale is None: the plain metric."""
    import oracle
    import parity_record
    x = img.astype(np.float64)
    if prologue is not None:
        x = (np.float32(prologue[0]) * img + np.float32(prologue[1])).astype(np.float64) if img.dtype == np.float32 \
            else prologue[0] * x + prologue[1]
    f = (apply_filter or oracle.apply_filter)(x, scans, clamped)
    if epilogue is None:
        return f, None
    parity_record.note_epilogue(epilogue, x)
    e0, e1, e2 = (float(v) for v in epilogue)
    return e0 * f + e1 * x + e2, pointwise_scale(f, x, epilogue)


def pointwise_scale(f, x, epilogue):
    """|e0 F| + |e1 x'| + |e2|, pixel by pixel.  Where F itself changes sign (a prologue that centres the input, a high-pass
    filter) its small values are sums of large cancelling contributions, ill-conditioned like any zero crossing: there the
    filtered term counts with at least HIGHPASS_FLOOR of its peak, the floor rel_err_highpass_floor gives such an F on its own."""
    e0, e1, e2 = (float(v) for v in epilogue)
    tf = np.abs(e0 * f)
    if has_zero_crossings(f):
        tf = np.maximum(tf, HIGHPASS_FLOOR * float(tf.max()))
    return tf + np.abs(e1 * x) + abs(e2)
