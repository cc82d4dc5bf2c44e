"""Pin the CPU oracle (oracle/recfilter_oracle.c) before anything trusts it.

Sources of truth, all from the reference itself:
  * the loop references inside its tests/apps, restated independently in ref_loops.py;
  * the known answers those loops give on the reference's all-ones input;
  * coefficient / matrix values of lib/iir_coeff.cpp and lib/coefficients.cpp.
"""
import numpy as np
import pytest

import oracle
import ref_cases as rc
import ref_loops


def _anchors(arr):
    centre = tuple(s // 2 for s in arr.shape)
    return (float(arr.flat[0]), float(arr.flat[-1]), float(arr[centre]), float(arr.sum(dtype=np.float64)))


@pytest.mark.parametrize("name", sorted(rc.ANCHORS_ALL_ONES))
def test_known_answers_all_ones(name):
    case = rc.REFERENCE_TESTS[name]
    out = oracle.apply_filter(rc.ones_image(case["shape"], case["dtype"]), case["scans"], case["clamped"])
    got = _anchors(out)
    np.testing.assert_allclose(got, rc.ANCHORS_ALL_ONES[name], rtol=2e-6)


@pytest.mark.parametrize("name", sorted(rc.REFERENCE_TESTS))
@pytest.mark.parametrize("gen", ["ones", "random"])
def test_oracle_matches_reference_test_loops(name, gen):
    case = rc.REFERENCE_TESTS[name]
    img = rc.ones_image(case["shape"], case["dtype"]) if gen == "ones" else rc.random_image(case["shape"], case["dtype"])
    got = oracle.apply_filter(img, case["scans"], case["clamped"])
    want = ref_loops.zero_border_loops(img, case["scans"])
    if np.issubdtype(img.dtype, np.integer):
        np.testing.assert_array_equal(got, want)          # bit-exact for integer pixel types
    else:
        assert rc.rel_err(got, want) < 2e-6


def test_trivial_closed_form():
    # tests/test_trivial.cpp: SAT of ones = (x+1)(y+1)
    case = rc.REFERENCE_TESTS["test_trivial"]
    out = oracle.apply_filter(rc.ones_image(case["shape"]), case["scans"])
    yy, xx = np.mgrid[0:20, 0:20]
    np.testing.assert_array_equal(out, ((xx + 1) * (yy + 1)).astype(np.float32))


@pytest.mark.parametrize("dtype", [np.float32, np.int32])
def test_summed_table_app_loops(dtype):
    img = rc.random_image((24, 40), dtype)
    scans = rc.BASELINE_CONFIGS["cfg2_summed_table"]["scans"]
    got = oracle.apply_filter(img, scans)
    want = ref_loops.summed_table_loops(img)
    if dtype is np.int32:
        np.testing.assert_array_equal(got, want)
        np.testing.assert_array_equal(got, img.cumsum(axis=1).cumsum(axis=0))
    else:
        assert rc.rel_err(got, want) < 1e-6


@pytest.mark.parametrize("coeff", [rc.BICUBIC_COEFF, rc.GAUSS2])
def test_clamped_app_loops(coeff):
    # apps/bspline/bicubic_filter.cpp:124-156 applies +x,+y,-x,-y; the filter is +x,-x,+y,-y
    img = rc.random_image((20, 28))
    got = oracle.apply_filter(img, rc.xy_pm(coeff), clamped=True)
    want = ref_loops.clamped_xy_loops(img, coeff)
    assert rc.rel_err(got, want) < 5e-5   # scan order differs (x/y commute up to rounding)


@pytest.mark.parametrize("app,image", sorted(rc.APP_ANCHORS))
def test_app_check_loop_anchors(app, image):
    """The apps' own check loops as anchors (ref_cases.APP_ANCHORS): the oracle on the apps' filters and the stored values
    of those loops; the loops themselves (ref_loops.py) are re-run as well, so a drift of either side shows."""
    yy, xx = np.mgrid[0:24, 0:32]
    img = np.ones((24, 32), np.float32) if image == "ones" else ((xx % 7) + 2 * (yy % 5)).astype(np.float32)
    if app == "summed_table":
        scans, clamped = rc.BASELINE_CONFIGS["cfg2_summed_table"]["scans"], False
        loops = ref_loops.summed_table_loops(img)
    else:
        coeff = list(rc.BICUBIC_COEFF) + ([0.1] if app == "biquintic" else [])     # biquintic_cascaded_filter.cpp:48
        scans, clamped = rc.xy_pm(coeff), True
        loops = ref_loops.clamped_xy_loops(img, coeff)
    want = rc.APP_ANCHORS[(app, image)]
    np.testing.assert_allclose(_anchors(loops), want, rtol=1e-6, atol=1e-6)
    got = oracle.apply_filter(img, scans, clamped)
    np.testing.assert_allclose(_anchors(got), want, rtol=2e-5, atol=2e-5)
    assert rc.rel_err(got, loops) < 5e-5


def _lfilter_scan(x, causal, coeff):
    """One zero-border scan y[i] = b x[i] + sum_j a_j y[i -/+ (j+1)] with scipy.signal.lfilter: transfer function
    b / (1 - a_1 z^-1 - ... - a_k z^-k), anticausal = the causal filter on the flipped line."""
    from scipy.signal import lfilter
    b, a = [float(coeff[0])], [1.0] + [-float(c) for c in coeff[1:]]
    return lfilter(b, a, x, axis=-1) if causal else lfilter(b, a, x[..., ::-1], axis=-1)[..., ::-1]


@pytest.mark.parametrize("seed", range(8))
def test_zero_border_scans_against_scipy_lfilter(seed):
    """An implementation nobody here wrote: scipy.signal.lfilter (direct-form IIR, zero initial state) is the reference's
    zero-border scan (lib/recfilter.cpp:337-340: taps before the border contribute 0).  Random f64 lines, orders 1..6,
    causal and anticausal, scans chained in place as add_filter chains them, along x and along y."""
    rng = np.random.default_rng(9000 + seed)
    img = rng.standard_normal((int(rng.integers(1, 9)), int(rng.integers(1, 200))))
    scans, want = [], img.copy()
    for _ in range(int(rng.integers(1, 6))):
        k = int(rng.integers(1, 7))
        poles = rng.uniform(-0.85, 0.85, size=k)
        coeff = [float(rng.uniform(0.2, 1.5))] + [float(-c) for c in np.poly(poles)[1:]]      # stable: poles inside the unit circle
        dim, causal = int(rng.integers(0, 2)), bool(rng.integers(0, 2))
        scans.append((dim, causal, coeff))
        if dim == 0:
            want = _lfilter_scan(want, causal, coeff)
        else:
            want = _lfilter_scan(want.T, causal, coeff).T
    # (f64 pixels: the oracle keeps the coefficients as the C ABI delivers them, float32 -- so does the comparison)
    scans32 = [(d, c, [float(np.float32(v)) for v in co]) for d, c, co in scans]
    want32 = img.copy()
    for d, c, co in scans32:
        want32 = _lfilter_scan(want32, c, co) if d == 0 else _lfilter_scan(want32.T, c, co).T
    got = oracle.apply_filter(img, scans32, clamped=False)
    assert np.max(np.abs(got - want32)) <= 1e-9 * max(1.0, float(np.max(np.abs(want32))))


def test_clamped_scan_against_lfilter_with_initial_state():
    """The clamped border (lib/recfilter.cpp:330-336) through lfilter as well: the scan reads the partially updated
    buffer, so at r = 0 every tap reads the old f[0] -- y[0] = (b + sum a) x[0] -- and from r = 1 on the taps beyond the
    border read y[0].  That is lfilter on x[1:] with the initial condition "all previous outputs equal y[0]"."""
    from scipy.signal import lfilter, lfiltic
    rng = np.random.default_rng(77)
    for k in (1, 2, 3):
        coeff = [0.4] + [float(-c) for c in np.poly(rng.uniform(-0.8, 0.8, size=k))[1:]]
        coeff = [float(np.float32(v)) for v in coeff]
        x = rng.standard_normal(150)
        b, a = [coeff[0]], [1.0] + [-c for c in coeff[1:]]
        y0 = (coeff[0] + sum(coeff[1:])) * x[0]
        zi = lfiltic(b, a, y=[y0] * k)
        rest, _ = lfilter(b, a, x[1:], zi=zi)
        want = np.concatenate([[y0], rest])
        got = oracle.apply_filter(x, [(0, True, coeff)], clamped=True)
        assert np.max(np.abs(got - want)) < 1e-10
        got_rev = oracle.apply_filter(x[::-1].copy(), [(0, False, coeff)], clamped=True)
        assert np.max(np.abs(got_rev[::-1] - want)) < 1e-10


def test_clamped_constant_is_fixed_point():
    # any clamped filter with b + sum(a) = 1 maps constants to themselves
    img = np.full((16, 32), 3.0, dtype=np.float64)
    out = oracle.apply_filter(img, rc.xy_pm([float(c) for c in rc.GAUSS2]), clamped=True)
    assert np.max(np.abs(out - 3.0)) < 1e-5


def test_cfg3_random64_anchor():
    img = np.random.default_rng(1234).random((64, 64), dtype=np.float32)
    out = oracle.apply_filter(img, rc.xy_pm(rc.GAUSS2), clamped=True)
    np.testing.assert_allclose(_anchors(out), rc.CFG3_RANDOM64, rtol=3e-6)


@pytest.mark.parametrize("order", [1, 2, 3])
def test_gaussian_weights(order):
    np.testing.assert_allclose(oracle.gaussian_weights(5.0, order), rc.GAUSS_SIGMA5[order], rtol=3e-7)


@pytest.mark.parametrize("n", [1, 2, 3])
def test_integral_image_coeff(n):
    np.testing.assert_array_equal(oracle.integral_image_coeff(n), np.array(rc.INTEGRAL_COEFF[n], dtype=np.float32))


def test_overlap_feedback_coeff():
    np.testing.assert_allclose(oracle.overlap_feedback_coeff([2, -1], [1]), [3, -3, 1])


def test_gaussian_box_filter():
    assert oracle.gaussian_box_filter(3, 1.0) == 2
    assert oracle.gaussian_box_filter(3, 2.0) == 4


def test_matrix_R_example():
    ex = rc.MATRIX_R_EXAMPLE
    np.testing.assert_allclose(oracle.matrix_R(ex["feedback"], ex["tile"]), ex["rows"])


def test_matrix_B_is_the_scan():
    # B (lib/coefficients.cpp:8-49) applied to a tile == the zero-border scan on that tile
    fb = [0.5, 0.25, 0.125]
    B = oracle.matrix_B(0.7, fb, 8)
    x = rc.random_image((8,))
    want = oracle.apply_filter(x, [(0, True, [0.7] + fb)])
    np.testing.assert_allclose(B @ x, want, rtol=1e-5)


def test_overlap_filter_order():
    # tests/test_overlap_filter_order.cpp:21-33: cascade of two filters == one higher-order filter
    img = rc.random_image((12, 12))
    f1 = [(0, True, [1.0, 2.0, -1.0]), (1, True, [1.0, 1.0])]
    f2 = [(0, True, [1.0, 1.0]), (1, True, [1.0, 2.0, -1.0])]
    cascaded = oracle.apply_filter(oracle.apply_filter(img, f1), f2)
    cx = oracle.overlap_feedback_coeff([2.0, -1.0], [1.0])
    cy = oracle.overlap_feedback_coeff([1.0], [2.0, -1.0])
    overlapped = oracle.apply_filter(img, [(0, True, [1.0] + list(cx)), (1, True, [1.0] + list(cy))])
    assert rc.rel_err(overlapped, cascaded) < 1e-5


def test_threads_do_not_change_results():
    img = rc.random_image((9, 33, 70))
    scans = rc.REFERENCE_TESTS["test_generic_xyz"]["scans"]
    a = oracle.apply_filter(img, scans, threads=1)
    b = oracle.apply_filter(img, scans, threads=4)
    np.testing.assert_array_equal(a, b)


def test_check_result_metric():
    ref = np.array([1.0, 2.0, 4.0], dtype=np.float32)
    out = np.array([1.0, 2.2, 4.0], dtype=np.float32)
    mx, mean = oracle.check_result(ref, out)
    assert abs(mx - 10.0) < 1e-3 and abs(mean - 10.0 / 3) < 1e-3


# ---- the tiled CPU counterpart (oracle/recfilter_cpu_tiled.c: bench.py's second cpu_baseline entry) ----------------
@pytest.mark.parametrize("name", sorted(n for n, c in rc.REFERENCE_TESTS.items() if c["dtype"] == np.float32))
def test_tiled_cpu_counterpart_matches_the_untiled_oracle_on_the_reference_tests(name):
    c = rc.REFERENCE_TESTS[name]
    img = rc.random_image(c["shape"], np.float32, 5)
    want = oracle.apply_filter(img.astype(np.float64), c["scans"], c["clamped"])
    got = oracle.apply_filter_tiled(img, c["scans"], c["clamped"], tile=c["tile"], threads=2)
    assert rc.rel_err(got, want) < 2e-6


@pytest.mark.parametrize("coeff,clamped,tile", [(rc.GAUSS2, True, 32), (rc.GAUSS3, True, 16), (rc.BICUBIC_COEFF, True, 8),
                                                (rc.GAUSS2, False, 32)])
def test_tiled_cpu_counterpart_clamped_and_mixed_tiles(coeff, clamped, tile):
    img = rc.random_image((96, 160), np.float32, 7)
    want = oracle.apply_filter(img.astype(np.float64), rc.xy_pm(coeff), clamped)
    got = oracle.apply_filter_tiled(img, rc.xy_pm(coeff), clamped, tile=[tile, 0] if tile == 8 else tile, threads=3)
    assert rc.rel_err(got, want) < 5e-5
    # in place on the caller's array
    work = img.copy()
    oracle.apply_filter_tiled(work, rc.xy_pm(coeff), clamped, tile=tile, inplace=True)
    assert rc.rel_err(work, want) < 5e-5


@pytest.mark.parametrize("order", [9, 12, 17, 24, 29, 32])
def test_high_order_scans_against_scipy_lfilter(order):
    """The orders round 5 added (RF_MAX_ORDER 32; the reference's sweep: apps/audio/audio_filter_high_order.cpp:38-42) against
    scipy.signal.lfilter / lfiltic: zero border, causal and anticausal, along x and y; and the clamped border through lfilter
    with the initial state "every previous output equals y[0]" (as test_clamped_scan_against_lfilter_with_initial_state)."""
    from scipy.signal import lfilter, lfiltic
    rng = np.random.default_rng(500 + order)
    a = rng.standard_normal(order) * np.exp(-0.15 * np.arange(order))
    a *= 0.85 / np.abs(a).sum()
    for coeff in ([1.0] + [0.01] * order, [0.4] + [float(np.float32(v)) for v in a]):
        coeff = [float(np.float32(v)) for v in coeff]
        img = rng.standard_normal((7, 300))
        for dim in (0, 1):
            for causal in (True, False):
                want = _lfilter_scan(img, causal, coeff) if dim == 0 else _lfilter_scan(img.T, causal, coeff).T
                got = oracle.apply_filter(img, [(dim, causal, coeff)], clamped=False)
                assert np.max(np.abs(got - want)) <= 1e-10 * max(1.0, float(np.max(np.abs(want))))
        x = rng.standard_normal(400)
        b, den = [coeff[0]], [1.0] + [-c for c in coeff[1:]]
        y0 = (coeff[0] + sum(coeff[1:])) * x[0]
        rest, _ = lfilter(b, den, x[1:], zi=lfiltic(b, den, y=[y0] * order))
        want = np.concatenate([[y0], rest])
        got = oracle.apply_filter(x, [(0, True, coeff)], clamped=True)
        assert np.max(np.abs(got - want)) < 1e-10
        got_rev = oracle.apply_filter(x[::-1].copy(), [(0, False, coeff)], clamped=True)
        assert np.max(np.abs(got_rev[::-1] - want)) < 1e-10
