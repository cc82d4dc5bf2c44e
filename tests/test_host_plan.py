"""CPU tests of the product's host side: the C-ABI library loads, exports every symbol the header
declares, its coefficient design matches the reference's values, its argument checks mirror the
reference's, and its plan tables (the tiling algebra) reproduce the untiled oracle when replayed by
the numpy emulator.  No kernel is launched here (no GPU in this container)."""
import ctypes
import re
import os

import numpy as np
import pytest

import oracle
import ref_cases as rc
import tiled_emulator as emu
import recfilter_amd as rfa
from recfilter_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "recfilter_amd.h")).read()
    declared = set(re.findall(r"\b(rf_[a-z_0-9]+)\s*\(", header))
    assert declared == set(capi.EXPORTED_SYMBOLS)
    L = capi.lib()
    for name in sorted(declared):
        assert hasattr(L, name), name
    assert b"gfx950" in L.rf_version()


def test_no_device_means_loud_failure_not_fallback():
    if capi.lib().rf_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(rfa.RecFilterError) as e:
        rfa.Plan((16, 16), [(0, True, [1.0, 1.0])])
    assert e.value.status == capi.RF_ERR_HIP


@pytest.mark.parametrize("order", [1, 2, 3])
def test_gaussian_weights_match_reference_values(order):
    got = rfa.gaussian_weights(5.0, order)
    np.testing.assert_allclose(got, rc.GAUSS_SIGMA5[order], rtol=3e-7)
    np.testing.assert_array_equal(np.float32(got), oracle.gaussian_weights(5.0, order))


@pytest.mark.parametrize("sigma", [0.8, 1.0, 2.5, 16.0])
def test_gaussian_weights_match_oracle_other_sigmas(sigma):
    for order in (1, 2, 3):
        np.testing.assert_allclose(rfa.gaussian_weights(sigma, order), oracle.gaussian_weights(sigma, order), rtol=1e-6)


def test_other_coefficient_helpers():
    for n in (1, 2, 3):
        assert rfa.integral_image_coeff(n) == rc.INTEGRAL_COEFF[n]
    assert rfa.overlap_feedback_coeff([2, -1], [1]) == [3, -3, 1]
    assert rfa.gaussian_box_filter(3, 1.0) == 2 and rfa.gaussian_box_filter(3, 2.0) == 4
    for k in (1, 2, 3, 4, 5):
        assert rfa.gaussian_box_filter(k, 3.0) == oracle.gaussian_box_filter(k, 3.0)


def _host_plan(shape, scans, **kw):
    return rfa.Plan(shape, scans, device=capi.RF_DEVICE_HOST_ONLY, **kw)


def test_argument_checks_mirror_the_reference():
    with pytest.raises(ValueError):                      # lib/recfilter.cpp:274-278
        _host_plan((8, 8), [(0, True, [1.0])])
    with pytest.raises(rfa.RecFilterError) as e:         # lib/recfilter.cpp:296-300
        _host_plan((8, 8), [(2, True, [1.0, 1.0])])
    assert e.value.status == capi.RF_ERR_INVALID_ARG
    with pytest.raises(rfa.RecFilterError):              # lib/recfilter.h:311
        _host_plan((8, 8), [(0, True, [1.0, 1.0])], tile=[3, 0], path=capi.RF_PATH_TILED_GENERIC)
    with pytest.raises(rfa.RecFilterError) as e:         # host-only plans never execute
        p = _host_plan((8, 8), [(0, True, [1.0, 1.0])], path=capi.RF_PATH_TILED_GENERIC)
        capi.check(capi.lib().rf_plan_execute(p._h, (ctypes.c_void_p * 1)(1), (ctypes.c_void_p * 1)(1), None))
    assert e.value.status == capi.RF_ERR_HIP


def test_matrix_A_is_tail_of_reference_matrix_R():
    # A[s][r][j] = R[T-1-r][j] with R = lib/coefficients.cpp:51-83 (via the oracle's restatement)
    fb = [0.5, 0.25]
    p = _host_plan((1, 32), [(0, True, [1.0] + fb)], dtype=np.float64, tile=[8, 0], path=capi.RF_PATH_TILED_GENERIC)
    A = p.table("A_x").reshape(2, 2)
    R = oracle.matrix_R(fb, 8)
    np.testing.assert_allclose(A, R[[7, 6], :], rtol=1e-6)
    prop = p.table("prop_x").reshape(4, 1, 1, 8, 2)
    np.testing.assert_allclose(prop[0, 0, 0], R, rtol=1e-6)


@pytest.mark.parametrize("name", sorted(rc.REFERENCE_TESTS))
def test_plan_tables_reproduce_oracle_on_reference_tests(name):
    case = rc.REFERENCE_TESTS[name]
    if np.issubdtype(np.dtype(case["dtype"]), np.integer):
        pytest.skip("integer tables are checked in test_integer_tables_are_exact")
    shape = case["shape"]
    tile = [case["tile"] if any(s[0] == d for s in case["scans"]) else 0 for d in range(len(shape))]
    p = _host_plan(shape, case["scans"], dtype=np.float64, clamped=case["clamped"], tile=tile,
                   path=capi.RF_PATH_TILED_GENERIC)
    assert p.path == capi.RF_PATH_TILED_GENERIC
    img = rc.random_image(shape).astype(np.float64)
    want = oracle.apply_filter(img, case["scans"], case["clamped"])
    got = emu.emulate_filter(img, case["scans"], p.tiles, case["clamped"], p)
    assert rc.rel_err(got, want) < 1e-11


@pytest.mark.parametrize("tile", [4, 8, 16])
@pytest.mark.parametrize("coeff", [rc.GAUSS2, rc.GAUSS3, rc.BICUBIC_COEFF])
def test_plan_tables_clamped_mixed_causality(tile, coeff):
    # the case where the reference's matrix_B(clamp) is inconsistent with its own add_filter
    # (SURVEY 8 a-4); the product's tables follow add_filter, so tiled == untiled to rounding
    shape = (32, 48)
    scans = rc.xy_pm([float(np.float32(c)) for c in coeff])
    p = _host_plan(shape, scans, dtype=np.float64, clamped=True, tile=[tile, tile], path=capi.RF_PATH_TILED_GENERIC)
    img = rc.random_image(shape).astype(np.float64)
    want = oracle.apply_filter(img, scans, True)
    got = emu.emulate_filter(img, scans, p.tiles, True, p)
    assert rc.rel_err(got, want) < 1e-11


def test_single_tile_per_dimension():
    shape = (8, 8)
    scans = rc.xy_pm([float(np.float32(c)) for c in rc.GAUSS2])
    p = _host_plan(shape, scans, dtype=np.float64, clamped=True, tile=[8, 8], path=capi.RF_PATH_TILED_GENERIC)
    img = rc.random_image(shape).astype(np.float64)
    got = emu.emulate_filter(img, scans, p.tiles, True, p)
    assert rc.rel_err(got, oracle.apply_filter(img, scans, True)) < 1e-12


def test_integer_tables_are_exact():
    # integer pixel types: tables are built in wrap-around integer arithmetic
    case = rc.REFERENCE_TESTS["test_type_invariance"]
    p = _host_plan(case["shape"], case["scans"], dtype=np.int32, tile=[4, 4], path=capi.RF_PATH_TILED_GENERIC)
    A = p.table("A_x")
    assert np.all(A == np.round(A))
    img = rc.random_image(case["shape"], np.int32)
    want = oracle.apply_filter(img, case["scans"])
    got = emu.emulate_filter(img.astype(np.float64), case["scans"], p.tiles, False, p)
    np.testing.assert_array_equal(got.astype(np.int64), want.astype(np.int64))


def test_auto_tiles_divide_extents():
    p = _host_plan((96, 200), [(0, True, [1.0, 0.5]), (1, False, [1.0, 0.5, 0.25])], dtype=np.float32,
                   path=capi.RF_PATH_TILED_GENERIC)
    tx, ty = p.tiles
    assert 200 % tx == 0 and 96 % ty == 0 and tx >= 1 and ty >= 2


def test_front_end_misuse_raises_like_the_reference_asserts():
    x, y = rfa.RecFilterDim("x", 16), rfa.RecFilterDim("y", 16)
    f = rfa.RecFilter("F")
    with pytest.raises(rfa.RecFilterUsageError):     # add_filter before define, recfilter.cpp:268-272
        f.add_filter(+x, [1.0, 0.5])

    class FakeTensor:
        shape = (16, 16)
        dtype = np.float32
    f.define([x, y], [FakeTensor()])
    with pytest.raises(rfa.RecFilterUsageError):     # redefinition, recfilter.cpp:205-208
        f.define([x, y], [FakeTensor()])
    with pytest.raises(rfa.RecFilterUsageError):     # clamped after define, recfilter.cpp:252-256
        f.set_clamped_image_border()
    with pytest.raises(rfa.RecFilterUsageError):     # <2 coefficients, recfilter.cpp:274-278
        f.add_filter(+x, [1.0])
    with pytest.raises(rfa.RecFilterUsageError):     # unknown dimension, recfilter.cpp:296-300
        f.add_filter(+rfa.RecFilterDim("w", 4), [1.0, 0.5])
    f.add_filter(+x, [1.0, 0.5])
    f.add_filter(-x, [1.0, 0.5])
    with pytest.raises(rfa.RecFilterUsageError):     # tiling a dimension without scans, split.cpp:1879-1883
        f.split(y, 4)
    with pytest.raises(rfa.RecFilterUsageError):     # opposite causality reordered, reorder.cpp:70-75
        f.cascade([1], [0])
    with pytest.raises(rfa.RecFilterUsageError):     # scan missing from the cascade lists
        f.cascade([0], [])
    parts = f.cascade([0], [1])
    assert len(parts) == 2 and parts[1]._contents["source"] is parts[0]
    f.split(x, 4)
    with pytest.raises(rfa.RecFilterUsageError):     # tiling twice, split.cpp:1851-1854
        f.split(x, 4)
    with pytest.raises(rfa.RecFilterUsageError):     # cascade after tiling, reorder.cpp:29-33
        f.cascade([0], [1])
    with pytest.raises(rfa.RecFilterUsageError):     # full_schedule on a tiled filter, recfilter.cpp:397-401
        f.full_schedule()
    f.intra_schedule(1).compute_locally().unroll("x").gpu_threads("a", "b")   # accepted, recorded
    assert len(f._contents["schedule_log"]) == 3


def test_overlap_to_higher_order_filter_matches_cascade_in_oracle():
    # tests/test_overlap_filter_order.cpp through the front-end's coefficient algebra
    x, y = rfa.RecFilterDim("x", 12), rfa.RecFilterDim("y", 12)

    class FakeTensor:
        shape = (12, 12)
        dtype = np.float32
    f1 = rfa.RecFilter("R1"); f1.define([x, y], [FakeTensor()])
    f1.add_filter(+x, [1.0, 2.0, -1.0]); f1.add_filter(+y, [1.0, 1.0])
    f2 = rfa.RecFilter("R2"); f2.define([x, y], f1)
    f2.add_filter(+x, [1.0, 1.0]); f2.add_filter(+y, [1.0, 2.0, -1.0])
    f3 = f2.overlap_to_higher_order_filter(f1, "O")
    img = rc.random_image((12, 12))
    cascaded = oracle.apply_filter(oracle.apply_filter(img, f1._contents["scans"]), f2._contents["scans"])
    overlapped = oracle.apply_filter(img, f3._contents["scans"])
    assert rc.rel_err(overlapped, cascaded) < 1e-5


# ---- fused x/y path: the whole tiling algebra (segment tables, chaining, cross-dimension residual) ----
FUSED_CASES = rc.FUSED_CASES


@pytest.mark.parametrize("name", sorted(FUSED_CASES))
def test_fused_plan_tables_reproduce_oracle(name):
    import fused_emulator
    case = FUSED_CASES[name]
    p = _host_plan(case["shape"], case["scans"], dtype=np.float32, clamped=case["clamped"],
                   path=capi.RF_PATH_TILED_FUSED)
    assert p.path == capi.RF_PATH_TILED_FUSED and p.tiles[0] == 256 and p.tiles[1] in (32, 64)
    img = rc.random_image(case["shape"]).astype(np.float64)
    want = oracle.apply_filter(img, case["scans"], case["clamped"])
    got = fused_emulator.FusedEmu(p, case["scans"], case["clamped"]).run(img)
    assert rc.rel_err(got, want) < 2e-6      # tables are stored in f32, the emulation runs in f64


@pytest.mark.parametrize("order", [4, 5, 6, 8])
def test_clamped_sections_tables_reproduce_oracle(order):
    """Orders 4..8 with a clamped border: the plan rewrites every scan into zero-border form behind a border
    modification (rf_plan_table("scans"): sections + the gammas of the scan as given).  The numpy replay of the fused
    pipeline from the plan's tables -- pass-1 contractions, carries, residual, final pass with the modification applied
    where the kernels apply it -- must reproduce the oracle run on the ORIGINAL high-order coefficients with its clamped
    border.  Checks the host side of the rewrite (gammas, tables built through them) without a GPU."""
    import fused_emulator
    poles = {4: [0.7, 0.6, 0.3 + 0.5j, 0.3 - 0.5j], 5: [0.8, 0.5 + 0.3j, 0.5 - 0.3j, -0.2 + 0.6j, -0.2 - 0.6j],
             6: [0.85, 0.1, 0.4 + 0.4j, 0.4 - 0.4j, -0.5 + 0.2j, -0.5 - 0.2j],
             8: [0.8, -0.7, 0.5, -0.3, 0.3 + 0.6j, 0.3 - 0.6j, -0.1 + 0.7j, -0.1 - 0.7j]}[order]
    co = [0.25] + [float(-v) for v in np.poly(poles).real[1:]]
    scans = [(0, True, co), (0, False, co), (1, True, co), (1, False, co)] if order <= 6 else [(0, False, co), (1, True, co)]
    shape = (96, 528)                                  # three 32-row tile rows; a partial last tile column of 16 columns
    p = _host_plan(shape, scans, dtype=np.float32, clamped=True, path=capi.RF_PATH_TILED_FUSED)
    assert p.path == capi.RF_PATH_TILED_FUSED
    run_scans, mods = fused_emulator.plan_scans(p)
    assert len(run_scans) > len(scans) and all(len(c) - 1 <= 3 for _, _, c in run_scans)      # sections of order <= 3
    assert all(m[0] >= 0 for m in mods)                                                        # every scan in mod form
    assert sorted(m[0] for m in mods if m[0] > 0) == [order] * len(scans)                      # one modification per scan as given
    img = rc.random_image(shape).astype(np.float64)
    want = oracle.apply_filter(img, scans, True)
    got = fused_emulator.FusedEmu(p, run_scans, True, mods).run(img)
    assert rc.rel_err(got, want) < 2e-5      # sections rounded to f32 recompose the denominator to 2e-6


def test_fused_not_applicable_falls_back_in_auto_mode():
    # a width that is not a multiple of 4: fused for 4- and 8-byte pixels (rows end in a partial chunk); int16 pixels are
    # moved in 8-byte pieces and keep the rule
    p = _host_plan((100, 302), rc.xy_pm(rc.GAUSS2), clamped=True)
    assert p.path == capi.RF_PATH_TILED_FUSED
    ints = [(0, True, [1.0, 1.0]), (1, True, [1.0, 1.0])]
    p = _host_plan((100, 302), ints, dtype=np.int16)
    assert p.path == capi.RF_PATH_TILED_GENERIC
    with pytest.raises(rfa.RecFilterError) as e:
        _host_plan((100, 302), ints, dtype=np.int16, path=capi.RF_PATH_TILED_FUSED)
    assert e.value.status == capi.RF_ERR_UNSUPPORTED
    p = _host_plan((64, 512), rc.xy_pm(rc.GAUSS2), clamped=True)
    assert p.path == capi.RF_PATH_TILED_FUSED


@pytest.mark.parametrize("name", ["gauss2_clamped", "generic_xy_zero", "bicubic_clamped_ty32"])
def test_tail_impulse_responses_match_tile_local_scans(name):
    """H_x / H_y (pass 1 as a contraction, kernels_tails.hip) reproduce the tails the tile-local scans give."""
    from tiled_emulator import scan_tile
    case = rc.FUSED_CASES[name]
    p = _host_plan(case["shape"], case["scans"], dtype=np.float32, clamped=case["clamped"], path=capi.RF_PATH_TILED_FUSED)
    rng = np.random.default_rng(3)
    for d, T, tname in ((0, 256, "H_x"), (1, p.tiles[1], "H_y")):
        scans = [(bool(c), [float(np.float32(v)) for v in co]) for dd, c, co in case["scans"] if dd == d]
        n = len(scans)
        K = max(len(co) - 1 for dd, c, co in case["scans"] if dd in (0, 1))
        H = p.table(tname).reshape(4, n, K, T)
        rows = rng.random((5, T))
        for v in range(4):
            work = rows.copy()
            for s, (causal, co) in enumerate(scans):
                a = list(co[1:]) + [0.0] * (K - len(co) + 1)
                clamp = case["clamped"] and ((v & 1) != 0 if causal else (v & 2) != 0)
                scan_tile(work, causal, co[0], a, K, clamp)
                for r in range(K):
                    pos = T - 1 - r
                    want = work[:, pos if causal else T - 1 - pos]
                    got = rows @ H[v, s, r]
                    np.testing.assert_allclose(got, want, rtol=2e-6, atol=1e-7 * np.abs(want).max())


def test_pointwise_desc_validation_and_kernel_count():
    """rf_pointwise_desc: float pixel types only; fused into the passes on the fused path (no extra kernels),
    stand-alone elementwise steps on the others."""
    scans = rc.BASELINE_CONFIGS["cfg3_gaussian2_xy"]["scans"]
    with pytest.raises(rfa.RecFilterError) as e:
        _host_plan((64, 256), [(0, True, [1.0, 1.0])], dtype=np.int32, epilogue=(1.0, 1.0, 0.0))
    assert e.value.status == capi.RF_ERR_UNSUPPORTED
    plain = _host_plan((64, 256), scans, clamped=True)
    fused = _host_plan((64, 256), scans, clamped=True, prologue=(2.0, 1.0), epilogue=(1.0, -1.0, 0.0))
    assert fused.path_name == "tiled_fused" and fused.num_kernels == plain.num_kernels
    gen = _host_plan((64, 256), scans, clamped=True, path=capi.RF_PATH_TILED_GENERIC)
    gen_pw = _host_plan((64, 256), scans, clamped=True, path=capi.RF_PATH_TILED_GENERIC,
                        prologue=(2.0, 1.0), epilogue=(1.0, -1.0, 0.0))
    assert gen_pw.num_kernels == gen.num_kernels + 2
    # layout of the descriptor the binding fills must match the header's struct
    assert ctypes.sizeof(capi.PointwiseDesc) == 28


def test_unsigned_byte_input_planes_are_a_prologue():
    """rf_pointwise_desc.in_dtype = RF_IN_U8: f32 pixels only; fused into the passes on the fused path, one conversion
    kernel in front of the others."""
    scans = rc.BASELINE_CONFIGS["cfg3_gaussian2_xy"]["scans"]
    with pytest.raises(rfa.RecFilterError) as e:
        _host_plan((64, 256), scans, dtype=np.float64, input_dtype=np.uint8)
    assert e.value.status == capi.RF_ERR_UNSUPPORTED
    plain = _host_plan((64, 256), scans, clamped=True)
    u8 = _host_plan((64, 256), scans, clamped=True, input_dtype=np.uint8, prologue=(1 / 255.0, 0.0))
    assert u8.path_name == "tiled_fused" and u8.num_kernels == plain.num_kernels
    # (a width that is not a multiple of 4: uint8 inputs are moved in 4-byte pieces and leave the fused kernels)
    gen = _host_plan((64, 250), scans, clamped=True, path=capi.RF_PATH_TILED_GENERIC)
    gen_u8 = _host_plan((64, 250), scans, clamped=True, input_dtype=np.uint8)
    assert gen_u8.path_name == "tiled_generic" and gen_u8.num_kernels == gen.num_kernels + 1


def test_second_order_sections_reproduce_the_scan():
    """recfilter_amd.second_order_sections: the sections' cascade (zero border) is the original scan."""
    rng = np.random.default_rng(3)
    x = rng.random(4000)
    for coeff in ([1.0] + [0.01] * 9, rfa.gaussian_weights(5.0, 3), [0.7, 0.5, -0.3, 0.1, 0.05], [1.0, 0.5]):
        secs = rfa.second_order_sections(coeff)
        assert all(len(s) <= 3 for s in secs)
        y = x.copy()
        for s in secs:
            y = oracle.apply_filter(y, [(0, True, s)], False)
        want = oracle.apply_filter(x, [(0, True, list(coeff))], False)
        assert rc.rel_err(y, want) < 1e-5


# ---- the fully overlapped N-D tiling (RF_PATH_TILED_OVERLAPPED): residuals between every pair of dimensions --------
import overlap_emulator as ovemu


@pytest.mark.parametrize("name", ["test_trivial", "test_causal_xy", "test_causal_anticausal_xy", "test_generic_xy", "test_generic_xyz"])
def test_overlapped_tables_reproduce_oracle_on_reference_tests(name):
    case = rc.REFERENCE_TESTS[name]
    shape = case["shape"]
    tile = [case["tile"] if any(s[0] == d for s in case["scans"]) else 0 for d in range(len(shape))]
    auto = _host_plan(shape, case["scans"], dtype=np.float64, clamped=case["clamped"], tile=tile)
    # split() along >= 2 dimensions: the fused kernels where they apply (these filters, any float / integer pixel type),
    # the fully overlapped tiling otherwise -- e.g. a clamped filter of order 4
    assert auto.path == capi.RF_PATH_TILED_FUSED
    o4 = [(0, True, [0.5, 0.2, 0.1, 0.05, 0.02]), (1, False, [0.5, 0.2, 0.1, 0.05, 0.02])]
    assert _host_plan((32, 32), o4, dtype=np.float64, clamped=True, tile=[8, 8]).path == capi.RF_PATH_TILED_OVERLAPPED
    p = _host_plan(shape, case["scans"], dtype=np.float64, clamped=case["clamped"], tile=tile, path=capi.RF_PATH_TILED_OVERLAPPED)
    assert p.path == capi.RF_PATH_TILED_OVERLAPPED and list(p.tiles) == tile
    img = rc.random_image(shape).astype(np.float64)
    want = oracle.apply_filter(img, case["scans"], case["clamped"])
    got = ovemu.emulate_overlapped(img, case["scans"], tile, case["clamped"], p)
    assert rc.rel_err(got, want) < 1e-11


@pytest.mark.parametrize("clamped", [False, True])
def test_overlapped_3d_mixed_tiles_orders_and_borders(clamped):
    # different tile widths and orders per dimension, causal + anticausal everywhere, a dimension without scans
    scans = [(0, True, [0.6, 0.3, -0.1]), (0, False, [0.7, 0.2]), (1, False, [0.5, 0.4, -0.1, 0.05]), (1, True, [0.8, 0.1]),
             (2, True, [0.9, 0.05]), (2, False, [0.5, 0.3, 0.1])]
    shape = (12, 10, 16)
    tile = [8, 5, 4]
    p = _host_plan(shape, scans, dtype=np.float64, clamped=clamped, tile=tile, path=capi.RF_PATH_TILED_OVERLAPPED)
    img = rc.random_image(shape, seed=3).astype(np.float64)
    got = ovemu.emulate_overlapped(img, scans, tile, clamped, p)
    assert rc.rel_err(got, oracle.apply_filter(img, scans, clamped)) < 1e-11
    yz = [s for s in scans if s[0] != 0]                    # x unfiltered: y -> z is the only residual
    p2 = _host_plan(shape, yz, dtype=np.float64, clamped=clamped, tile=[0, 5, 4], path=capi.RF_PATH_TILED_OVERLAPPED)
    got2 = ovemu.emulate_overlapped(img, yz, [0, 5, 4], clamped, p2)
    assert rc.rel_err(got2, oracle.apply_filter(img, yz, clamped)) < 1e-11


def test_overlapped_path_preconditions():
    scans = rc.REFERENCE_TESTS["test_generic_xyz"]["scans"]
    with pytest.raises(rfa.RecFilterError):                  # needs an explicit tile for every filtered dimension
        _host_plan((16, 16, 16), scans, tile=[4, 4, 0], path=capi.RF_PATH_TILED_OVERLAPPED)
    with pytest.raises(rfa.RecFilterError):                  # tile volume above 4096 samples
        _host_plan((64, 64, 64), scans, tile=[32, 16, 16], path=capi.RF_PATH_TILED_OVERLAPPED)
    # without explicit tiles the automatic choice is unchanged
    assert _host_plan((16, 16, 16), scans).path != capi.RF_PATH_TILED_OVERLAPPED


def test_high_order_scans_are_split_into_sections_for_the_fused_path():
    """Plan rewrite of sections.h, decided on the host: zero border + float pixels + at most four scans per dimension after
    the split -> fused; otherwise the scans run as given on another path."""
    import recfilter_amd as rfa
    from recfilter_amd import capi
    T = capi.RF_PLAN_TILED_ONLY

    def from_poles(poles, b=0.3):
        p = np.poly(poles).real
        return [b] + [float(-v) for v in p[1:]]
    o5 = from_poles([0.8, 0.5 + 0.3j, 0.5 - 0.3j, -0.2 + 0.6j, -0.2 - 0.6j])
    o6c = from_poles([0.6 + 0.2j, 0.6 - 0.2j, 0.4 + 0.4j, 0.4 - 0.4j, -0.5 + 0.2j, -0.5 - 0.2j])
    xy = lambda co: [(0, True, co), (0, False, co), (1, True, co), (1, False, co)]

    def path_of(scans, **kw):
        kw.setdefault("flags", T)
        with rfa.Plan((512, 512), scans, device=capi.RF_DEVICE_HOST_ONLY, **kw) as plan:
            return plan.path_name
    assert path_of(xy(o5), clamped=False) == "tiled_fused"                 # 5 = 3 + 2: four scans per dimension
    # clamped border (2-D / 3-D f32, width % 16 == 0, height % 32 == 0): the same sections in zero-border form behind border
    # modifications (plan.cpp, "clamped sections"); other shapes and f64 keep the scans as given on another path
    assert path_of(xy(o5), clamped=True) == "tiled_fused"
    assert path_of(xy(o5), clamped=True, dtype=np.float64) != "tiled_fused"
    with rfa.Plan((512, 520), xy(o5), clamped=True, device=capi.RF_DEVICE_HOST_ONLY, flags=T) as plan:
        assert plan.path_name != "tiled_fused"                              # width not a multiple of 16
    with rfa.Plan((520, 512), xy(o5), clamped=True, device=capi.RF_DEVICE_HOST_ONLY, flags=T) as plan:
        assert plan.path_name != "tiled_fused"                              # height not a multiple of 32
    with rfa.Plan((100_000,), [(0, True, o5)], clamped=True, device=capi.RF_DEVICE_HOST_ONLY, flags=T) as plan:
        assert plan.path_name != "tiled_fused" or plan.num_kernels > 0       # 1-D: the clamped 1-D path (plan_clamp1d.h), not this rewrite
    assert path_of(xy(o5), clamped=False, dtype=np.float64) == "tiled_fused"     # f64 pixels: sections in f64
    assert path_of(xy([1.0, 1.0, 0.0, 0.0, 1.0]), clamped=False, dtype=np.int32) != "tiled_fused"     # integer pixels: as given
    assert path_of(xy(o6c), clamped=False) == "tiled_fused"                # three conjugate pairs, twice: six sections per
                                                                           # dimension -> two stages of an in-plan cascade
    assert path_of(xy(o6c), clamped=False, flags=T | capi.RF_PLAN_NO_CASCADE) != "tiled_fused"     # (without the cascade: as given, another path)
    assert path_of([(0, True, o6c), (1, True, o6c)], clamped=False) == "tiled_fused"
    assert path_of(xy(o5), clamped=False, flags=T | capi.RF_PLAN_NO_SECTIONS) != "tiled_fused"


def test_in_plan_cascade_decisions():
    """More than four scans in a dimension, or a zero-padded 1-D signal whose anticausal scans follow causal ones, run as
    successive fused stages inside one plan (plan.cpp, build_cascade) instead of on the generic path; decided on the host."""
    import recfilter_amd as rfa
    from recfilter_amd import capi
    T = capi.RF_PLAN_TILED_ONLY

    def plan_of(shape, scans, **kw):
        kw.setdefault("flags", T)
        return rfa.Plan(shape, scans, device=capi.RF_DEVICE_HOST_ONLY, **kw)
    bq = [0.05, 1.6, -0.7]
    with plan_of((1_000_000,), [(0, True, bq)] * 4) as four, plan_of((1_000_000,), [(0, True, bq)] * 5) as five, \
         plan_of((1_000_000,), [(0, True, bq)] * 9) as nine:
        assert four.path_name == five.path_name == nine.path_name == "tiled_fused"
        assert five.num_kernels > four.num_kernels and nine.num_kernels > five.num_kernels
    # 1-D, length not a multiple of 8192: causal then anticausal is two stages (each copies only the signal out of its padding)
    with plan_of((100_000,), [(0, True, bq), (0, False, bq)]) as mixed, plan_of((100_000,), [(0, True, bq)]) as one, \
         plan_of((8192 * 12,), [(0, True, bq), (0, False, bq)]) as whole:       # (no padding: one stage)
        assert mixed.path_name == "tiled_fused" and mixed.num_kernels == 2 * one.num_kernels
        assert whole.path_name == "tiled_fused" and whole.num_kernels < mixed.num_kernels
    # 2-D: six x scans and one y scan
    with plan_of((256, 512), [(0, True, [0.5, 0.5])] * 6 + [(1, False, [0.6, 0.4])], clamped=True) as p:
        assert p.path_name == "tiled_fused"
    # an epilogue that reads the input cannot be cascaded (the last stage no longer sees it): as before
    with plan_of((256, 512), [(0, True, [0.5, 0.5])] * 5, epilogue=(1.0, 1.0, 0.0)) as p:
        assert p.path_name != "tiled_fused"
    # clamped 1-D signals: the zero-border fused plan (here a cascade of two stages) plus the border corrections
    # (plan_clamp1d.h) -- unless the filter does not decay (a running sum), which keeps the generic path
    with plan_of((100_000,), [(0, True, bq)] * 5, clamped=True) as p, plan_of((100_000,), [(0, True, bq)] * 5) as z:
        assert p.path_name == "tiled_fused" and p.num_kernels == z.num_kernels + 2
        assert p.table("clamp1d_L")[0] % 256 == 0 and p.table("clamp1d_H").size == 25
    with plan_of((100_000,), [(0, True, [1.0, 1.0])], clamped=True) as p:
        assert p.path_name != "tiled_fused"
    with plan_of((1_000_000,), [(0, True, bq)] * 5, flags=T | capi.RF_PLAN_NO_CASCADE) as p:
        assert p.path_name != "tiled_fused"


@pytest.mark.parametrize("seed", range(8))
def test_clamped_1d_border_correction_tables(seed):
    """plan_clamp1d.h on the host: a clamped scan is the zero-border scan plus (the border sample) x (a fixed decaying
    sequence), so a clamped 1-D filter is its zero-border form plus rank-one corrections of the two ends whose weights
    (w: what the border samples were before each scan, as dot products with the input's ends; H: how earlier corrections
    move later border samples; G: what each correction adds to the output) depend on the filter alone.  Replayed here in
    numpy from the plan's tables against the oracle's clamped filter."""
    rng = np.random.default_rng(4400 + seed)
    n = int(rng.integers(1, 6))
    scans = []
    for _ in range(n):
        k = int(rng.integers(1, 4))
        co = [float(rng.uniform(0.3, 1.2))] + [float(-c) for c in np.poly(rng.uniform(-0.8, 0.8, size=k))[1:]]
        scans.append((0, bool(rng.integers(0, 2)), [float(np.float32(v)) for v in co]))
    N = 40_000
    p = _host_plan((N,), scans, clamped=True)
    assert p.path == capi.RF_PATH_TILED_FUSED
    L = int(p.table("clamp1d_L")[0])
    w, G, H = p.table("clamp1d_w").reshape(n, L), p.table("clamp1d_G").reshape(n, L), p.table("clamp1d_H").reshape(n, n)
    x = rng.standard_normal(N)
    out = oracle.apply_filter(x, scans, clamped=False)
    beta = np.zeros(n)
    for s in range(n):
        win = x[:L] if scans[s][1] else x[N - L:]
        beta[s] = w[s] @ win + sum(beta[q] * H[q, s] for q in range(s))
    for s in range(n):
        if scans[s][1]:
            out[:L] += beta[s] * G[s]
        else:
            out[N - L:] += beta[s] * G[s]
    want = oracle.apply_filter(x, scans, clamped=True)
    assert np.max(np.abs(out - want)) <= 1e-11 * np.max(np.abs(want))


def test_one_read_pass1_z_tail_responses_reproduce_the_tile_local_scans():
    """kernels_tails_walk.hip contracts the planes of a z tile with the table H_z: for a one-tile extent it must give the
    tails of the dimension's scans run on the column (the oracle's loops), clamped border included; a plan that keeps the
    two first passes (RF_PLAN_STAGED_PASS1) or cannot take the kernel (integer pixels) has no such table."""
    scans = rc.REFERENCE_TESTS["test_generic_xyz"]["scans"]
    zs = [(0, c, co) for d, c, co in scans if d == 2]
    rng = np.random.default_rng(3)
    for clamped in (False, True):
        p = _host_plan((64, 64, 256), scans, clamped=clamped, path=capi.RF_PATH_TILED_FUSED, flags=capi.RF_PLAN_WALK_PASS1)
        H = p.table("H_z").reshape(4, len(zs) * 2, 64)
        col = rng.random(64)
        y1 = oracle.apply_filter(col, zs[:1], clamped)
        y2 = oracle.apply_filter(col, zs, clamped)
        want = [y1[63], y1[62], y2[0], y2[1]]           # causal: the last samples; anticausal: the first
        np.testing.assert_allclose(H[3] @ col, want, rtol=1e-12, atol=1e-12)
        if not clamped:                                 # zero border: every variant is the interior one
            np.testing.assert_allclose(H[0] @ col, want, rtol=1e-12, atol=1e-12)
    with pytest.raises(Exception):
        _host_plan((64, 64, 256), scans, path=capi.RF_PATH_TILED_FUSED, flags=capi.RF_PLAN_STAGED_PASS1).table("H_z")
    with pytest.raises(Exception):
        _host_plan((64, 64, 256), [(0, True, [1, 1]), (1, True, [1, 1]), (2, True, [1, 1])], dtype=np.int32,
                   path=capi.RF_PATH_TILED_FUSED, flags=capi.RF_PLAN_WALK_PASS1).table("H_z")
    with pytest.raises(Exception):                      # 32 patch columns: not the default choice
        _host_plan((64, 64, 256), scans, path=capi.RF_PATH_TILED_FUSED).table("H_z")
    assert _host_plan((512, 512, 512), scans).table("H_z").size == 4 * 4 * 64
    # z slabs take the one-read pass 1 with the early exchange (it is then the step the exchange waits for), not with the late one
    slab = dict(shard_rank=3, shard_world=8)
    assert _host_plan((256, 1024, 1024), scans, **slab).table("H_z").size == 4 * 4 * 128
    with pytest.raises(Exception):
        _host_plan((256, 1024, 1024), scans, flags=capi.RF_PLAN_LATE_EXCHANGE, **slab).table("H_z")


# ---- the matrix path (orders up to 32 in their direct form): its tables replayed in numpy against the untiled oracle ----
def _stable(order, seed, b=0.4, mass=0.85):
    rng = np.random.default_rng(seed)
    a = rng.standard_normal(order) * np.exp(-0.15 * np.arange(order))
    a *= mass / np.abs(a).sum()
    return [b] + [float(np.float32(v)) for v in a]


@pytest.mark.parametrize("clamped", [False, True])
@pytest.mark.parametrize("case", ["audio_1d", "xy_pm_12", "xyz_mixed", "order32_1d_levels", "ragged_xy", "ragged_1d_short", "ragged_xyz",
                                  "pair_1d", "pair_xy_tiles", "pair_then_single", "wide_one_chain"])
def test_matrix_path_tables_reproduce_the_oracle(case, clamped):
    import matrix_emulator as mxe
    if case == "audio_1d":            # apps/audio/audio_filter_high_order.cpp:41-42 at its highest order
        shape, scans = (4096,), [(0, True, [1.0] + [0.01] * 29)]
    elif case == "xy_pm_12":
        c = _stable(12, 3)
        shape, scans = (96, 160), [(0, True, c), (0, False, c), (1, True, c), (1, False, c)]
    elif case == "xyz_mixed":
        shape, scans = (64, 32, 96), [(2, False, _stable(9, 1)), (0, True, _stable(17, 2)), (1, False, _stable(32, 4)), (0, False, _stable(4, 5))]
    elif case == "pair_1d":           # a causal / anticausal pair of different orders (same number of tail pieces) over 40 tiles of 128
        shape, scans = (5120,), [(0, True, _stable(9, 11)), (0, False, _stable(16, 12))]
    elif case == "pair_xy_tiles":     # pairs along x (4 tiles of 128) and y (3 tiles of 64): one final pass per dimension
        c = _stable(7, 13)
        shape, scans = (192, 512), [(0, True, c), (0, False, _stable(5, 14)), (1, True, c), (1, False, c)]
    elif case == "pair_then_single":  # x: pair + a third scan; y: an anticausal scan in front of a causal one (no pair)
        shape, scans = (64, 256), [(0, True, _stable(12, 15)), (0, False, _stable(12, 16)), (0, True, _stable(3, 17)),
                                   (1, False, _stable(6, 18)), (1, True, _stable(6, 19))]
    elif case == "wide_one_chain":    # 16384 lines fill the chip: 29 tiles of 32 are chained in ONE go (one level), not as chunks + propagation
        shape, scans = (16384, 32 * 29), [(0, True, _stable(9, 20)), (0, False, _stable(16, 21))]
    elif case == "ragged_xy":         # extents that no tile divides: padding where each scan leaves the image
        c = _stable(12, 3)
        shape, scans = (77, 300), [(0, True, c), (0, False, c), (1, True, c), (1, False, _stable(30, 8))]
    elif case == "ragged_1d_short":   # shorter than a tile, shorter than the order
        shape, scans = (20,), [(0, True, _stable(29, 4)), (0, False, _stable(9, 5))]
    elif case == "ragged_xyz":
        shape, scans = (50, 33, 68), [(2, False, _stable(9, 1)), (2, True, _stable(9, 2)), (0, False, _stable(17, 2)), (1, False, _stable(32, 4)), (1, True, _stable(4, 5))]
    else:                             # 1024 tiles of 32: three levels of the chain (1024 -> 64 -> 4)
        shape, scans = (32 * 1024 + 0,), [(0, False, _stable(32, 9)), (0, True, _stable(20, 10))]
        shape = (32 * 1031,)          # a prime number of tiles: partial last chunks on every level
    with rfa.Plan(shape, scans, clamped=clamped, path=capi.RF_PATH_TILED_MATRIX, device=capi.RF_DEVICE_HOST_ONLY) as plan:
        assert plan.path == capi.RF_PATH_TILED_MATRIX
        if case == "wide_one_chain":
            assert plan.tiles[0] == 32 and list(plan.table("mx_levels_0")) == [29.0, 0.0]
        if case.startswith("pair") or case in ("xy_pm_12", "wide_one_chain"):        # (the first scan of a dimension, in the plan's order, heads a pair)
            assert int(plan.table("mx_pair_0")[0]) == 1
        img = rc.random_image(shape, np.float32, 5)
        got = mxe.run(plan, img, clamped)
    want = oracle.apply_filter(img.astype(np.float64), scans, clamped)
    assert rc.rel_err(got, want) < 5e-5            # (f32 replay of up to five scans; the bar of the path is 1e-4)


def test_matrix_path_choice_and_refusals():
    H = dict(device=capi.RF_DEVICE_HOST_ONLY)
    hi = [(0, True, [1.0] + [0.01] * 15)]
    with rfa.Plan((1 << 16,), hi, **H) as plan:                       # automatic: orders above 8 in their direct form
        assert plan.path == capi.RF_PATH_TILED_MATRIX and plan.tiles[0] == 128
    with rfa.Plan((96, 160), [(1, False, _stable(9, 1))], clamped=True, **H) as plan:
        assert plan.path == capi.RF_PATH_TILED_MATRIX and tuple(plan.tiles)[:2] == (0, 96)
    with rfa.Plan((100, 160), [(1, False, _stable(9, 1))], **H) as plan:          # 100 rows: one tile of 128, padded
        assert plan.path == capi.RF_PATH_TILED_MATRIX and tuple(plan.tiles)[:2] == (0, 128)
    with rfa.Plan((100, 162), [(0, False, _stable(9, 1))], **H) as plan:          # a width that is not a multiple of 4
        assert plan.path != capi.RF_PATH_TILED_MATRIX
    with rfa.Plan((96, 160), [(1, False, [1.0] + [1.0] * 9)], dtype=np.int32, **H) as plan:
        assert plan.path != capi.RF_PATH_TILED_MATRIX
    with pytest.raises(rfa.RecFilterError):
        rfa.Plan((100, 162), [(0, False, _stable(9, 1))], path=capi.RF_PATH_TILED_MATRIX, **H)
    # chain levels whose transfer matrix rounds to zero in f32 are left out: the audio app's taps decay to 4e-5 across one tile
    # of 128 samples and to nothing across sixteen (pass 1, chain 0, apply 0, pass 2); a slowly decaying filter keeps the levels
    with rfa.Plan((1 << 20,), [(0, True, [1.0] + [0.01] * 29)], **H) as plan:
        assert plan.path == capi.RF_PATH_TILED_MATRIX and plan.num_kernels == 4
        assert list(plan.table("mx_levels_0").reshape(-1, 2)[:, 1]) == [0.0, 1.0, 1.0, 1.0]
    with rfa.Plan((1 << 20,), [(0, True, [0.01] + [0.099] * 10)], **H) as plan:          # sum of the taps 0.99
        assert plan.path == capi.RF_PATH_TILED_MATRIX and plan.num_kernels == 8
        assert list(plan.table("mx_levels_0").reshape(-1, 2)[:, 1]) == [0.0, 0.0, 0.0, 1.0]
    with rfa.Plan((1 << 20,), [(0, True, [1.0, 2.0, -1.0] + [0.0] * 6)], **H) as plan:          # an integrator: nothing decays
        assert plan.path == capi.RF_PATH_TILED_MATRIX and plan.num_kernels == 9
    # an ill-conditioned cascade is not sectioned (sections.h, sections_well_conditioned): order 7 of the audio app's
    # polynomial stays a direct form
    with rfa.Plan((1 << 16,), [(0, True, [1.0] + [0.01] * 3)], **H) as plan:
        assert plan.path in (capi.RF_PATH_TILED_FUSED, capi.RF_PATH_UNTILED)


def test_matrix_path_sharded_plan_structure():
    """A row-sharded filter of order 12 (host-only plans, no device): the matrix path, one exchange per scan along the sharded
    dimension, `lines x 8 ceil(k / 8)` floats per plane and rank; unequal slabs and 1-D signals are refused there."""
    H = dict(device=capi.RF_DEVICE_HOST_ONLY)
    c12, c20 = _stable(12, 3), _stable(20, 4)
    scans = [(0, True, c12), (1, True, c12), (1, False, c20), (0, False, c12)]
    for rank in range(4):
        with rfa.Plan((128, 512), scans, planes=2, shard_rank=rank, shard_world=4, **H) as plan:
            assert plan.path == capi.RF_PATH_TILED_MATRIX and plan.num_exchanges == 2
            assert plan.exchange_bytes(0) == 2 * 512 * 16 * 4 and plan.exchange_bytes(1) == 2 * 512 * 24 * 4
    with rfa.Plan((128, 512), scans, shard_rank=1, shard_world=2, shard_extents=[256, 128], **H) as plan:
        assert plan.path != capi.RF_PATH_TILED_MATRIX            # unequal slabs: the generic path's per-slab transfer tables
    with pytest.raises(rfa.RecFilterError):
        rfa.Plan((128, 512), scans, shard_rank=1, shard_world=2, shard_extents=[256, 128], path=capi.RF_PATH_TILED_MATRIX, **H)


def test_merged_runs_are_probed_over_the_whole_memory_of_the_cascade():
    """Consecutive same-direction scans of a 1-D signal become ONE direct-form scan only where (i) the f32 direct form tracks the
    f64 cascade until the cascade's response has died out -- not over a fixed 768 samples: slow poles drift later -- and (ii)
    the merged scan really runs in its direct form on the matrix path (no second rounding into sections); otherwise the scans
    stay as given."""
    n = 1 << 20
    row = 5 + 2 * capi.RF_MAX_ORDER

    def planned_orders(scans):
        with _host_plan((n,), scans) as p:
            return [int(r[2]) for r in p.table("scans").reshape(-1, row)], p.path
    fast = (0, True, [1.0, 0.1, 0.1])                       # poles at 0.37 and -0.27: five of them are one scan of order 10
    orders, path = planned_orders([fast] * 5)
    assert orders == [10] and path == capi.RF_PATH_TILED_MATRIX
    # two resonators with poles of radius 0.9995 (time constant 2000 samples): a 768-sample window cannot see the merged
    # f32 polynomial drift; the probe that follows the response to its end refuses the merge
    r, th = 0.9995, 0.01
    slow = (0, True, [1e-3, 2 * r * np.cos(th), -r * r])
    orders, _ = planned_orders([slow, slow])
    assert orders == [2, 2]
    # merged orders 5 and 4 that the planner would split into sections again: kept as given -- six scans, an in-plan cascade of
    # two fused stages (the table lists the first stage's four scans)
    mild = [(0, True, [0.5, 0.3, 0.1]), (0, True, [0.8, 0.2]), (0, True, [0.7, 0.2, -0.1]), (0, False, [0.6, 0.3]), (0, False, [0.9, 0.1, 0.05]), (0, False, [0.6, 0.4])]
    orders, path = planned_orders(mild)
    assert orders == [2, 1, 2, 1] and path == capi.RF_PATH_TILED_FUSED


def test_a_descriptor_of_another_abi_revision_is_refused():
    """rf_filter_desc.abi: a caller built against another revision of the header (RF_MAX_ORDER 8: other scan strides) gets an
    error instead of a mis-strided scans array."""
    d = capi.FilterDesc()
    d.ndim, d.n_planes, d.dtype, d.n_scans = 1, 1, capi.RF_F32, 1
    d.extent[0] = 64
    d.device = capi.RF_DEVICE_HOST_ONLY
    sc = (capi.ScanDesc * 1)()
    sc[0].dim, sc[0].causal, sc[0].order, sc[0].feedfwd = 0, 1, 1, 1.0
    sc[0].feedback[0] = 0.5
    d.scans = sc
    h = ctypes.c_void_p()
    for abi, want in ((0, capi.RF_ERR_INVALID_ARG), (2, capi.RF_ERR_INVALID_ARG), (capi.RF_ABI, capi.RF_OK)):
        d.abi = abi
        rc_ = capi.lib().rf_plan_create(ctypes.byref(d), ctypes.byref(h))
        assert rc_ == want, (abi, rc_)
        if rc_ == capi.RF_OK:
            capi.lib().rf_plan_destroy(h)
        else:
            assert b"abi" in capi.lib().rf_last_error_string()
    assert b"abi 3" in capi.lib().rf_version()
