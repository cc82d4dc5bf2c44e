"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs."""
import numpy as np
import pytest

import oracle
import ref_cases as rc

pytestmark = pytest.mark.gpu

from recfilter_amd import capi

TOL = 1e-4     # north_star: 1e-4 relative for floating point; integers bit-exact
TILED = capi.RF_PLAN_TILED_ONLY      # what tests/conftest.py makes the default of plans created without explicit flags


def _run(shape, scans, dtype=np.float32, clamped=False, planes=1, tile=None, path=0, seed=1234, inplace=False, flags=None):
    import torch
    import recfilter_amd as rfa
    imgs = [rc.random_image(shape, dtype, seed + i) for i in range(planes)]
    dev = [torch.from_numpy(im).cuda() for im in imgs]
    with rfa.Plan(shape, scans, dtype=dtype, clamped=clamped, planes=planes, tile=tile, path=path, flags=flags) as plan:
        outs = plan.execute(dev, dev if inplace else None)
        torch.cuda.synchronize()
        info = (plan.path, plan.tiles)
    return imgs, [o.cpu().numpy() for o in outs], info


def _check(imgs, outs, scans, clamped):
    for im, out in zip(imgs, outs):
        if np.issubdtype(im.dtype, np.integer):
            np.testing.assert_array_equal(out, oracle.apply_filter(im, scans, clamped))
        else:
            want = oracle.apply_filter(im.astype(np.float64), scans, clamped)
            err = rc.rel_err(out, want)
            assert err < TOL, f"rel err {err}"


@pytest.mark.parametrize("path", [1, 2, 4], ids=["untiled", "tiled_generic", "tiled_overlapped"])
@pytest.mark.parametrize("name", sorted(rc.REFERENCE_TESTS))
def test_reference_tests(name, path):
    """Every configuration of /root/reference/tests/test_*.cpp, literal shapes and tile widths.  tiled_overlapped is
    the reference's own structure: all dimensions in one pass 1 / one pass 2, cross-dimension residuals in between."""
    case = rc.REFERENCE_TESTS[name]
    nd = len(case["shape"])
    tile = [case["tile"] if any(s[0] == d for s in case["scans"]) else 0 for d in range(nd)]
    imgs, outs, (got_path, tiles) = _run(case["shape"], case["scans"], case["dtype"], case["clamped"],
                                         tile=tile if path != 1 else None, path=path)
    assert got_path == path
    if path != 1:
        assert list(tiles) == tile
    _check(imgs, outs, case["scans"], case["clamped"])


@pytest.mark.parametrize("path", [1, 2], ids=["untiled", "tiled_generic"])
@pytest.mark.parametrize("dtype", [np.float32, np.float64, np.int32, np.int16])
def test_pixel_types(dtype, path):
    scans = [(0, True, [1.0, 2.0, -1.0]), (1, False, [1.0, 1.0]), (0, False, [1.0, 1.0])]
    imgs, outs, _ = _run((40, 72), scans, dtype, path=path)
    _check(imgs, outs, scans, False)


@pytest.mark.parametrize("path", [1, 2], ids=["untiled", "tiled_generic"])
@pytest.mark.parametrize("coeff", [rc.BICUBIC_COEFF, rc.GAUSS2, rc.GAUSS3], ids=["bicubic", "gauss2", "gauss3"])
def test_clamped_border_xy(coeff, path):
    scans = rc.xy_pm(coeff)
    imgs, outs, _ = _run((96, 160), scans, clamped=True, planes=3, path=path)
    _check(imgs, outs, scans, True)


@pytest.mark.parametrize("path", [1, 2], ids=["untiled", "tiled_generic"])
def test_3d_and_1d_and_inplace(path):
    scans = rc.REFERENCE_TESTS["test_generic_xyz"]["scans"]
    imgs, outs, _ = _run((24, 32, 40), scans, path=path, inplace=True)
    _check(imgs, outs, scans, False)
    s1 = [(0, True, [1.0, 1.0])]
    imgs, outs, _ = _run((4096,), s1, path=path)        # BASELINE cfg1: 1-D prefix sum
    _check(imgs, outs, s1, False)


@pytest.mark.parametrize("dtype", [np.float32, np.float64, np.int32, np.int16])
@pytest.mark.parametrize("clamped", [False, True])
def test_overlapped_3d_all_residual_pairs(dtype, clamped):
    """RF_PATH_TILED_OVERLAPPED on a 3-D filter with different tiles / orders per dimension: x->y, x->z and y->z
    residuals (lib/split.cpp:1814-1820), causal + anticausal scans in every dimension, every pixel type; also a filter
    whose x dimension has no scans (y->z only) and the automatic path choice for split() filters."""
    if np.issubdtype(dtype, np.integer):
        scans = [(0, True, [1.0, 1.0]), (0, False, [1.0, 2.0, -1.0]), (1, False, [1.0, 1.0]), (1, True, [1.0, 3.0, -3.0, 1.0]),
                 (2, True, [1.0, 1.0]), (2, False, [1.0, 1.0])]
    else:
        scans = [(0, True, [0.6, 0.3, -0.1]), (0, False, [0.7, 0.2]), (1, False, [0.5, 0.4, -0.1, 0.05]), (1, True, [0.8, 0.1]),
                 (2, True, [0.9, 0.05]), (2, False, [0.5, 0.3, 0.1])]
    shape, tile = (24, 20, 48), [16, 5, 4]
    imgs, outs, (path, tiles) = _run(shape, scans, dtype, clamped, planes=2, tile=tile, path=4)
    assert path == 4 and list(tiles) == tile
    _check(imgs, outs, scans, clamped)
    imgs, outs, (path, tiles) = _run(shape, scans, dtype, clamped, tile=tile, path=0)
    # split() along several dimensions: the fused kernels where they apply (f32 / f64 / i32 / i16, orders <= 3), else overlapped
    assert path == 3
    _check(imgs, outs, scans, clamped)
    yz = [s for s in scans if s[0] != 0]
    imgs, outs, (path, _) = _run(shape, yz, dtype, clamped, tile=[0, 5, 4], path=4, inplace=True)
    assert path == 4
    _check(imgs, outs, yz, clamped)


def test_overlapped_2d_large_and_timed():
    import torch
    import recfilter_amd as rfa
    scans = rc.xy_pm(rc.GAUSS3)
    imgs, outs, (path, _) = _run((512, 768), scans, clamped=True, tile=[32, 32], path=4)
    assert path == 4
    _check(imgs, outs, scans, True)
    with rfa.Plan((128, 128), scans, clamped=True, tile=[32, 32], path=4) as plan:
        _, times = plan.execute_timed([torch.from_numpy(rc.random_image((128, 128))).cuda()])
        assert [n for n, _ in times] == ["overlap_pass1", "carry_x", "overlap_residual_y", "carry_y", "overlap_pass2"]


def test_high_order_and_ragged_extents():
    scans = [(0, True, [0.5, 0.3, 0.2, -0.1, 0.05, 0.02]), (1, False, [1.0, 0.4, 0.1, 0.05, 0.02, 0.01, 0.005, 0.001])]
    for shape in [(7, 13), (1, 1), (3, 1), (1, 5), (33, 97)]:
        imgs, outs, _ = _run(shape, scans, path=1, clamped=True)
        _check(imgs, outs, scans, True)
    imgs, outs, (path, tiles) = _run((33, 97), scans, path=0)   # auto: no tile divides 97 -> still on the GPU
    _check(imgs, outs, scans, False)


def test_front_end_mirror_of_test_trivial():
    """tests/test_trivial.cpp through the RecFilter front-end mirror."""
    import torch
    import recfilter_amd as rfa
    width = height = 20
    tile = 4
    image = torch.ones((height, width), dtype=torch.float32, device="cuda")
    x, y = rfa.RecFilterDim("x", width), rfa.RecFilterDim("y", height)
    f = rfa.RecFilter()
    f[x, y] = image
    f.add_filter(+x, [1.0, 1.0])
    f.add_filter(+y, [1.0, 1.0])
    f.split(x, tile, y, tile)
    out = f.realize()[0].cpu().numpy()
    yy, xx = np.mgrid[0:height, 0:width]
    np.testing.assert_array_equal(out, ((xx + 1) * (yy + 1)).astype(np.float32))
    assert f.profile(3) > 0.0


def test_cascade_and_overlap_front_end():
    import torch
    import recfilter_amd as rfa
    img = rc.random_image((64, 64))
    x, y = rfa.RecFilterDim("x", 64), rfa.RecFilterDim("y", 64)
    F = rfa.RecFilter("G")
    F.set_clamped_image_border()
    F[x, y] = torch.from_numpy(img).cuda()
    W1, W2 = rfa.gaussian_weights(5.0, 1), rfa.gaussian_weights(5.0, 2)
    for W in (W1, W2):
        F.add_filter(+x, W); F.add_filter(-x, W); F.add_filter(+y, W); F.add_filter(-y, W)
    fc = F.cascade([0, 1, 2, 3], [4, 5, 6, 7])       # apps/gaussian/gaussian_filter_1xy_2xy.cpp:54
    for f in fc:
        f.split_all_dimensions(16)
    out = fc[-1].realize()[0].cpu().numpy()
    want = oracle.apply_filter(img.astype(np.float64), F._contents["scans"], True)
    assert rc.rel_err(out, want) < TOL
    by_dim = F.cascade_by_dimension()
    out2 = by_dim[-1].realize()[0].cpu().numpy()
    assert rc.rel_err(out2, want) < TOL
    # A cascade IS the filter it was made from (scans of different dimensions commute, the scans of a dimension stay in
    # order): the last stage runs ONE plan with the scans of every stage -- four per dimension of order <= 2 here, one
    # fused stage instead of two -- and an upstream stage asked for its own result still gives that.
    assert fc[-1]._contents["merged_stages"] == 2 and by_dim[-1]._contents["merged_stages"] == 2
    assert fc[-1].plan().num_kernels <= 5
    first = fc[0].realize()[0].cpu().numpy()
    assert rc.rel_err(first, oracle.apply_filter(img.astype(np.float64), F._contents["scans"][:4], True)) < TOL
    rfa.RecFilter.merge_cascades = False
    try:
        chained = F.cascade([0, 1, 2, 3], [4, 5, 6, 7])
        for f in chained:
            f.split_all_dimensions(16)
        out3 = chained[-1].realize()[0].cpu().numpy()
        assert chained[-1]._contents["merged_stages"] == 0 and rc.rel_err(out3, want) < TOL
    finally:
        rfa.RecFilter.merge_cascades = True
    assert rc.rel_err(out, out3.astype(np.float64)) < 1e-5


def test_cascade_whose_merged_plan_does_not_exist_runs_as_a_chain():
    """36 scans in nine stages: every stage's own plan exists, ONE plan of all 36 does not (RF_MAX_SCANS = 32) -- the last stage
    falls back to the chain of per-stage plans instead of failing; and a stage compiled while merge_cascades was on is
    planned again once it is switched off."""
    import torch
    import recfilter_amd as rfa
    img = rc.random_image((64, 256), np.float32, 21)
    x, y = rfa.RecFilterDim("x", 256), rfa.RecFilterDim("y", 64)
    F = rfa.RecFilter("Many")
    F[x, y] = torch.from_numpy(img).cuda()
    for i in range(18):
        F.add_filter(+x if i % 2 == 0 else -x, [0.6, 0.4])
        F.add_filter(+y if i % 2 == 0 else -y, [0.7, 0.3])
    stages = F.cascade(*[list(range(4 * i, 4 * i + 4)) for i in range(9)])
    for f in stages:
        f.split_all_dimensions(32)
    out = stages[-1].realize()[0].cpu().numpy()
    want = oracle.apply_filter(img.astype(np.float64), F._contents["scans"], False)
    assert stages[-1]._contents["merged_stages"] == 0 and rc.rel_err(out, want) < TOL
    mid = stages[3].realize()[0].cpu().numpy()             # (an upstream stage asked for by itself: 16 scans, one merged plan)
    assert rc.rel_err(mid, oracle.apply_filter(img.astype(np.float64), F._contents["scans"][:16], False)) < TOL
    F2 = rfa.RecFilter("Eight")
    F2[x, y] = torch.from_numpy(img).cuda()
    for dim_causal_coeff in F._contents["scans"][:8]:
        d, causal, coeff = dim_causal_coeff
        F2.add_filter((+x if causal else -x) if d == 0 else (+y if causal else -y), coeff)
    two = F2.cascade(list(range(0, 4)), list(range(4, 8)))
    for f in two:
        f.split_all_dimensions(32)
    a = two[-1].realize()[0].cpu().numpy()
    assert two[-1]._contents["merged_stages"] == 2
    rfa.RecFilter.merge_cascades = False
    try:
        b = two[-1].realize()[0].cpu().numpy()                 # compiled merged above: planned again as a chain
        assert two[-1]._contents["merged_stages"] == 0
    finally:
        rfa.RecFilter.merge_cascades = True
    want8 = oracle.apply_filter(img.astype(np.float64), F._contents["scans"][:8], False)
    assert rc.rel_err(a, want8) < TOL and rc.rel_err(b, want8) < TOL


def test_cascade_and_overlap_keep_the_prologue():
    """A filter defined on `scale*in + bias` (demo/demo_gaussian_filter.cpp:51-53) keeps that defining expression when
    it is cascaded or merged into a higher-order filter: stage 0 / the merged filter read the image through it."""
    import torch
    import recfilter_amd as rfa
    img = rc.random_image((64, 256), seed=77) * 255.0
    dev = torch.from_numpy(img).cuda()
    x, y = rfa.RecFilterDim("x", 256), rfa.RecFilterDim("y", 64)
    W1, W2 = rfa.gaussian_weights(5.0, 1), rfa.gaussian_weights(5.0, 2)
    scale, bias = 1.0 / 255.0, 0.125

    def build():
        F = rfa.RecFilter("P")
        F.set_clamped_image_border()
        F.define([x, y], dev, scale=scale, bias=bias)
        for W in (W1, W2):
            F.add_filter(+x, W); F.add_filter(-x, W); F.add_filter(+y, W); F.add_filter(-y, W)
        return F
    F = build()
    want = oracle.apply_filter(img.astype(np.float64) * np.float32(scale) + np.float32(bias), F._contents["scans"], True)
    whole = build()
    whole.split_all_dimensions(32)
    assert rc.rel_err(whole.realize()[0].cpu().numpy(), want) < TOL
    for stages in (F.cascade([0, 1, 2, 3], [4, 5, 6, 7]), build().cascade_by_dimension()):
        assert stages[0]._contents["prologue"] == (scale, bias) and all(s._contents["prologue"] is None for s in stages[1:])
        assert rc.rel_err(stages[-1].realize()[0].cpu().numpy(), want) < TOL
    # overlap: the two cascade stages merged back into one filter of orders 3
    fa, fb = build().cascade([0, 1, 2, 3], [4, 5, 6, 7])
    merged = fb.overlap_to_higher_order_filter(fa)
    assert merged._contents["prologue"] == (scale, bias)
    got = merged.realize()[0].cpu().numpy()
    # (with a clamped border the merged filter differs from the cascade near the border, so compare with the oracle
    # run on the merged filter's own scans)
    want_m = oracle.apply_filter(img.astype(np.float64) * np.float32(scale) + np.float32(bias), merged._contents["scans"], True)
    assert rc.rel_err(got, want_m) < TOL
    # a rejected re-definition must not change the prologue of the filter it was rejected for
    with pytest.raises(rfa.RecFilterUsageError):
        whole.define([x, y], dev, scale=3.0)
    assert whole._contents["prologue"] == (scale, bias)


def test_timed_execute_reports_every_kernel():
    import torch
    import recfilter_amd as rfa
    scans = rc.xy_pm(rc.GAUSS2)
    img = torch.from_numpy(rc.random_image((128, 128))).cuda()
    with rfa.Plan((128, 128), scans, clamped=True, path=2) as plan:
        outs, times = plan.execute_timed([img])
        assert len(times) == plan.num_kernels == 2 * 4   # per dimension: pass1, carry x2, pass2
        assert all(ms >= 0 for _, ms in times)


# ---- fused x/y path (kernels_fused.hip) ---------------------------------------------------------------
@pytest.mark.parametrize("name", sorted(rc.FUSED_CASES))
def test_fused_small_cases(name):
    case = rc.FUSED_CASES[name]
    imgs, outs, (path, tiles) = _run(case["shape"], case["scans"], clamped=case["clamped"], path=3)
    assert path == 3 and tiles[0] == 256
    _check(imgs, outs, case["scans"], case["clamped"])


@pytest.mark.parametrize("cfg", ["cfg2_summed_table", "cfg3_gaussian2_xy", "cfg4a_bicubic_rgb", "cfg4b_gaussian3_rgb"])
def test_fused_baseline_configs_reduced(cfg):
    """BASELINE configs 2-4 at 1024^2 (the oracle finishes in seconds); auto path must pick the fused kernels."""
    c = rc.BASELINE_CONFIGS[cfg]
    imgs, outs, (path, tiles) = _run((1024, 1024), c["scans"], clamped=c["clamped"], planes=c.get("planes", 1))
    assert path == 3
    _check(imgs, outs, c["scans"], c["clamped"])


def test_fused_summed_table_int32_bit_exact():
    scans = rc.BASELINE_CONFIGS["cfg2_summed_table"]["scans"]
    imgs, outs, (path, _) = _run((512, 1024), scans, dtype=np.int32)
    assert path == 3
    np.testing.assert_array_equal(outs[0], imgs[0].astype(np.int64).cumsum(0).cumsum(1).astype(np.int32))
    # second-order integral with wrap-around: still bit-exact
    s2 = [(0, True, [1.0, 2.0, -1.0]), (1, True, [1.0, 2.0, -1.0]), (0, False, [1.0, 1.0])]
    imgs, outs, _ = _run((256, 512), s2, dtype=np.int32)
    _check(imgs, outs, s2, False)


def test_fused_3d_generic_xyz():
    """BASELINE config 5 at 64 x 128 x 256: fused x/y per plane, then z."""
    scans = rc.BASELINE_CONFIGS["cfg5_generic_xyz"]["scans"]
    imgs, outs, (path, tiles) = _run((64, 128, 256), scans)
    assert path == 3
    _check(imgs, outs, scans, False)


def test_fused_inplace_and_non_square():
    scans = rc.xy_pm(rc.GAUSS2)
    imgs, outs, (path, _) = _run((96, 1280), scans, clamped=True, inplace=True)     # TY = 32
    assert path == 3
    _check(imgs, outs, scans, True)


def test_fused_matches_untiled_gpu_at_full_size_properties():
    """16384^2 (BASELINE config 3 at full size) is too large for the CPU oracle inside a test, so check
    size-independent properties: (i) a clamped filter with b+sum(a)=1 keeps a constant image constant,
    (ii) linearity: F(a*u + v) = a*F(u) + F(v), (iii) agreement with the untiled GPU path on a crop of rows
    is covered at 2048^2 where the oracle is still affordable."""
    import torch
    import recfilter_amd as rfa
    n = 16384
    scans = rc.xy_pm(rc.GAUSS2)
    with rfa.Plan((n, n), scans, clamped=True) as plan:
        assert plan.path == 3
        const = torch.full((n, n), 3.0, device="cuda")
        out = plan.execute([const])[0]
        assert float((out - 3.0).abs().max()) < 3e-4          # sum of f32 coefficients differs from 1 by ~1e-7
        g = torch.Generator(device="cuda").manual_seed(5)
        u = torch.rand((n, n), device="cuda", generator=g)
        v = torch.rand((n, n), device="cuda", generator=g)
        fu = plan.execute([u])[0]
        fv = plan.execute([v])[0]
        fw = plan.execute([2.5 * u + v])[0]
        lin = float((fw - (2.5 * fu + fv)).abs().max() / fw.abs().max())
        assert lin < 1e-5
        del const, out, fw
        # crop check against the oracle: the first and last 256 rows depend on every row only through
        # carries, so compare a 2048-column band of the full result with the oracle run on full columns
    small = rc.random_image((2048, 2048))
    with rfa.Plan((2048, 2048), scans, clamped=True) as plan:
        got = plan.execute([torch.from_numpy(small).cuda()])[0].cpu().numpy()
    want = oracle.apply_filter(small.astype(np.float64), scans, True, threads=8)
    assert rc.rel_err(got, want) < TOL


# ---- sharded execution (stepping API of the C ABI), all "ranks" emulated on the one GPU of the test box ----
def _run_sharded(shape, scans, clamped, world, path, planes=1, dtype=np.float32, extents=None, tile=None, flags=None,
                 interior=True):
    import torch
    import recfilter_amd as rfa
    if np.issubdtype(dtype, np.integer):
        full = [np.random.default_rng(31 + p).integers(0, 4, size=shape).astype(dtype) for p in range(planes)]
    else:
        full = [rc.random_image(shape, dtype, 31 + p) for p in range(planes)]
    ext = list(extents) if extents is not None else [shape[0] // world] * world      # slab extents along the sharded dimension
    assert sum(ext) == shape[0] and len(ext) == world
    lo = [sum(ext[:r]) for r in range(world)]
    plans = [rfa.Plan((ext[r],) + tuple(shape[1:]), scans, dtype=dtype, clamped=clamped, planes=planes, path=path, tile=tile,
                      shard_rank=r, shard_world=world, shard_extents=extents, flags=flags)
             for r in range(world)]
    ins = [[torch.from_numpy(np.ascontiguousarray(f[lo[r]:lo[r] + ext[r]])).cuda() for f in full] for r in range(world)]
    outs = [[torch.empty_like(t) for t in ins[r]] for r in range(world)]
    for r in range(world):
        plans[r].begin(ins[r], outs[r])
    nex = plans[0].num_exchanges
    for e in range(nex):
        nbytes = plans[0].exchange_bytes(e)
        gathered = torch.empty(world * nbytes, dtype=torch.uint8, device="cuda")
        for r in range(world):                         # "all-gather": every rank's send lands rank-major
            plans[r].exchange_local(e, gathered.data_ptr() + r * nbytes)
        if interior and e == nex - 1:                  # what a driver runs beside the collective (rf_plan_interior);
            for r in range(world):                     # without the call exchange_apply / finish run it themselves
                plans[r].interior()
        for r in range(world):
            plans[r].exchange_apply(e, gathered.data_ptr())
    for r in range(world):
        plans[r].finish()
    torch.cuda.synchronize()
    got = [np.concatenate([outs[r][p].cpu().numpy() for r in range(world)], axis=0) for p in range(planes)]
    info = (plans[0].path, nex)
    _run_sharded.has_interior = plans[0].has_interior
    for p in plans:
        p.close()
    return full, got, info


@pytest.mark.parametrize("world", [2, 4, 8])
def test_sharded_fused_2d_rows(world):
    """cfg3's filter, rows sharded over `world` slabs: ONE exchange for both y scans (merged exchange)."""
    scans = rc.xy_pm(rc.GAUSS2)
    full, got, (path, nex) = _run_sharded((64 * 2 * world, 512), scans, True, world, path=0, planes=2)
    assert path == 3 and nex == 1
    _check(full, got, scans, True)


@pytest.mark.parametrize("extents", [[128, 64], [64, 192, 128], [64, 128, 64, 192, 64], [96, 32, 64]])
def test_sharded_fused_2d_rows_unequal_slabs(extents):
    """Row slabs of different heights (rf_filter_desc.shard_extents): the tile height comes from the slabs' common divisor
    (32 rows for the last case), every slab's exit transfer from its own tile count; still ONE exchange, applied by pass 2."""
    scans = rc.xy_pm(rc.GAUSS2)
    ints = [(0, True, [1.0, 2.0, -1.0]), (1, True, [1.0, 2.0, -1.0]), (1, False, [1.0, 1.0])]      # bit-exact in the int32 ring
    for dtype, sc, clamped in ((np.float32, scans, True), (np.int32, ints, False)):
        full, got, (path, nex) = _run_sharded((sum(extents), 512), sc, clamped, len(extents), path=0, planes=2, dtype=dtype,
                                              extents=extents)
        assert path == 3 and nex == 1
        _check(full, got, sc, clamped)


def test_sharded_unequal_slabs_other_paths():
    """Unequal slabs on the strided z stage (merged exchange + merged_apply), on the generic path with one exchange per
    scan (five scans along y: A^(tiles of a slab) per slab) and with the merged exchange of the generic path."""
    scans = rc.BASELINE_CONFIGS["cfg5_generic_xyz"]["scans"]
    full, got, (path, nex) = _run_sharded((64 + 128 + 64, 32, 256), scans, False, 3, path=0, extents=[64, 128, 64])
    assert path == 3 and nex == 1
    _check(full, got, scans, False)
    five = [(1, bool(i % 2), [0.5, 0.4 - 0.05 * i]) for i in range(5)] + [(0, True, [0.7, 0.3])]
    full, got, (path, nex) = _run_sharded((24 + 8 + 16, 40), five, True, 3, path=2, extents=[24, 8, 16], tile=[8, 8])
    assert path == 2 and nex == 5
    _check(full, got, five, True)
    three = rc.xy_pm(rc.GAUSS2)
    full, got, (path, nex) = _run_sharded((24 + 8 + 16, 40), three, True, 3, path=2, extents=[24, 8, 16], tile=[8, 8])
    assert path == 2 and nex == 1
    _check(full, got, three, True)


def test_high_order_filters_outside_the_matrix_paths_shape_rules():
    """Orders above 8 run in their direct form on the matrix path, whose passes move 16 bytes per lane and whose slabs are equal
    (plan_matrix.cpp: matrix_plan_applicable).  Outside those rules -- a width that is no multiple of four, slabs of different
    extents (the reference only asks that the tile divide the extent, lib/recfilter.h:311) -- the automatic choice falls back to
    the generic tiled path, which takes every order up to 32 (the reach is there, not the matrix cores' speed)."""
    from test_gpu_high_order import stable_coeff, MX
    c12, c9 = stable_coeff(12, 51), stable_coeff(9, 52)
    scans = [(0, True, c12), (0, False, c9), (1, True, c9), (1, False, c12)]
    for shape in ((96, 301), (130, 258)):
        for clamped in (False, True):
            imgs, outs, (path, _) = _run(shape, scans, clamped=clamped)
            assert path not in (MX, 3), (shape, path)
            _check(imgs, outs, scans, clamped)
    full, got, (path, nex) = _run_sharded((64 + 128 + 32, 256), scans, True, 3, path=0, extents=[64, 128, 32])
    assert path != MX and nex >= 1
    _check(full, got, scans, True)
    # (equal slabs of the same filter do take the matrix path)
    full, got, (path, nex) = _run_sharded((64 * 3, 256), scans, True, 3, path=0)
    assert path == MX
    _check(full, got, scans, True)


def test_shard_extents_are_validated():
    import recfilter_amd as rfa
    scans = rc.xy_pm(rc.GAUSS2)
    with pytest.raises(rfa.RecFilterError):          # this rank's entry is not its extent
        rfa.Plan((128, 512), scans, clamped=True, shard_rank=0, shard_world=2, shard_extents=[64, 128])
    with pytest.raises(rfa.RecFilterError):
        rfa.Plan((128, 512), scans, clamped=True, shard_rank=0, shard_world=2, shard_extents=[128, 0])
    with pytest.raises(ValueError):
        rfa.Plan((128, 512), scans, clamped=True, shard_rank=0, shard_world=2, shard_extents=[128])


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_3d_z_slabs(world):
    """cfg5's filter (tests/test_generic_xyz.cpp), z sharded: fused x/y per plane, exchanges on the z scans."""
    scans = rc.BASELINE_CONFIGS["cfg5_generic_xyz"]["scans"]
    full, got, (path, nex) = _run_sharded((16 * world, 64, 256), scans, False, world, path=0)
    assert path == 3 and nex == 1
    _check(full, got, scans, False)


@pytest.mark.parametrize("world", [1, 2, 4, 8])
@pytest.mark.parametrize("clamped", [False, True])
def test_sharded_3d_early_exchange(world, clamped):
    """A z-sharded volume on the strided z stage exchanges the z carries of the RAW input (the z operators commute with the
    x/y filter), runs its x/y stage as the exchange-independent work (rf_plan_interior) and filters the completed carry
    planes along x/y afterwards (plan_strided.h, "early exchange").  Emulated ranks; against the oracle on the whole
    volume, and equal within rounding to the late exchange (RF_PLAN_LATE_EXCHANGE: carries of the filtered data).
    world = 1: one slab built with the exchange structure (RF_PLAN_FORCE_EXCHANGE)."""
    scans = rc.BASELINE_CONFIGS["cfg5_generic_xyz"]["scans"]
    force = capi.RF_PLAN_FORCE_EXCHANGE if world == 1 else 0
    for shape, extents, planes in (((32 * world, 64, 256), None, 1), ((64 * world + 64, 40, 516), [128] + [64] * (world - 1), 2)):
        if extents is not None and world == 1:
            extents = None
        full, got, (path, nex) = _run_sharded(shape, scans, clamped, world, path=0, planes=planes, extents=extents, flags=TILED | force)
        assert path == 3 and nex == 1 and _run_sharded.has_interior, (shape, path, nex)
        _check(full, got, scans, clamped)
        # the driver that never calls rf_plan_interior: exchange_apply runs the pending work itself
        _, lazy, _ = _run_sharded(shape, scans, clamped, world, path=0, planes=planes, extents=extents, flags=TILED | force, interior=False)
        for a, b in zip(got, lazy):
            assert np.array_equal(a, b)
        _, late, _ = _run_sharded(shape, scans, clamped, world, path=0, planes=planes, extents=extents,
                                  flags=TILED | force | capi.RF_PLAN_LATE_EXCHANGE)
        assert not _run_sharded.has_interior
        for a, b in zip(got, late):
            assert rc.rel_err(a, b.astype(np.float64)) < 1e-5
    # integer pixels: the ring arithmetic commutes exactly -- bit-exact against the oracle
    ints = [(0, True, [1.0, 1.0]), (1, True, [1.0, 2.0, -1.0]), (2, True, [1.0, 1.0]), (2, False, [1.0, 1.0, -1.0])]
    full, got, (path, nex) = _run_sharded((32 * world, 32, 256), ints, False, world, path=0, dtype=np.int32, flags=TILED | force)
    assert path == 3 and nex == 1 and _run_sharded.has_interior
    _check(full, got, ints, False)
    # a prologue is not linear in the carries (its bias): such plans keep the late exchange
    import recfilter_amd as rfa
    with rfa.Plan((64, 64, 256), scans, shard_rank=0, shard_world=2, prologue=(0.5, 0.25)) as plan:
        assert plan.path == 3 and not plan.has_interior


@pytest.mark.parametrize("case", ["rows_2d", "rows_2d_int", "generic_2d"])
def test_forced_exchange_on_one_slab(case):
    """RF_PLAN_FORCE_EXCHANGE: ONE slab built and driven like one of several -- per-scan launches around the exchange
    point, exit carries, the gather walk (over one slab: zero entering carries), the correction inside the final pass --
    gives the plain filter.  What lets a one-GPU box run every call of an N-GPU rank, the RCCL all-gather included
    (tests/test_dist_gpu.py)."""
    import torch
    import recfilter_amd as rfa
    if case == "rows_2d":
        shape, scans, clamped, dtype, path = (256, 768), rc.xy_pm(rc.GAUSS2), True, np.float32, 0
    elif case == "rows_2d_int":
        shape, scans, clamped, dtype, path = (192, 516), [(0, True, [1.0, 1.0]), (1, True, [1.0, 2.0, -1.0]), (1, False, [1.0, 1.0])], False, np.int32, 0
    else:
        shape, scans, clamped, dtype, path = (48, 40), rc.REFERENCE_TESTS["test_generic_xy"]["scans"], False, np.float32, 2
    full, got, (p, nex) = _run_sharded(shape, scans, clamped, 1, path=path, planes=2, dtype=dtype, tile=[8, 8] if path == 2 else None,
                                       flags=TILED | capi.RF_PLAN_FORCE_EXCHANGE)
    assert p == (3 if path == 0 else 2) and nex == 1
    _check(full, got, scans, clamped)
    with rfa.Plan(shape, scans, dtype=dtype, clamped=clamped, flags=TILED | capi.RF_PLAN_FORCE_EXCHANGE, path=path,
                  tile=[8, 8] if path == 2 else None) as plan:
        with pytest.raises(rfa.RecFilterError, match="begin/exchange/finish"):           # insists on the stepping calls
            plan.execute([torch.zeros(shape, device="cuda", dtype=torch.int32 if dtype == np.int32 else torch.float32)])


def test_sharded_generic_path_uneven_scans():
    scans = rc.REFERENCE_TESTS["test_generic_xy"]["scans"]      # 4 x scans, 3 y scans
    full, got, (path, nex) = _run_sharded((48, 40), scans, False, 3, path=2)
    assert path == 2 and nex == 1
    _check(full, got, scans, False)
    with pytest.raises(Exception):                                # a sharded plan refuses the one-shot execute
        import torch
        import recfilter_amd as rfa
        p = rfa.Plan((16, 40), scans, shard_rank=0, shard_world=3, path=2)
        p.execute([torch.zeros((16, 40), device="cuda")])


def test_cpp_front_end_runs_the_reference_tests():
    """include/recfilter.hpp (the Halide-free RecFilter front-end) driving the C ABI from C++:
    tests/cpp/test_frontend.cpp restates test_trivial / test_generic_xy / test_generic_xyz /
    test_type_invariance / test_overlap_filter_order and the gaussian cascade app."""
    import os
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    exe = os.path.join(here, "cpp", "test_frontend")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(here, "cpp")])
    res = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    print(res.stdout)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "all front-end tests passed" in res.stdout


def test_cpp_front_end_over_real_rccl():
    """a C++ caller whose all-gather is ncclAllGather on a communicator it initialised itself
    (ncclCommInitRank, one rank: the pool's boxes have one GPU), handed to RecFilter::realize_sharded as
    include/recfilter.hpp documents.  Row shards (collective on the filter's stream) and z slabs (collective on the side
    stream, rf_plan_interior beside it) on plans with RF_PLAN_FORCE_EXCHANGE, against realize() and raster loops.  A fresh
    child process linked against librccl: nothing of this (torch-initialised) process is involved."""
    import os
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    exe = os.path.join(here, "cpp", "test_frontend_rccl")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(here, "cpp"), "test_frontend_rccl"])
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "RCCL_ID_FILE")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    res = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
    print(res.stdout)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "all rccl front-end tests passed" in res.stdout and "RCCL communicator: 1 rank(s)" in res.stdout
    assert "(3 ncclAllGather calls on 1 rank(s))" in res.stdout


# ---- randomized filters: every path against the oracle ------------------------------------------------------
def _random_filter(rng, ndim, max_order=3, max_per_dim=4, stable=True):
    scans = []
    for d in range(ndim):
        for _ in range(int(rng.integers(0, max_per_dim + 1))):
            k = int(rng.integers(1, max_order + 1))
            # feedback with sum |a| < 0.9 keeps the response decaying so f32 stays well conditioned
            a = rng.uniform(-1.0, 1.0, size=k)
            a *= rng.uniform(0.2, 0.9) / np.sum(np.abs(a))
            scans.append((d, bool(rng.integers(0, 2)), [float(rng.uniform(0.3, 1.5))] + [float(v) for v in a]))
    if not scans:
        scans.append((0, True, [1.0, 0.5]))
    return scans


@pytest.mark.parametrize("seed", range(24))
def test_random_filters_fused_and_generic(seed):
    rng = np.random.default_rng(1000 + seed)
    ndim = 2 if seed % 3 else 3
    if ndim == 2:
        shape = (32 * int(rng.integers(1, 7)), 256 * int(rng.integers(1, 4)))
    else:
        shape = (int(rng.choice([8, 32, 64, 96])), 32 * int(rng.integers(1, 4)), 256 * int(rng.integers(1, 3)))
    scans = _random_filter(rng, ndim)
    clamped = bool(rng.integers(0, 2))
    planes = int(rng.integers(1, 3))
    imgs, outs, (path, tiles) = _run(shape, scans, clamped=clamped, planes=planes, seed=seed)
    _check(imgs, outs, scans, clamped)
    # the same filter through the generic tiled path must agree too
    imgs2, outs2, (path2, _) = _run(shape, scans, clamped=clamped, planes=planes, seed=seed, path=2)
    assert path2 == 2
    _check(imgs2, outs2, scans, clamped)


@pytest.mark.parametrize("seed", range(6))
def test_random_integer_filters_bit_exact(seed):
    rng = np.random.default_rng(2000 + seed)
    shape = (32 * int(rng.integers(1, 5)), 256 * int(rng.integers(1, 3)))
    scans = []
    for d in range(2):
        for _ in range(int(rng.integers(1, 4))):
            k = int(rng.integers(1, 4))
            scans.append((d, bool(rng.integers(0, 2)), [float(rng.integers(1, 4))] + [float(rng.integers(-3, 4)) for _ in range(k)]))
    clamped = bool(rng.integers(0, 2))
    imgs, outs, (path, _) = _run(shape, scans, dtype=np.int32, clamped=clamped, seed=seed)
    assert path == 3
    _check(imgs, outs, scans, clamped)          # wrap-around arithmetic, bit-exact
    # int16 pixels (the reference's tests/test_type_invariance.cpp) ride the same fused kernels: 16-bit planes, 32-bit ring
    imgs, outs, (path, _) = _run(shape, scans, dtype=np.int16, clamped=clamped, seed=seed, planes=2)
    assert path == 3
    _check(imgs, outs, scans, clamped)


def test_int16_fused_partial_tiles_3d_and_reference_shape():
    """int16 on the fused path beyond whole tiles: widths that are multiples of 4 only, partial tile rows, a volume
    (fused x/y + strided z), in place, and the literal tests/test_type_invariance.cpp filter on a large image."""
    s2 = rc.REFERENCE_TESTS["test_type_invariance"]["scans"]
    for shape in [(70, 300), (33, 20), (135, 1000)]:
        imgs, outs, (path, _) = _run(shape, s2, dtype=np.int16, seed=5)
        assert path == 3
        _check(imgs, outs, s2, False)
    s3 = [(0, True, [1.0, 1.0]), (0, False, [1.0, 2.0, -1.0]), (1, False, [1.0, 1.0]), (2, True, [1.0, 1.0, -1.0]), (2, False, [1.0, 1.0])]
    imgs, outs, (path, _) = _run((64, 96, 512), s3, dtype=np.int16, seed=6, inplace=True)
    assert path == 3
    _check(imgs, outs, s3, False)
    imgs, outs, (path, _) = _run((2048, 2048), s2, dtype=np.int16, clamped=True, seed=7)
    assert path == 3
    _check(imgs, outs, s2, True)


# ---- pointwise prologue / epilogue (rf_pointwise_desc; compute_at of a pointwise consumer) ----------------
def _pointwise_err(out, img, scans, clamped, prologue, epilogue):
    """The error of a plan with pointwise stages, judged against the magnitude of the terms the epilogue adds up
    (rc.pointwise_want / rc.rel_err_scaled): the terms may cancel, and a pointwise relative error of the cancelled result
    is ill-conditioned for any f32 implementation, the reference operator included."""
    want, scale = rc.pointwise_want(img, scans, clamped, prologue, epilogue)
    return rc.rel_err(out, want, scale=scale)


@pytest.mark.parametrize("path", [0, 1, 2], ids=["auto_fused", "untiled", "tiled_generic"])
@pytest.mark.parametrize("which", ["pre", "post", "both", "post_no_input"])
def test_pointwise_stages_all_paths(path, which):
    import torch
    import recfilter_amd as rfa
    shape = (128, 512)
    scans = rc.BASELINE_CONFIGS["cfg3_gaussian2_xy"]["scans"]
    prologue = (1.0 / 255.0, 0.125) if which in ("pre", "both") else None
    epilogue = {"pre": None, "post": (-1.0, 2.0, 0.0), "both": (-0.75, 1.75, 0.5), "post_no_input": (3.0, 0.0, -1.0)}[which]
    img = (rc.random_image(shape, np.float32, 7) * (255.0 if prologue else 1.0)).astype(np.float32)
    dev = torch.from_numpy(img).cuda()
    with rfa.Plan(shape, scans, clamped=True, path=path, prologue=prologue, epilogue=epilogue) as plan:
        out = plan.execute([dev])[0].cpu().numpy()
        names = [n for n, _ in plan.execute_timed([dev])[1]]
        if path == 0:
            assert plan.path_name == "tiled_fused"
            assert not any(n.startswith("pointwise") for n in names)       # fused into pass 1 / pass 2
        else:
            assert ("pointwise_pre" in names) == (prologue is not None)
            assert ("pointwise_post" in names) == (epilogue is not None)
    assert _pointwise_err(out, img, scans, True, prologue, epilogue) < TOL


def test_pointwise_f64_3d_and_planes():
    import torch
    import recfilter_amd as rfa
    cfg = rc.BASELINE_CONFIGS["cfg5_generic_xyz"]
    shape = (32, 32, 256)
    for dtype in (np.float32, np.float64):        # f32: fused x/y + strided z (epilogue after z); f64: generic
        imgs = [rc.random_image(shape, dtype, 3 + i) for i in range(2)]
        dev = [torch.from_numpy(im).cuda() for im in imgs]
        with rfa.Plan(shape, cfg["scans"], dtype=dtype, planes=2, prologue=(0.5, -0.25), epilogue=(1.0, -1.0, 0.0)) as plan:
            outs = [o.cpu().numpy() for o in plan.execute(dev)]
        for im, out in zip(imgs, outs):
            assert _pointwise_err(out, im, cfg["scans"], False, (0.5, -0.25), (1.0, -1.0, 0.0)) < TOL


def test_pointwise_misuse_is_rejected():
    import torch
    import recfilter_amd as rfa
    scans = [(0, True, [1.0, 1.0])]
    with pytest.raises(rfa.RecFilterError):                       # integer pixels
        rfa.Plan((64, 256), scans, dtype=np.int32, prologue=(2.0, 0.0))
    dev = torch.zeros((64, 256), device="cuda")
    with rfa.Plan((64, 256), scans, epilogue=(1.0, 1.0, 0.0)) as plan:
        with pytest.raises(rfa.RecFilterError):                   # epilogue reads the input: out must differ from in
            plan.execute([dev], [dev])
    with rfa.Plan((64, 256), scans, epilogue=(2.0, 0.0, 1.0)) as plan:
        out = plan.execute([dev], [dev])[0]                       # no input operand: in place is fine
        assert float(out.min()) == 1.0 and float(out.max()) == 1.0


def test_unsharp_mask_front_end():
    """apps/usm/unsharp_mask_optimized.cpp: USM = (1+w)*I - w*Blur(I), Blur computed at USM's tiles."""
    import torch
    import recfilter_amd as rfa
    w, h, weight = 512, 128, 1.0
    img = rc.random_image((h, w), np.float32, 11)
    x, y = rfa.RecFilterDim("x", w), rfa.RecFilterDim("y", h)
    W3 = rfa.gaussian_weights(5.0, 3)
    B = rfa.RecFilter("Blur")
    B.set_clamped_image_border()
    B[x, y] = torch.from_numpy(img).cuda()
    B.add_filter(+x, W3); B.add_filter(-x, W3); B.add_filter(+y, W3); B.add_filter(-y, W3)
    B.split_all_dimensions(32)
    B.compute_at(rfa.Pointwise(w_filtered=-weight, w_input=1.0 + weight))
    with pytest.raises(rfa.RecFilterUsageError):
        B.compute_at(rfa.Pointwise())                              # already has a consumer
    out = B.realize()[0].cpu().numpy()
    assert B.plan().path_name == "tiled_fused" and B.plan().num_kernels == 4      # (two tiles per row: no carry_x launch)
    # The mask is a difference of two O(1) terms, so the 1e-4 bar is taken relative to the terms it combines
    # (|(1+w) I| + |w Blur|), not to the possibly cancelled result.
    err = _pointwise_err(out, img, B._contents["scans"], True, None, (-weight, 1.0 + weight, 0.0))
    assert err < TOL, f"rel err {err}"


# ---- long 1-D signals on the fused path (rows chained through their entering states) ------------------------
@pytest.mark.parametrize("n,scans", [
    (8192, [(0, True, [1.0, 0.5])]),
    (8192 * 3, [(0, True, rc.GAUSS2), (0, False, rc.GAUSS2)]),
    (1 << 20, [(0, True, rc.GAUSS3), (0, False, rc.GAUSS3)]),
    (1 << 20, [(0, False, [0.3, 0.4, 0.2]), (0, True, [0.5, 0.5]), (0, True, [1.0, 0.25, -0.125]), (0, False, [0.9, 0.1])]),
    (10 << 20, [(0, True, [1.0, 0.1, 0.1])] * 4),        # apps/audio/audio_filter_biquads.cpp: cascaded biquads
], ids=["8k_o1", "24k_gauss2_pm", "1M_gauss3_pm", "1M_mixed4", "10M_biquads4"])
def test_long_1d_signal_chained_rows(n, scans):
    # (RF_PLAN_NO_OVERLAP: the scans as given on the fused kernels -- since round 5 the automatic plan merges the four biquads
    # into one scan of order 8 for the matrix path, below)
    imgs, outs, (path, tiles) = _run((n,), scans, planes=2 if n <= (1 << 20) else 1, flags=TILED | capi.RF_PLAN_NO_OVERLAP)
    assert path == 3 and tiles == (256,)
    _check(imgs, outs, scans, False)


def test_runs_of_same_direction_scans_of_a_signal_are_merged():
    """apps/audio/audio_filter_biquads.cpp: n biquads behind one another are ONE scan whose transfer function is the product of
    theirs (overlap_feedback_coeff, lib/iir_coeff.cpp:236-263); the automatic plan of a 1-D signal merges such runs where that
    leaves fewer stages and the merged direct form passes the conditioning probe -- fifteen biquads: one scan of order 30 on the
    matrix path, four launches.  Mixed directions merge run by run; a clamped border, an integer signal and RF_PLAN_NO_OVERLAP keep
    the scans as given."""
    import torch
    import recfilter_amd as rfa
    n = 1 << 20
    biquad = (0, True, [1.0, 0.1, 0.1])
    for count in (2, 5, 15):
        scans = [biquad] * count
        imgs, outs, (path, _) = _run((n,), scans)
        assert path == capi.RF_PATH_TILED_MATRIX
        _check(imgs, outs, scans, False)
        with rfa.Plan((n,), scans, flags=TILED) as plan:
            assert plan.num_kernels in (2, 4)          # (these poles decay to nothing across a tile: no carry chain at all)
            merged = plan.table("scans").reshape(-1, 5 + 2 * capi.RF_MAX_ORDER)
            assert merged.shape[0] == 1 and int(merged[0][2]) == 2 * count
    mixed = [(0, True, rc.GAUSS3), (0, True, rc.GAUSS2), (0, True, [0.7, 0.3]), (0, False, rc.GAUSS2), (0, False, rc.GAUSS2), (0, False, [0.6, 0.4])]
    imgs, outs, (path, _) = _run((n,), mixed)
    _check(imgs, outs, mixed, False)
    # (these recursive Gaussians have their poles at 0.8 .. 0.9: the merged direct forms of order 6 and 5 fail the conditioning probe
    # in f32 and the scans stay as given -- an in-plan cascade of two fused stages)
    with rfa.Plan((n,), mixed, flags=TILED) as plan:
        assert plan.path == 3
    mild = [(0, True, [0.5, 0.3, 0.1]), (0, True, [0.8, 0.2]), (0, True, [0.7, 0.2, -0.1]), (0, False, [0.6, 0.3]), (0, False, [0.9, 0.1, 0.05]), (0, False, [0.6, 0.4])]
    imgs, outs, (path, _) = _run((n,), mild)
    _check(imgs, outs, mild, False)
    with rfa.Plan((n,), mild, flags=TILED) as plan:
        # (round 6: merged run by run these would be orders 5 and 4, which the planner splits into sections again -- the poles
        #  rounded twice -- so the scans stay as given: an in-plan cascade of two fused stages, the table lists the first one)
        orders = [int(r[2]) for r in plan.table("scans").reshape(-1, 5 + 2 * capi.RF_MAX_ORDER)]
        assert orders == [2, 1, 2, 1] and plan.path == 3
    for kw in (dict(clamped=True), dict(dtype=np.int32), dict(flags=TILED | capi.RF_PLAN_NO_OVERLAP)):
        sc = [(0, True, [1.0, 1.0])] * 3 if kw.get("dtype") is not None else [biquad] * 3
        imgs, outs, (path, _) = _run((n,), sc, **kw)
        assert path == 3
        _check(imgs, outs, sc, kw.get("clamped", False))


def test_long_1d_signal_int32_prefix_sum_bit_exact_and_fallbacks():
    import recfilter_amd as rfa
    scans = [(0, True, [1.0, 1.0])]
    imgs, outs, (path, _) = _run((1 << 18,), scans, dtype=np.int32)
    assert path == 3
    np.testing.assert_array_equal(outs[0], np.cumsum(imgs[0].astype(np.int64)).astype(np.int32))
    # lengths that do not fold into rows run on zero-padded copies (test_1d_fused_any_length); signals shorter than one row
    # and clamped borders take the generic path
    imgs, outs, (path, _) = _run((8192 + 64,), scans)
    assert path == 3
    _check(imgs, outs, scans, False)
    imgs, outs, (path, _) = _run((8000,), scans)
    assert path == 2
    _check(imgs, outs, scans, False)
    # (a clamped 1-D signal: fused since round 3 -- the zero-border plan plus the border corrections, plan_clamp1d.h)
    imgs, outs, (path, _) = _run((8192,), [(0, True, rc.GAUSS2), (0, False, rc.GAUSS2)], clamped=True)
    assert path == 3
    _check(imgs, outs, [(0, True, rc.GAUSS2), (0, False, rc.GAUSS2)], True)


# ---- partial last tiles: any width that is a multiple of 16 stays on the fused path -----------------------------
def test_partial_tiles_other_features():
    import torch
    import recfilter_amd as rfa
    g2 = rc.xy_pm(rc.GAUSS2)
    # int32 bit-exact, in place, two planes
    imgs, outs, (path, tiles) = _run((64, 336), [(0, True, [1.0, 1.0]), (0, False, [1.0, 2.0]), (1, True, [1.0, 1.0])],
                                     dtype=np.int32, planes=2, inplace=True)
    assert path == 3
    _check(imgs, outs, [(0, True, [1.0, 1.0]), (0, False, [1.0, 2.0]), (1, True, [1.0, 1.0])], False)
    # 3-D (fused x/y per plane + strided z) and the 1080p-like 1920-wide frame
    cfg = rc.BASELINE_CONFIGS["cfg5_generic_xyz"]
    imgs, outs, (path, _) = _run((32, 64, 208), cfg["scans"])
    assert path == 3
    _check(imgs, outs, cfg["scans"], False)
    for shape in [(1088, 1920), (1080, 1920), (2160, 3840)]:       # 1080p / 4K frames: partial in x, in y, in both
        imgs, outs, (path, tiles) = _run(shape, g2, clamped=True)
        assert path == 3 and tiles[0] == 256 and tiles[1] in (32, 64)       # small frames take half-height tiles
        _check(imgs, outs, g2, True)
    # partial rows: int32 bit-exact with a clamped border, 3-D with partial x and y, two planes in place
    sc = [(1, True, [1.0, 1.0]), (1, False, [1.0, 2.0, 1.0]), (0, False, [1.0, 1.0])]
    imgs, outs, (path, _) = _run((75, 272), sc, dtype=np.int32, clamped=True, planes=2, inplace=True)
    assert path == 3
    _check(imgs, outs, sc, True)
    imgs, outs, (path, _) = _run((32, 50, 208), cfg["scans"])
    assert path == 3
    _check(imgs, outs, cfg["scans"], False)
    # pointwise epilogue on a partial width
    for shape, scans in [((64, 400), g2), ((45, 400), g2), ((45, 400), rc.xy_pm(rc.GAUSS3))]:    # order 3: re-read epilogue
        img = rc.random_image(shape, np.float32, 5)
        with rfa.Plan(shape, scans, clamped=True, prologue=(0.5, 0.125), epilogue=(-1.0, 2.0, 0.25)) as plan:
            assert plan.path_name == "tiled_fused"
            out = plan.execute([torch.from_numpy(img).cuda()])[0].cpu().numpy()
        assert _pointwise_err(out, img, scans, True, (0.5, 0.125), (-1.0, 2.0, 0.25)) < TOL


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_rows_with_partial_width(world):
    scans = rc.xy_pm(rc.GAUSS2)
    full, got, (path, nex) = _run_sharded((64 * world, 464), scans, True, world, path=0)
    assert path == 3 and nex == 1
    _check(full, got, scans, True)


@pytest.mark.parametrize("seed", range(24))
def test_sharded_random_filters_merged_exchange(seed):
    """Row / z-slab shards with random scans of orders 1-3 along every dimension (up to 4 per dimension, any mix of
    directions) and 2-5 slabs: one all-gather carries every scan of the sharded dimension, the cross-scan transfers
    Y / X (plan_generic.h) do the rest.  Every third case is an integer filter, checked bit-exact."""
    rng = np.random.default_rng(7000 + seed)
    world = int(rng.integers(2, 6))
    ndim = 2 if seed % 4 else 3
    integer = seed % 3 == 2
    path = 0 if seed % 2 == 0 else 2                              # fused / strided builders and the generic one
    if integer:
        scans = []
        for d in range(ndim):
            for _ in range(int(rng.integers(1, 4))):
                k = int(rng.integers(1, 4))
                scans.append((d, bool(rng.integers(0, 2)), [1.0] + [float(v) for v in rng.integers(-1, 2, size=k)]))
    else:
        scans = _random_filter(rng, ndim)
        if not any(d == ndim - 1 for d, _, _ in scans):
            scans.append((ndim - 1, True, [0.7, 0.4, -0.2]))
    clamped = bool(rng.integers(0, 2))
    if ndim == 2:
        shape = (32 * int(rng.integers(1, 4)) * world, 4 * int(rng.integers(8, 120)))
    else:
        shape = (32 * world, 32 * int(rng.integers(1, 3)), 64 * int(rng.integers(1, 4)))
    dtype = np.int32 if integer else np.float32
    full, got, (got_path, nex) = _run_sharded(shape, scans, clamped, world, path=path, dtype=dtype)
    n_outer = sum(1 for d, _, _ in scans if d == ndim - 1)
    assert nex == (1 if n_outer <= 4 else n_outer)
    _check(full, got, scans, clamped)


@pytest.mark.parametrize("seed", range(32))
def test_random_filters_random_shapes_partial_tiles(seed):
    """Arbitrary heights and widths that are multiples of 4: partial last tiles in x, y or both, scans that enter inside a
    16-sample segment; float filters against the oracle, every third case an integer filter bit-exact."""
    rng = np.random.default_rng(3000 + seed)
    shape = (int(rng.integers(1, 200)), 4 * int(rng.integers(1, 200)))
    clamped = bool(rng.integers(0, 2))
    if seed % 3 == 2:
        scans = []
        for d in range(2):
            for _ in range(int(rng.integers(1, 3))):
                k = int(rng.integers(1, 4))
                scans.append((d, bool(rng.integers(0, 2)), [float(rng.integers(1, 3))] + [float(rng.integers(-2, 3)) for _ in range(k)]))
        imgs, outs, (path, _) = _run(shape, scans, dtype=np.int32, clamped=clamped, seed=seed, planes=2)
    else:
        scans = _random_filter(rng, 2)
        imgs, outs, (path, _) = _run(shape, scans, clamped=clamped, seed=seed)
    assert path == 3, (shape, scans)
    _check(imgs, outs, scans, clamped)


# ---- rf_box_difference: the finite-difference consumer of the box-filter apps ------------------------------------
@pytest.mark.parametrize("shape,order,radius", [((64, 96), [1, 1], 5), ((40, 130), [2, 0], 3), ((70, 33), [0, 2], 5),
                                                ((50, 64), [2, 2], 1), ((12, 20, 24), [1, 2, 1], 2), ((300,), [2], 7),
                                                # the streaming kernel: several strips, strips shorter than the reach of
                                                # the taps, a radius beyond the image, a ring too large for LDS (gather)
                                                ((700, 300), [1, 1], 5), ((1000, 260), [0, 2], 7), ((520, 512), [2, 2], 3),
                                                ((20, 64), [1, 1], 30), ((90, 70), [1, 2], 14), ((300, 100), [1, 1], 40),
                                                ((3, 40, 300), [1, 1, 0], 4)])
def test_box_difference_matches_reference_expression(shape, order, radius):
    import torch
    import recfilter_amd as rfa
    import ref_loops
    for dtype in (np.float32, np.float64):
        tab = np.cumsum(rc.random_image(shape, dtype, 3), axis=-1).astype(dtype)
        out = rfa.box_difference(torch.from_numpy(tab).cuda(), radius, order).cpu().numpy()
        assert rc.rel_err(out, ref_loops.box_difference(tab, radius, order)) < (1e-4 if dtype == np.float32 else 1e-12)


def test_box_filter_app_is_a_box_filter():
    """apps/box/box_filter_1.cpp: summed-area table + differences = mean over the (2B+1)^2 window (interior pixels)."""
    import torch
    import recfilter_amd as rfa
    B, n = 5, 256
    img = rc.random_image((n, n), np.float64, 9)
    img[:B + 2], img[-B - 2:], img[:, :B + 2], img[:, -B - 2:] = 0, 0, 0, 0      # the apps pad the image with zeros
    x, y = rfa.RecFilterDim("x", n), rfa.RecFilterDim("y", n)
    F = rfa.RecFilter("Box1_Sat")
    F[x, y] = torch.from_numpy(img).cuda()
    F.add_filter(+x, [1.0, 1.0]); F.add_filter(+y, [1.0, 1.0])
    F.split(x, 32, y, 32)
    out = rfa.box_difference(F.realize()[0], B, [1, 1]).cpu().numpy()
    want = np.zeros_like(img)
    for dy in range(-B, B + 1):
        for dx in range(-B, B + 1):
            want += np.roll(np.roll(img, dy, axis=0), dx, axis=1)
    want /= (2 * B + 1) ** 2
    inner = (slice(2 * B + 2, n - 2 * B - 2),) * 2
    assert np.max(np.abs(out[inner] - want[inner])) < 1e-9
    with pytest.raises(rfa.RecFilterError):
        t = torch.zeros((8, 8), device="cuda")
        rfa.box_difference(t, 1, [1, 1], out=t)


def test_fused_path_rejects_misaligned_planes():
    """The fused kernels move 16 bytes per lane: an image pointer that is not 16-byte aligned is refused with an error
    code instead of being handed to the kernels (the generic path takes any element-aligned pointer)."""
    import torch
    import recfilter_amd as rfa
    n = 256
    buf = rc.cuda_image((n * n + 8,), np.float32, 907)
    shifted = buf[1:1 + n * n].view(n, n)                       # contiguous, 4 bytes off a 16-byte boundary
    aligned = buf[4:4 + n * n].view(n, n)
    scans = rc.xy_pm(rc.GAUSS2)
    with rfa.Plan((n, n), scans, clamped=True) as plan:
        assert plan.path_name == "tiled_fused"
        with pytest.raises(rfa.RecFilterError, match="aligned"):
            plan.execute([shifted])
        with pytest.raises(rfa.RecFilterError, match="aligned"):
            plan.execute([aligned], [torch.empty(n * n + 8, device="cuda")[1:1 + n * n].view(n, n)])
        out = plan.execute([aligned])[0].cpu().numpy()
    want = oracle.apply_filter(aligned.cpu().numpy().astype(np.float64), scans, True)
    assert rc.rel_err(out, want) < TOL
    with rfa.Plan((n, n), scans, clamped=True, path=2) as plan:          # the generic path has no such requirement
        out = plan.execute([shifted])[0].cpu().numpy()
    assert rc.rel_err(out, oracle.apply_filter(shifted.cpu().numpy().astype(np.float64), scans, True)) < TOL
    # the line-parallel untiled kernels (what the automatic path picks for an image this small, and path=1) move 16 bytes
    # per lane as well: same refusal, same message; aligned planes run
    for kw in (dict(flags=0), dict(path=1)):
        with rfa.Plan((n, n), scans, clamped=True, **kw) as plan:
            assert plan.path_name == "untiled"
            with pytest.raises(rfa.RecFilterError, match="aligned"):
                plan.execute([shifted])
            with pytest.raises(rfa.RecFilterError, match="aligned"):
                plan.execute([aligned], [shifted])
            out = plan.execute([aligned])[0].cpu().numpy()
        assert rc.rel_err(out, want) < TOL
    # an untiled filter the line kernels do not take (a width that is not a multiple of 16) runs one thread per line: any pointer
    odd = buf[1:1 + 250 * 200].view(200, 250)
    with rfa.Plan((200, 250), scans, clamped=True, path=1) as plan:
        out = plan.execute([odd])[0].cpu().numpy()
    assert rc.rel_err(out, oracle.apply_filter(odd.cpu().numpy().astype(np.float64), scans, True)) < TOL
    # a cascade whose stages are all line-kernel stages inherits the requirement
    five = [(0, True, rc.GAUSS2)] * 5
    with rfa.Plan((n, n), five, flags=0) as plan:
        with pytest.raises(rfa.RecFilterError, match="aligned"):
            plan.execute([shifted])


# ---- widths that are not multiples of 4: element-aligned rows, a partial chunk at every row's end ----------------------
@pytest.mark.parametrize("seed", range(16))
def test_fused_odd_widths_random(seed):
    """Any width on the fused kernels for 4- and 8-byte pixels (the reference only asks that the tile divide the extent,
    lib/recfilter.h:311): random filters, shapes, pixel types, planes; 64- and 128-row tiles; in place; a plane that is
    only element-aligned (rows of such an image are not 16-byte aligned anyway)."""
    import torch
    import recfilter_amd as rfa
    rng = np.random.default_rng(8800 + seed)
    shape = (int(rng.integers(1, 300)), int(rng.integers(1, 1100)) | 1 if seed % 2 else int(rng.integers(1, 1100)))
    if shape[1] % 4 == 0:
        shape = (shape[0], shape[1] + int(rng.integers(1, 4)))
    clamped = bool(rng.integers(0, 2))
    dtype = [np.float32, np.float32, np.int32, np.float64][seed % 4]
    if dtype == np.int32:
        scans = []
        for d in range(2):
            for _ in range(int(rng.integers(1, 3))):
                k = int(rng.integers(1, 4))
                scans.append((d, bool(rng.integers(0, 2)), [float(rng.integers(1, 3))] + [float(rng.integers(-2, 3)) for _ in range(k)]))
    else:
        scans = _random_filter(rng, 2)
    flags = TILED | (capi.RF_PLAN_TILE_ROWS(128) if seed % 3 == 0 and dtype == np.float32 else 0)
    planes = 1 + seed % 3
    imgs, outs, (path, tiles) = _run(shape, scans, dtype=dtype, clamped=clamped, planes=planes, seed=seed, flags=flags, inplace=bool(seed % 2))
    assert path == 3, (shape, dtype)
    _check(imgs, outs, scans, clamped)
    if dtype == np.float32:
        # an element-aligned plane (4 bytes off a 16-byte boundary), prologue + epilogue
        n = shape[0] * shape[1]
        buf = rc.cuda_image((n + 8,), np.float32, 9760 + seed)
        view = buf[1:1 + n].view(shape)
        pre, post = (0.5, 0.25), (-0.7, 1.7, 0.1)
        with rfa.Plan(shape, scans, clamped=clamped, prologue=pre, epilogue=post) as plan:
            assert plan.path == 3
            out = plan.execute([view])[0].cpu().numpy()
        assert _pointwise_err(out, view.cpu().numpy(), scans, clamped, pre, post) < TOL


def test_fused_odd_widths_other_features():
    """Odd widths with the rest of the fused path: a volume with a z stage, row shards, Tuple planes of order 3; int16 pixels
    and uint8 inputs keep the multiple-of-4 rule (they are moved in 8- and 4-byte pieces)."""
    import torch
    import recfilter_amd as rfa
    s3 = rc.BASELINE_CONFIGS["cfg5_generic_xyz"]["scans"]
    imgs, outs, (path, _) = _run((40, 50, 263), s3, clamped=False)
    assert path == 3
    _check(imgs, outs, s3, False)
    scans = rc.xy_pm(rc.GAUSS3)
    imgs, outs, (path, _) = _run((200, 1027), scans, clamped=True, planes=3)
    assert path == 3
    _check(imgs, outs, scans, True)
    full, got, (path, nex) = _run_sharded((64 + 128 + 64, 517), rc.xy_pm(rc.GAUSS2), True, 3, path=0, planes=2, extents=[64, 128, 64])
    assert path == 3 and nex == 1
    _check(full, got, rc.xy_pm(rc.GAUSS2), True)
    with rfa.Plan((64, 301), [(0, True, [1.0, 1.0]), (1, True, [1.0, 1.0])], dtype=np.int16) as plan:
        assert plan.path != 3
    with pytest.raises(rfa.RecFilterError):
        rfa.Plan((64, 301), rc.xy_pm(rc.GAUSS2), input_dtype=np.uint8, path=3)


# ---- Tuple planes batched into one launch per step (FusedArgs::plane_batch) ------------------------------------------
@pytest.mark.parametrize("planes,shape,dtype", [(3, (128, 512), np.float32), (5, (75, 464), np.float32),
                                                (4, (96, 300), np.int32), (16, (64, 256), np.float32)],
                         ids=["rgb", "five_partial", "int_partial", "max_planes"])
def test_batched_planes_match_the_per_plane_launches(planes, shape, dtype):
    """All planes of a 2-D filter ride in one launch per step, as the z planes of a volume whose planes are separate
    buffers: same results bit for bit as one launch per plane, the oracle's within tolerance, in place too."""
    import torch
    import recfilter_amd as rfa
    integer = np.issubdtype(dtype, np.integer)
    scans = [(0, True, [1.0, 1.0]), (0, False, [1.0, 1.0, -1.0]), (1, True, [1.0, 2.0, -1.0])] if integer else rc.xy_pm(rc.GAUSS2)
    rng = np.random.default_rng(77)
    imgs = [rng.integers(0, 5, size=shape).astype(dtype) if integer else rc.random_image(shape, np.float32, 90 + p)
            for p in range(planes)]
    dev = [torch.from_numpy(im).cuda() for im in imgs]
    with rfa.Plan(shape, scans, dtype=dtype, clamped=not integer, planes=planes) as plan:
        assert plan.path_name == "tiled_fused"
        outs, timed = plan.execute_timed(dev)
        assert [n for n, _ in timed] == ["fused_tails", "xscan_rows", "carry_y", "fused_pass2"]      # launches per step, not per plane (few tiles per row: no carry_x)
        batched = [o.cpu().numpy() for o in outs]
        inplace = [d.clone() for d in dev]
        plan.execute(inplace, inplace)
        for a, b in zip(batched, inplace):
            assert np.array_equal(a, b.cpu().numpy())
    with rfa.Plan(shape, scans, dtype=dtype, clamped=not integer, planes=planes,
                  flags=rfa.capi.RF_PLAN_TILED_ONLY | rfa.capi.RF_PLAN_NO_PLANE_BATCH) as plan:
        single = [o.cpu().numpy() for o in plan.execute(dev)]
    for a, b in zip(batched, single):
        assert np.array_equal(a, b)
    _check(imgs, batched, scans, not integer)


def test_one_plan_executes_concurrently_on_distinct_streams():
    """a plan is immutable after create, executes on distinct streams with distinct workspaces may overlap.
    Four images through ONE plan on four streams back to back (nothing waits in between), twice; then from four host
    threads at once.  Every result against the oracle; the plan built replicas for the overlapping executions and reuses
    an idle instance for a later execute on yet another stream."""
    import threading
    import torch
    import recfilter_amd as rfa
    scans = rc.xy_pm(rc.GAUSS2)
    shape = (1024, 2048)
    imgs = [rc.random_image(shape, np.float32, 700 + i) for i in range(4)]
    wants = [oracle.apply_filter(im.astype(np.float64), scans, True) for im in imgs]
    dev = [torch.from_numpy(im).cuda() for im in imgs]
    streams = [torch.cuda.Stream() for _ in range(4)]
    with rfa.Plan(shape, scans, clamped=True) as plan:
        assert plan.path == 3 and plan.num_instances == 1
        outs = [torch.empty_like(d) for d in dev]
        for st in streams:                                   # every stream is busy for a while: the executes below are
            with torch.cuda.stream(st):                      # all in flight at the same time, whatever the host's pace
                torch.cuda._sleep(50_000_000)
        for rep in range(2):
            for i in range(4):
                plan.execute([dev[i]], [outs[i]], stream=streams[i])
        torch.cuda.synchronize()
        n_over = plan.num_instances
        assert n_over == 4, n_over                           # (building a replica does not wait for the busy streams)
        for o, w in zip(outs, wants):
            assert rc.rel_err(o.cpu().numpy(), w) < TOL
        # everything has drained: an execute on a fifth stream takes an idle instance instead of building another
        extra = torch.cuda.Stream()
        out5 = plan.execute([dev[0]], stream=extra)[0]
        torch.cuda.synchronize()
        assert plan.num_instances == n_over and rc.rel_err(out5.cpu().numpy(), wants[0]) < TOL
        # four host threads, one stream each, three executes each
        outs2 = [torch.empty_like(d) for d in dev]
        errors = []

        def worker(i):
            try:
                for _ in range(3):
                    plan.execute([dev[i]], [outs2[i]], stream=streams[i])
            except Exception as exc:      # pragma: no cover
                errors.append(exc)
        threads = [threading.Thread(target=worker, args=(i,)) for i in range(4)]
        for th in threads:
            th.start()
        for th in threads:
            th.join()
        torch.cuda.synchronize()
        assert not errors, errors
        for o, w in zip(outs2, wants):
            assert rc.rel_err(o.cpu().numpy(), w) < TOL
        # same stream: one instance however many executes are queued
    with rfa.Plan(shape, scans, clamped=True) as plan:
        for i in range(8):
            plan.execute([dev[i % 4]], [outs[i % 4]])
        torch.cuda.synchronize()
        assert plan.num_instances == 1


def test_abandoned_stepping_execute_is_recovered():
    """a caller that fails between rf_plan_begin and rf_plan_finish (a collective that raised) must not leave
    the plan unusable.  begin() again on the same thread restarts; rf_plan_abort hands the instance back; another thread is
    not blocked by the abandoned execute; a plan may be destroyed while an execute is begun; ShardedFilter aborts when its
    collective raises."""
    import threading
    import torch
    import recfilter_amd as rfa
    from recfilter_amd import capi
    from recfilter_amd.dist import ShardedFilter
    scans = rc.xy_pm(rc.GAUSS2)
    shape = (512, 1024)
    img = rc.random_image(shape, np.float32, 811)
    want = oracle.apply_filter(img.astype(np.float64), scans, True)
    dev = torch.from_numpy(img).cuda()
    out = torch.empty_like(dev)
    flags = capi.RF_PLAN_FORCE_EXCHANGE | capi.RF_PLAN_TILED_ONLY

    def whole(plan, o):
        plan.begin([dev], [o])
        for i in range(plan.num_exchanges):
            send = torch.empty(plan.exchange_bytes(i), dtype=torch.uint8, device="cuda")
            plan.exchange_local(i, send.data_ptr())
            plan.exchange_apply(i, send.data_ptr())           # one rank: gathered == send
        plan.finish()
        torch.cuda.synchronize()
        assert rc.rel_err(o.cpu().numpy(), want) < TOL

    with rfa.Plan(shape, scans, clamped=True, flags=flags) as plan:
        plan.begin([dev], [out])                              # ... and the caller "fails" here
        whole(plan, out)                                      # begin on the same thread starts afresh
        plan.begin([dev], [out])
        plan.abort()
        plan.abort()                                          # a no-op without a begun execute
        with pytest.raises(RuntimeError):
            plan.finish()                                     # nothing to finish after the abort
        whole(plan, out)
        # an execute begun and abandoned on THIS thread does not block another thread on the same (null) stream
        plan.begin([dev], [out])
        plan.abort()
        errors = []

        def other():
            try:
                whole(plan, torch.empty_like(dev))
            except Exception as exc:          # pragma: no cover
                errors.append(exc)
        th = threading.Thread(target=other)
        th.start()
        th.join(timeout=120)
        assert not th.is_alive() and not errors, errors
        assert plan.num_instances == 1
        plan.begin([dev], [out])                              # destroyed while begun: nothing dangles (the `with` closes it)
    # a new plan -- possibly at the same address -- starts clean
    with rfa.Plan(shape, scans, clamped=True, flags=flags) as plan:
        with pytest.raises(RuntimeError):
            plan.finish()
        whole(plan, out)

    # the Python driver: a collective that raises leaves the plan ready for the next execute
    calls = [0]

    def flaky(gathered, send):
        calls[0] += 1
        if calls[0] == 1:
            raise RuntimeError("collective failed")
        gathered.copy_(send)
        return None
    filt = ShardedFilter(shape, scans, clamped=True, rank=0, world=1, force_exchange=True, flags=capi.RF_PLAN_TILED_ONLY,
                         collective=flaky)
    with pytest.raises(RuntimeError, match="collective failed"):
        filt.execute([dev], [out])
    out.zero_()
    filt.execute([dev], [out])
    torch.cuda.synchronize()
    assert rc.rel_err(out.cpu().numpy(), want) < TOL


def test_instance_pool_under_contention():
    """an instance is judged idle only while nobody owns it.  Eight host threads hammer ONE plan, two threads per
    stream (so that same-stream executes contend for one instance while other streams look for idle ones); every output is
    checked after every execute against the oracle, which a shared workspace between two executions in flight would break."""
    import threading
    import torch
    import recfilter_amd as rfa
    scans = rc.xy_pm(rc.GAUSS2)
    shape = (512, 1024)
    imgs = [rc.random_image(shape, np.float32, 820 + i) for i in range(8)]
    wants = [oracle.apply_filter(im.astype(np.float64), scans, True) for im in imgs]
    dev = [torch.from_numpy(im).cuda() for im in imgs]
    streams = [torch.cuda.Stream() for _ in range(4)]
    errors = []
    with rfa.Plan(shape, scans, clamped=True) as plan:
        def worker(i):
            try:
                st = streams[i % 4]
                out = torch.empty_like(dev[i])
                for rep in range(12):
                    plan.execute([dev[i]], [out], stream=st)
                    if rep % 4 == 3:
                        st.synchronize()
                        err = rc.rel_err(out.cpu().numpy(), wants[i])
                        assert err < TOL, (i, rep, err)
            except BaseException as exc:      # pragma: no cover
                errors.append(exc)
        threads = [threading.Thread(target=worker, args=(i,)) for i in range(8)]
        for th in threads:
            th.start()
        for th in threads:
            th.join()
        torch.cuda.synchronize()
        assert not errors, errors
        assert 1 <= plan.num_instances <= 8


def test_plans_release_their_device_memory():
    """rf_plan_destroy frees everything a plan allocated -- workspace, tables, the replicas concurrent executes made it build,
    the stages of a cascade, the helper plan of an early exchange: free device memory returns to where it was."""
    import torch
    import recfilter_amd as rfa
    torch.cuda.synchronize()
    x = rc.cuda_image((1024, 2048), np.float32, 1222)
    vol = rc.cuda_image((64, 64, 256), np.float32, 1223)
    streams = [torch.cuda.Stream() for _ in range(3)]

    def cycle():
        with rfa.Plan((1024, 2048), rc.xy_pm(rc.GAUSS2), clamped=True) as p:
            for st in streams:
                with torch.cuda.stream(st):
                    torch.cuda._sleep(20_000_000)          # (keeps the stream busy: the executes overlap for certain)
                p.execute([x], stream=st)
            torch.cuda.synchronize()
            assert p.num_instances >= 2
        with rfa.Plan((1024, 2048), [(0, True, [0.5, 0.5])] * 6 + [(1, False, [0.6, 0.4])]) as p:      # in-plan cascade
            p.execute([x])
        with rfa.Plan((64, 64, 256), rc.BASELINE_CONFIGS["cfg5_generic_xyz"]["scans"], shard_rank=0, shard_world=2) as p:   # helper plan
            assert p.has_interior
        with rfa.Plan((1024, 2048), rc.xy_pm(rc.GAUSS2), clamped=True, path=4, tile=[32, 32]) as p:
            p.execute([x])
        torch.cuda.synchronize()
    cycle()                                   # (first use: code objects, torch's allocator pools)
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for _ in range(5):
        cycle()
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < (8 << 20), f"device memory leaked: {(free0 - free1) / 2 ** 20:.1f} MiB over 5 cycles"
    del vol


# ---- unsigned-byte input planes converted on load (rf_pointwise_desc.in_dtype = RF_IN_U8) -------------------------
@pytest.mark.parametrize("shape,path", [((128, 512), 0), ((75, 464), 0), ((64, 250), 0), ((64, 256), 1), ((40, 16, 272), 0)],
                         ids=["fused", "fused_partial", "generic_auto", "untiled", "fused_3d"])
@pytest.mark.parametrize("post", [None, (-1.0, 2.0, 0.1)], ids=["plain", "unsharp"])
def test_uint8_input_planes(shape, path, post):
    import torch
    import recfilter_amd as rfa
    scans = rc.xy_pm(rc.GAUSS2) if len(shape) == 2 else rc.BASELINE_CONFIGS["cfg5_generic_xyz"]["scans"]
    clamped = len(shape) == 2
    rng = np.random.default_rng(5)
    imgs = [rng.integers(0, 256, size=shape, dtype=np.uint8) for _ in range(2)]
    dev = [torch.from_numpy(im).cuda() for im in imgs]
    with rfa.Plan(shape, scans, clamped=clamped, planes=2, path=path, input_dtype=np.uint8, prologue=(1.0 / 255.0, 0.0),
                  epilogue=post) as plan:
        outs = [o.cpu().numpy() for o in plan.execute(dev)]
        assert outs[0].dtype == np.float32
        with pytest.raises(TypeError):
            plan.execute([d.float() for d in dev])                    # the plan wants byte planes
    for im, out in zip(imgs, outs):
        x = np.float32(1.0 / 255.0) * im.astype(np.float32)            # the conversion the kernels apply on load
        want, scale = rc.pointwise_want(x, scans, clamped, None, post)
        assert rc.rel_err(out, want, scale=scale) < TOL


def test_uint8_input_without_scale_and_order3_epilogue():
    import torch
    import recfilter_amd as rfa
    scans = rc.xy_pm(rc.GAUSS3)
    im = np.random.default_rng(6).integers(0, 256, size=(70, 400), dtype=np.uint8)
    with rfa.Plan((70, 400), scans, clamped=True, input_dtype=np.uint8, epilogue=(1.0, -1.0, 0.0)) as plan:
        assert plan.path_name == "tiled_fused"
        out = plan.execute([torch.from_numpy(im).cuda()])[0].cpu().numpy()
    # a difference of two O(255) terms: the bar is relative to the terms (see test_unsharp_mask_front_end)
    want, scale = rc.pointwise_want(im.astype(np.float32), scans, True, None, (1.0, -1.0, 0.0))
    assert rc.rel_err(out, want, scale=scale) < TOL


def test_steps_in_flight_keep_their_images_apart():
    """ShardedFilter(inflight=3).submit on one GPU: seven different images through three slots (own stream, plan and
    workspace each), every output against the oracle -- a slot must never see another slot's tails."""
    import torch
    from recfilter_amd.dist import ShardedFilter
    scans = rc.xy_pm(rc.GAUSS2)
    shape = (320, 1024)
    filt = ShardedFilter(shape, scans, clamped=True, inflight=3)
    assert filt.plan.path_name == "tiled_fused"
    imgs = [rc.random_image(shape, np.float32, 300 + i) for i in range(7)]
    dev = [torch.from_numpy(im).cuda() for im in imgs]
    outs = [torch.empty_like(d) for d in dev]
    for d, o in zip(dev, outs):
        filt.submit([d], [o])
    filt.drain()
    torch.cuda.synchronize()
    for im, o in zip(imgs, outs):
        want = oracle.apply_filter(im.astype(np.float64), scans, True)
        assert rc.rel_err(o.cpu().numpy(), want) < 1e-4


# ---- line-parallel untiled kernels (kernels_lines.hip): RF_PATH_UNTILED, and what RF_PATH_AUTO picks for small images ----
@pytest.mark.parametrize("dtype", [np.float32, np.float64, np.int32, np.int16])
@pytest.mark.parametrize("shape,clamped", [((512, 512), True), ((64, 64), False), ((192, 320), True), ((1024, 768), False),
                                            ((16, 4096), True), ((48, 32, 80), True), ((2048,), False),
                                            ((1296, 1808), True), ((2048, 1040), False)])      # lines of 1025..2048 samples: eight tiles in registers
def test_line_parallel_untiled_kernels(dtype, shape, clamped):
    """Every scan walks the whole line (no tiling algebra at all): the x phase's 16-lane segment scan, tile after tile,
    the state handed on in registers.  Widths / heights that are multiples of 16 only (partial last tiles), 1-D, 3-D."""
    import torch
    import recfilter_amd as rfa
    nd = len(shape)
    if np.issubdtype(dtype, np.integer):
        scans = [(0, True, [1.0, 2.0, -1.0]), (0, False, [1.0, 1.0])] + ([(1, True, [1.0, 1.0]), (1, False, [2.0, 1.0, -1.0, 1.0])] if nd > 1 else []) \
            + ([(2, False, [1.0, 1.0])] if nd > 2 else [])
    else:
        scans = [s for s in rc.xy_pm(rc.GAUSS3) if s[0] < nd] + ([(2, True, rc.GAUSS2), (2, False, [0.5, 0.5])] if nd > 2 else [])
    imgs, outs, _ = _run(shape, scans, dtype, clamped, planes=2, path=1)
    _check(imgs, outs, scans, clamped)
    dev = torch.from_numpy(imgs[0]).cuda()
    with rfa.Plan(shape, scans, dtype=dtype, clamped=clamped, path=1) as plan:
        _, times = plan.execute_timed([dev])
        assert [n for n, _ in times] == ["line_scans_" + "xyz"[d] for d in range(nd)]      # one launch per dimension


def test_auto_path_sends_small_images_to_the_line_kernels(shipped_defaults):
    import torch
    import recfilter_amd as rfa
    scans = rc.xy_pm(rc.GAUSS2)
    for n, want in ((256, 1), (1024, 1), (1536, 3), (2048, 3)):
        with rfa.Plan((n, n), scans, clamped=True) as plan:
            assert plan.path == want, (n, plan.path_name)
    for n, want in ((1024, 1), (1536, 1), (1984, 1), (2048, 3)):        # order 3, four scans: the line kernels up to 1984
        with rfa.Plan((n, n), rc.xy_pm(rc.GAUSS3), clamped=True) as plan:
            assert plan.path == want, (n, plan.path_name)
    imgs, outs, (path, _) = _run((512, 512), scans, clamped=True)
    assert path == 1
    _check(imgs, outs, scans, True)
    # f64 pixels: the fused kernels (256 x 32 tiles) above the small-image limit, the line kernels below
    imgs, outs, (path, _) = _run((2048, 1536), scans, np.float64, True)
    assert path == 3
    _check(imgs, outs, scans, True)
    imgs, outs, (path, _) = _run((512, 384), scans, np.float64, True)
    assert path == 1
    _check(imgs, outs, scans, True)
    # split() widths are hints where the fused kernels apply (the tile size never changes the result): still the line
    # kernels; pointwise stages stay on the tiled passes they are fused into
    with rfa.Plan((512, 512), scans, clamped=True, tile=[32, 32]) as plan:
        assert plan.path == 1
    with rfa.Plan((512, 512), scans, clamped=True, prologue=(0.5, 0.0)) as plan:
        assert plan.path == 3
    with rfa.Plan((512, 512), scans, dtype=np.float64, clamped=True, tile=[32, 32]) as plan:
        assert plan.path == 1


def test_tap_filter_against_numpy():
    """rf_tap_filter (the clamped difference Funcs of apps/DoG/diff_gauss.cpp:176-197) against its numpy restatement:
    2-D with two input planes, 3-D, 1-D, f32 and f64, widths that are not multiples of 4; misuse is rejected."""
    import torch
    import recfilter_amd as rfa
    import ref_loops
    rng = np.random.default_rng(5)
    for shape, dt in (((37, 50), np.float32), ((64, 131), np.float64), ((5, 9, 22), np.float32), ((301,), np.float32)):
        nd = len(shape)
        planes = [rng.random(shape).astype(dt) for _ in range(2)]
        taps = [(int(rng.integers(0, 2)), [int(v) for v in rng.integers(-9, 10, size=nd)], float(np.float32(rng.uniform(-1, 1)))) for _ in range(7)]      # (weights travel as f32)
        got = rfa.tap_filter([torch.from_numpy(p).cuda() for p in planes], taps).cpu().numpy()
        want = ref_loops.tap_filter(planes, taps)
        assert np.abs(got - want).max() < (1e-5 if dt == np.float32 else 1e-12)
    B1, B2 = 3, 5
    t = ref_loops.dog_taps(B1, B2)
    assert len(t["dog"]) == 6 and len(t["box1"][0]) == 4
    a = rc.cuda_image((16, 16), np.float32, 1385)
    with pytest.raises(rfa.RecFilterError):
        rfa.tap_filter([a], [(1, (0, 0), 1.0)])                    # plane out of range
    with pytest.raises(rfa.RecFilterError):
        rfa.tap_filter([a], [(0, (0, 0), 1.0)], out=a)             # the operator gathers
    with pytest.raises(rfa.RecFilterError):
        rfa.tap_filter([a], [(0, (0, 0), 1.0)] * 17)               # more than RF_MAX_TAPS


# ---- 256 x 128 tiles (kernels_fused_tall.hip): what RF_PATH_AUTO picks for large images of order >= 2, forced here on small ones ----
@pytest.mark.parametrize("seed", range(10))
def test_tall_tiles_random_filters_and_shapes(seed, plan_flags):
    """The 128-row final pass (two 64-row halves through the LDS, the column in registers), the 128-row tail extraction and
    the two-block residual: random filters (every y scan pattern, orders 1..3), heights and widths with partial last tiles
    (a last tile row of fewer than 64 rows leaves the second half empty), float against the oracle, integers bit-exact."""
    plan_flags(TILED | capi.RF_PLAN_TILE_ROWS(128))
    rng = np.random.default_rng(5100 + seed)
    shape = (int(rng.integers(1, 420)), 4 * int(rng.integers(1, 200)))
    clamped = bool(rng.integers(0, 2))
    if seed % 3 == 2:
        scans = []
        for d in range(2):
            for _ in range(int(rng.integers(1, 3))):
                k = int(rng.integers(1, 4))
                scans.append((d, bool(rng.integers(0, 2)), [float(rng.integers(1, 3))] + [float(rng.integers(-2, 3)) for _ in range(k)]))
        imgs, outs, (path, tiles) = _run(shape, scans, dtype=np.int32, clamped=clamped, seed=seed, planes=2)
    else:
        scans = _random_filter(rng, 2)
        if seed % 3 == 0:
            scans = rc.xy_pm(rc.GAUSS3 if seed % 2 else rc.GAUSS2)            # the fixed causal / anticausal pattern
        imgs, outs, (path, tiles) = _run(shape, scans, clamped=clamped, seed=seed)
    assert path == 3 and (list(tiles)[:2] == [256, 128] or not any(s[0] == 1 for s in scans) or not any(s[0] == 0 for s in scans)), (shape, tiles)
    _check(imgs, outs, scans, clamped)


def test_tall_tiles_other_features(plan_flags):
    """128-row tiles with the rest of the fused path: row shards (the entering carries applied by the final pass, slabs of
    different heights), a 3-D volume with a z stage behind, batched Tuple planes, uint8 input and a pointwise epilogue,
    int16 pixels; and what the automatic choice is."""
    import torch
    import recfilter_amd as rfa
    plan_flags(TILED | capi.RF_PLAN_TILE_ROWS(128))
    scans = rc.xy_pm(rc.GAUSS2)
    full, got, (path, nex) = _run_sharded((128 + 384 + 256, 512), scans, True, 3, path=0, planes=2, extents=[128, 384, 256])
    assert path == 3 and nex == 1
    _check(full, got, scans, True)
    s3 = rc.BASELINE_CONFIGS["cfg5_generic_xyz"]["scans"]
    imgs, outs, (path, tiles) = _run((64, 256, 512), s3, clamped=False)
    assert path == 3 and list(tiles)[:2] == [256, 128]
    _check(imgs, outs, s3, False)
    imgs, outs, (path, tiles) = _run((384, 512), rc.xy_pm(rc.GAUSS3), clamped=True, planes=3)
    assert path == 3 and list(tiles)[:2] == [256, 128]
    _check(imgs, outs, rc.xy_pm(rc.GAUSS3), True)
    ints = [(0, True, [1.0, 2.0, -1.0]), (0, False, [1.0, 1.0]), (1, True, [1.0, 2.0, -1.0]), (1, False, [1.0, 1.0])]
    imgs, outs, (path, tiles) = _run((200, 520), ints, dtype=np.int16, clamped=False)
    assert path == 3 and list(tiles)[:2] == [256, 128]
    _check(imgs, outs, ints, False)
    # uint8 input, prologue and unsharp-mask epilogue
    img8 = rc.cuda_image((300, 768), np.uint8, 1443)
    w = 0.7
    with rfa.Plan((300, 768), scans, clamped=True, prologue=(1.0 / 255.0, 0.0), epilogue=(-w, 1.0 + w, 0.0), input_dtype=np.uint8) as plan:
        assert list(plan.tiles)[:2] == [256, 128]
        out = plan.execute([img8])[0].cpu().numpy()
    x = img8.cpu().numpy().astype(np.float64) / 255.0
    want = (1.0 + w) * x - w * oracle.apply_filter(x, scans, True)
    assert np.abs(out - want).max() < 2e-5
    plan_flags(TILED)
    for shape, sc, want_ty in (((16384, 8192), scans, 128), ((16384, 8192), rc.xy_pm([1.3, -0.3]), 128), ((2048, 2048), scans, 32),
                               ((8192, 8192), scans, 64), ((16384 + 64, 8192), scans, 128), ((16380, 8188), scans, 128)):      # (partial last tile row / column: strips of their own)
        with rfa.Plan(shape, sc, clamped=True) as plan:
            assert list(plan.tiles)[:2] == [256, want_ty], (shape, plan.tiles)


# ---- orders above 3: the plan splits a scan into first/second/third-order sections (sections.h) and stays on the fused kernels ----
def _from_poles(poles, b=0.3):
    p = np.poly(poles).real
    return [b] + [float(-v) for v in p[1:]]


@pytest.mark.parametrize("name,poles", [
    ("order4", [0.7, 0.6, 0.3 + 0.5j, 0.3 - 0.5j]),
    ("order5", [0.8, 0.5 + 0.3j, 0.5 - 0.3j, -0.2 + 0.6j, -0.2 - 0.6j]),
    ("order6", [0.85, 0.1, 0.4 + 0.4j, 0.4 - 0.4j, -0.5 + 0.2j, -0.5 - 0.2j]),
])
def test_high_order_scans_run_as_sections_on_the_fused_path(name, poles):
    """lib/split.cpp:575-578 takes any order; here a zero-border float filter of order 4..6 is rewritten into sections of
    order <= 3 with the same transfer function and runs on the fused kernels.  Checked against the
    oracle run on the ORIGINAL high-order coefficients."""
    import recfilter_amd as rfa
    co = _from_poles(poles)
    scans = [(0, True, co), (0, False, co), (1, True, co), (1, False, co)]
    imgs, outs, (path, tiles) = _run((328, 1024), scans, clamped=False)
    assert path == 3, (name, path)
    _check(imgs, outs, scans, False)
    # a clamped border: the sections run in zero-border form behind border modifications (next test); shapes that form does
    # not take (a width that is not a multiple of 16) keep the scans as given on another path
    imgs, outs, (path, _) = _run((128, 256), scans, clamped=True)
    assert path == 3
    _check(imgs, outs, scans, True)
    imgs, outs, (path, _) = _run((128, 260), scans, clamped=True)
    assert path != 3
    _check(imgs, outs, scans, True)


@pytest.mark.parametrize("name,poles", [
    ("order4", [0.7, 0.6, 0.3 + 0.5j, 0.3 - 0.5j]),
    ("order5", [0.8, 0.5 + 0.3j, 0.5 - 0.3j, -0.2 + 0.6j, -0.2 - 0.6j]),
    ("order6", [0.85, 0.1, 0.4 + 0.4j, 0.4 - 0.4j, -0.5 + 0.2j, -0.5 - 0.2j]),
    ("order8", [0.8, -0.7, 0.5, -0.3, 0.3 + 0.6j, 0.3 - 0.6j, -0.1 + 0.7j, -0.1 - 0.7j]),
])
def test_clamped_high_order_scans_as_sections_behind_border_modifications(name, poles):
    """orders 4..8 with a CLAMPED border on the fused kernels.  A clamped scan is the zero-border scan of
    an input whose first k samples in scan direction are x_r + g_r x_0 (lib/recfilter.cpp:330-336 rearranged; plan.cpp), and the
    zero-border scan factors into sections: pass 1 and the carries see the modification through their tables, the kernels that
    run recurrences (both final passes, xscan_rows, the strided z pass) apply it on the tile where a scan enters the image.
    Against the oracle run on the ORIGINAL coefficients with its clamped border: every tile height, one tile / partial last
    tile column / many tiles, two scans per dimension and one, mixed with low-order scans, planes, a volume."""
    import recfilter_amd as rfa
    from recfilter_amd import capi
    co = _from_poles(poles, b=0.25)
    pm = [(0, True, co), (0, False, co), (1, True, co), (1, False, co)]
    shapes = [(128, 256), (320, 1024), (352, 1280), (64, 16), (96, 48)] if len(poles) <= 6 else [(128, 256), (352, 1280)]
    for shape in shapes:
        scans = pm if len(poles) <= 6 else [(0, True, co), (1, False, co)]      # (order 8: three sections per scan, four scans per dimension at most)
        imgs, outs, (path, tiles) = _run(shape, scans, clamped=True)
        assert path == 3, (name, shape, path)
        _check(imgs, outs, scans, True)
    if len(poles) > 6:
        return
    for ty in (32, 64, 128):
        imgs, outs, (path, tiles) = _run((256, 768), pm, clamped=True, flags=capi.RF_PLAN_TILED_ONLY | capi.RF_PLAN_TILE_ROWS(ty))
        assert path == 3 and list(tiles)[1] == ty
        _check(imgs, outs, pm, True)
    # high order along x only, the usual order-2 pair along y (every scan of the plan is put into the same form); anticausal first
    mixed = [(0, False, co), (0, True, co), (1, True, rc.GAUSS2), (1, False, rc.GAUSS2)]
    imgs, outs, (path, _) = _run((160, 512), mixed, clamped=True, planes=3)
    assert path == 3
    _check(imgs, outs, mixed, True)
    # a volume: high order along z (the strided kernels), order 3 along x and y
    vol = [(0, True, rc.GAUSS3), (0, False, rc.GAUSS3), (1, True, rc.GAUSS3), (2, True, co), (2, False, co)]
    imgs, outs, (path, _) = _run((64, 96, 256), vol, clamped=True)
    assert path == 3
    _check(imgs, outs, vol, True)


def test_clamped_sections_full_size_and_fallbacks():
    """An order-5 clamped x/y filter at 4096^2 on path 3 (the 128-row final pass is forced on small shapes in the test above);
    f64 pixels and heights that are not multiples of 32 keep the scans as given (another path), same result."""
    import recfilter_amd as rfa
    co = _from_poles([0.8, 0.5 + 0.3j, 0.5 - 0.3j, -0.2 + 0.6j, -0.2 - 0.6j], b=0.25)
    pm = [(0, True, co), (0, False, co), (1, True, co), (1, False, co)]
    imgs, outs, (path, tiles) = _run((4096, 4096), pm, clamped=True)
    assert path == 3, (path, tiles)
    _check(imgs, outs, pm, True)
    imgs, outs, (path, _) = _run((72, 256), pm, clamped=True)                   # 72 rows: not a multiple of 32
    assert path != 3
    _check(imgs, outs, pm, True)
    # pointwise stages: an affine prologue and an affine epilogue ride along; an epilogue with an input operand keeps the scans
    # as given (another path)
    import torch
    img = rc.random_image((128, 512), np.float32, 77)
    dev = torch.from_numpy(img).cuda()
    with rfa.Plan((128, 512), pm, clamped=True, prologue=(0.5, 0.25), epilogue=(2.0, 0.0, -1.0)) as plan:
        assert plan.path == 3
        got = plan.execute([dev])[0].cpu().numpy()
    assert _pointwise_err(got, img, pm, True, (0.5, 0.25), (2.0, 0.0, -1.0)) < TOL
    with rfa.Plan((128, 512), pm, clamped=True, epilogue=(-0.7, 1.7, 0.0)) as plan:
        assert plan.path != 3
        got = plan.execute([dev])[0].cpu().numpy()
    want = 1.7 * img.astype(np.float64) - 0.7 * oracle.apply_filter(img.astype(np.float64), pm, True)
    assert np.abs(got - want).max() < 2e-5
    imgs, outs, (path, _) = _run((128, 256), pm, dtype=np.float64, clamped=True)
    assert path != 3
    for im, o in zip(imgs, outs):
        assert rc.rel_err(o, oracle.apply_filter(im, pm, True)) < 1e-9


def test_high_order_sections_other_cases():
    """One causal scan of order 8 per dimension (three sections each), a 1-D signal of order 9 on the chained-rows path,
    three conjugate pairs twice per dimension (six sections > four scans: an in-plan cascade of two stages), and a filter
    the rewrite must leave alone: integer pixels."""
    import recfilter_amd as rfa
    o8 = _from_poles([0.8, -0.7, 0.5, -0.3, 0.3 + 0.6j, 0.3 - 0.6j, -0.1 + 0.7j, -0.1 - 0.7j], b=0.2)
    scans = [(0, True, o8), (1, True, o8)]
    imgs, outs, (path, _) = _run((256, 512), scans, clamped=False)
    assert path == 3
    _check(imgs, outs, scans, False)
    # ONE scan of a 1-D signal takes its direct form on the matrix path since round 5 (four launches whatever the order);
    # asked for by name, the fused kernels still run its sections on the chained-rows path
    sig, out1, (path, _) = _run((8192 * 4,), [(0, True, o8)], clamped=False)
    assert path == capi.RF_PATH_TILED_MATRIX
    _check(sig, out1, [(0, True, o8)], False)
    sig, out1, (path, _) = _run((8192 * 4,), [(0, True, o8)], clamped=False, path=3)
    assert path == 3
    _check(sig, out1, [(0, True, o8)], False)
    o6c = _from_poles([0.6 + 0.2j, 0.6 - 0.2j, 0.4 + 0.4j, 0.4 - 0.4j, -0.5 + 0.2j, -0.5 - 0.2j])
    scans = [(0, True, o6c), (0, False, o6c), (1, True, o6c)]
    imgs, outs, (path, _) = _run((128, 256), scans, clamped=False)
    assert path == 3                                                               # (two stages of an in-plan cascade)
    _check(imgs, outs, scans, False)
    ints = [(0, True, [1.0, 1.0, 0.0, 0.0, 1.0]), (1, True, [1.0, 1.0])]
    imgs, outs, (path, _) = _run((64, 256), ints, dtype=np.int32, clamped=False)
    assert path != 3
    _check(imgs, outs, ints, False)


# ---- f64 pixels on the fused kernels (256 x 32 tiles: the LDS footprint of a 256 x 64 tile of f32) ----
@pytest.mark.parametrize("seed", range(8))
def test_f64_on_the_fused_path(seed):
    """Random filters (orders 1..3, every scan pattern), random heights and widths with partial last tiles, both borders:
    f64 pixels through pass 1 / carries / residual / pass 2, against the f64 oracle at 1e-10."""
    rng = np.random.default_rng(7100 + seed)
    shape = (int(rng.integers(1, 300)), 4 * int(rng.integers(1, 260)))
    clamped = bool(rng.integers(0, 2))
    scans = rc.xy_pm(rc.GAUSS3 if seed % 2 else rc.GAUSS2) if seed % 3 == 0 else _random_filter(rng, 2)
    imgs, outs, (path, tiles) = _run(shape, scans, dtype=np.float64, clamped=clamped, seed=seed, path=3)
    assert path == 3 and (list(tiles)[1] == 32 or not any(s[0] == 1 for s in scans))
    for im, o in zip(imgs, outs):
        want = oracle.apply_filter(im.astype(np.float64), scans, clamped)
        assert rc.rel_err_strict(o, want) < 1e-9 or rc.rel_err(o, want) < 1e-10


def test_f64_fused_other_features():
    """f64 on the fused path with a z stage behind (generic kernels for z), Tuple planes, row shards with slabs of different
    heights, a pointwise epilogue; and the automatic choice for f64 (fused above the small-image limit)."""
    import torch
    import recfilter_amd as rfa
    s3 = rc.BASELINE_CONFIGS["cfg5_generic_xyz"]["scans"]
    imgs, outs, (path, tiles) = _run((24, 96, 256), s3, dtype=np.float64, clamped=False, path=3)
    assert path == 3
    for im, o in zip(imgs, outs):
        assert rc.rel_err(o, oracle.apply_filter(im.astype(np.float64), s3, False)) < 1e-10
    scans = rc.xy_pm(rc.GAUSS2)
    imgs, outs, (path, _) = _run((160, 512), scans, dtype=np.float64, clamped=True, planes=3, path=3)
    assert path == 3
    for im, o in zip(imgs, outs):
        assert rc.rel_err(o, oracle.apply_filter(im.astype(np.float64), scans, True)) < 1e-10
    full, got, (path, nex) = _run_sharded((64 + 160 + 96, 512), scans, True, 3, path=3, dtype=np.float64, extents=[64, 160, 96])
    assert path == 3 and nex == 1
    for im, o in zip(full, got):
        assert rc.rel_err(o, oracle.apply_filter(im.astype(np.float64), scans, True)) < 1e-10
    img = rc.cuda_image((200, 768), np.float64, 1627)
    w = 0.5                                        # (the weights travel as f32: exactly representable ones)
    with rfa.Plan((200, 768), scans, dtype=np.float64, clamped=True, epilogue=(-w, 1.0 + w, 0.0), path=3) as plan:
        out = plan.execute([img])[0].cpu().numpy()
    x = img.cpu().numpy()
    assert np.abs(out - ((1.0 + w) * x - w * oracle.apply_filter(x, scans, True))).max() < 1e-10
    with rfa.Plan((4096, 4096), scans, dtype=np.float64, clamped=True) as plan:
        assert plan.path == 3 and list(plan.tiles)[:2] == [256, 32]


@pytest.mark.parametrize("n", [10000, 12345, 8192 * 3 + 4, 100000, 1000003])
def test_1d_fused_any_length(n):
    """Zero-border 1-D signals whose length is not a multiple of 8192 (apps/audio use 10 000 000 samples): the fused kernels
    run on zero-padded copies, as long as no anticausal scan follows a causal one (it would pick up the causal scan's ringing in
    the padding): cascaded causal scans of orders 1..3 on two planes, anticausal scans first, an order-5 scan as sections; a
    causal scan followed by an anticausal one runs as two such stages inside the plan."""
    scans = [(0, True, rc.GAUSS3), (0, True, rc.GAUSS2), (0, True, [0.7, 0.3])]
    imgs, outs, (path, _) = _run((n,), scans, clamped=False, planes=2)
    assert path == 3
    _check(imgs, outs, scans, False)
    scans = [(0, False, rc.GAUSS2), (0, False, [0.6, 0.4]), (0, True, rc.GAUSS3)]
    imgs, outs, (path, _) = _run((n,), scans, clamped=False)
    assert path == 3
    _check(imgs, outs, scans, False)
    o5 = _from_poles([0.8, 0.5 + 0.3j, 0.5 - 0.3j, -0.2 + 0.6j, -0.2 - 0.6j])
    imgs, outs, (path, _) = _run((n,), [(0, False, o5)], clamped=False, inplace=True)
    # (ONE scan of order above 3: the matrix path for lengths that are multiples of 4 samples, round 5; else sections on the fused kernels)
    assert path == (capi.RF_PATH_TILED_MATRIX if n % 4 == 0 else 3)
    _check(imgs, outs, [(0, False, o5)], False)
    imgs, outs, (path, _) = _run((n,), [(0, False, o5)], clamped=False, inplace=True, path=3)
    assert path == 3
    _check(imgs, outs, [(0, False, o5)], False)
    scans = [(0, True, rc.GAUSS2), (0, False, rc.GAUSS2)]
    imgs, outs, (path, _) = _run((n,), scans, clamped=False)
    assert path == 3                    # two stages of an in-plan cascade: each copies only the signal out of its padding
    _check(imgs, outs, scans, False)


@pytest.mark.parametrize("rows", [1, 2, 65, 130, 8 * 64 + 1, 450])
def test_last_tile_row_shorter_than_the_order(rows, plan_flags):
    """A partial last tile row with FEWER rows than the filter order under a clamped border: the tail the next tile receives
    has entries from before the border, which read the scan's first output (tables.h, tail_position).  Found by
    tools/stress_shapes.py at the end of round 2 (8641 x 5776: 135 tile rows and one row); the bug dated from round 1."""
    scans = [(0, False, [0.5, -0.28, 0.42]), (1, True, [0.98, 0.53]), (1, False, [0.7, 0.23, -0.21, -0.38]),
             (1, False, [0.6, 0.3, 0.2])]
    for clamped in (True, False):
        for ty in (64, 32, 128):
            plan_flags(TILED | capi.RF_PLAN_TILE_ROWS(ty))
            imgs, outs, (path, _) = _run((rows, 516), scans, clamped=clamped, planes=2)
            assert path == 3
            _check(imgs, outs, scans, clamped)
    plan_flags(TILED)
    ints = [(1, False, [1.0, 1.0, -1.0, 1.0]), (0, True, [1.0, 1.0])]
    imgs, outs, (path, _) = _run((rows, 260), ints, dtype=np.int32, clamped=True)
    assert path == 3
    _check(imgs, outs, ints, True)


@pytest.mark.parametrize("dtype", [np.float32, np.int32, np.float64], ids=["f32", "i32", "f64"])
@pytest.mark.parametrize("shape,planes,clamped", [((192, 4096), 1, False), ((200, 3004), 1, True), ((96, 1280), 3, True),
                                                  ((130, 260), 1, False), ((64, 8192), 1, True), ((96, 6400), 1, False)])
@pytest.mark.parametrize("order", [1, 2, 3])
def test_x_carry_scan_inside_xscan_rows(shape, planes, clamped, dtype, order):
    """Images of at most 16 tiles per row (32 at order 1): `xscan_rows` completes the x tails itself (kernels_tails.hip,
    XC) and the plan has no `carry_x` launch.  Whole and partial tiles, 16 tiles per row, Tuple planes, both borders,
    causal + anticausal x scans (the chaining terms) -- against the oracle like every other case."""
    import torch
    import recfilter_amd as rfa
    if np.issubdtype(dtype, np.integer):
        fb = [[1.0], [2.0, -1.0], [1.0, -1.0, 1.0]][order - 1]
        scans = [(0, True, [1.0] + fb), (0, False, [1.0, 2.0]), (1, True, [1.0, 1.0]), (1, False, [2.0] + fb)]
        clamped = False
    else:
        co = [rc.BICUBIC_COEFF, rc.GAUSS2, rc.GAUSS3][order - 1]
        scans = [(0, True, co), (0, False, co), (1, True, co), (1, False, [0.7, 0.3])]
    if order == 3 and shape[1] > 2048:
        shape = (shape[0], shape[1] // 8 * 4)
    if order == 3 and dtype == np.float64 and shape[1] > 2048:
        shape = (shape[0], 2048)                                  # (f64, order 3: 16 tiles per row of tails do not fit the LDS)
    if order >= 2 and shape[1] > 4096:
        shape = (shape[0], 4096 - 256 + shape[1] % 256)          # (more than 16 tiles per row: order 1 only)
    imgs = [rc.random_image(shape, dtype, 40 + p) for p in range(planes)]
    with rfa.Plan(shape, scans, dtype=dtype, clamped=clamped, planes=planes, path=3) as plan:
        dev = [torch.from_numpy(im).cuda() for im in imgs]
        outs, times = plan.execute_timed(dev)
        names = [n for n, _ in times]
        outs = [o.cpu().numpy() for o in outs]
    assert "carry_x" not in names and "xscan_rows" in names, names
    _check(imgs, outs, scans, clamped)


_BIQUAD = [0.05, 1.6, -0.7]
_CASCADE_CASES = {
    "five_biquads_1d": dict(shape=(1_000_000,), scans=[(0, True, _BIQUAD)] * 5),
    "nine_biquads_1d": dict(shape=(300_000,), scans=[(0, True, _BIQUAD)] * 9),
    "mixed_causality_padded_1d": dict(shape=(100_000,), scans=[(0, True, _BIQUAD), (0, False, _BIQUAD)]),
    "mixed_causality_padded_1d_four": dict(shape=(123_456,), scans=[(0, True, _BIQUAD), (0, False, _BIQUAD), (0, True, [0.5, 0.5]),
                                                                  (0, False, [0.5, 0.5])]),
    "six_x_two_y_clamped": dict(shape=(300, 1024), scans=[(0, True, [0.5, 0.5])] * 3 + [(0, False, [0.5, 0.5])] * 3 +
                                [(1, True, [0.6, 0.4])] * 2, clamped=True),
    "five_x_int32": dict(shape=(300, 1024), scans=[(0, True, [1.0, 1.0])] * 5 + [(1, True, [1.0, 1.0])], dtype=np.int32),
    "five_x_two_z_volume": dict(shape=(64, 96, 512), scans=[(0, True, [0.5, 0.5])] * 5 + [(2, True, [0.6, 0.4])] * 2),
    "six_x_rgb_partial_tiles": dict(shape=(250, 500), scans=[(0, True, [0.5, 0.5])] * 6 + [(1, False, [0.6, 0.4])], planes=3, clamped=True),
    "five_x_f64": dict(shape=(256, 512), scans=[(0, True, [0.5, 0.3, 0.2])] * 5 + [(1, False, [0.6, 0.4])], dtype=np.float64),
    "five_x_small_line_kernels": dict(shape=(96, 128), scans=[(0, True, [0.5, 0.5])] * 5 + [(1, False, [0.6, 0.4])], small=True),
}


@pytest.mark.parametrize("name", sorted(_CASCADE_CASES))
def test_in_plan_cascade(name, plan_flags):
    """Filters the fused kernels cannot take in one piece -- more than four scans in a dimension; a zero-padded 1-D signal
    whose anticausal scans follow causal ones -- run as successive stages inside one plan (plan.cpp, build_cascade): stage 0
    reads the input, later stages filter the output planes in place.  Against the oracle on the scans as given."""
    import torch
    import recfilter_amd as rfa
    case = dict(_CASCADE_CASES[name])
    shape, scans = case["shape"], case["scans"]
    dtype, clamped, planes = case.get("dtype", np.float32), case.get("clamped", False), case.get("planes", 1)
    if case.get("small"):
        plan_flags(0)        # (the shipped defaults: the suite pins the automatic path off the line kernels)
    imgs = [rc.random_image(shape, dtype, 60 + p) for p in range(planes)]
    with rfa.Plan(shape, scans, dtype=dtype, clamped=clamped, planes=planes) as plan:
        dev = [torch.from_numpy(im).cuda() for im in imgs]
        outs, timed = plan.execute_timed(dev)
        assert any(n.startswith("stage1.") for n, _ in timed), [n for n, _ in timed]
        assert plan.path_name == ("untiled" if case.get("small") else "tiled_fused")
        inplace = [d.clone() for d in dev]
        plan.execute(inplace, inplace)                              # the caller's buffers may be the same, too
        outs = [o.cpu().numpy() for o in outs]
        for a, b in zip(outs, inplace):
            assert np.array_equal(a, b.cpu().numpy())
    _check(imgs, outs, scans, clamped)


def test_in_plan_cascade_keeps_prologue_and_epilogue():
    """The prologue belongs to the first stage, an epilogue without an input operand to the last one."""
    import torch
    import recfilter_amd as rfa
    shape, scans = (256, 512), [(0, True, [0.5, 0.5])] * 5 + [(1, True, [0.5, 0.5])]
    img = rc.random_image(shape, np.float32, 5)
    with rfa.Plan(shape, scans, prologue=(2.0, 0.5), epilogue=(0.5, 0.0, 1.0)) as plan:
        out, timed = plan.execute_timed([torch.from_numpy(img).cuda()])
        assert any(n.startswith("stage1.") for n, _ in timed)
        got = out[0].cpu().numpy()
    assert _pointwise_err(got, img, scans, False, (2.0, 0.5), (0.5, 0.0, 1.0)) < TOL


@pytest.mark.parametrize("n", [12345, 100_001, 1_000_003, 8192 * 3 + 2])
def test_1d_signal_end_is_masked_not_copied(n):
    """A 1-D signal whose length is not a whole number of rows runs on the caller's buffers (FusedArgs::lin_limit): what
    lies behind the signal's end in the input buffer must not reach the result (NaNs there), and nothing may be written
    behind its end in the output buffer (sentinels there)."""
    import torch
    import recfilter_amd as rfa
    scans = [(0, True, rc.GAUSS2), (0, True, [0.7, 0.3])]
    sig = rc.random_image((n,), np.float32, 3)
    big_in = torch.full((n + 64,), float("nan"), device="cuda")
    big_in[:n] = torch.from_numpy(sig).cuda()
    big_out = torch.full((n + 64,), -7.0, device="cuda")
    with rfa.Plan((n,), scans) as plan:
        assert plan.path_name == "tiled_fused"
        _, timed = plan.execute_timed([big_in[:n]], [big_out[:n]])
        assert not any("pad_copy" in name for name, _ in timed)
    got = big_out.cpu().numpy()
    assert np.all(got[n:] == -7.0)
    _check([sig], [got[:n]], scans, False)


@pytest.mark.parametrize("dtype", [np.int32, np.int16], ids=["i32", "i16"])
@pytest.mark.parametrize("n", [12345, 100_001, 8192 * 5 + 7])
def test_1d_integer_signals_any_length(n, dtype):
    """Integer 1-D signals whose length is not a whole number of rows: bit-exact against the untiled path (ring arithmetic
    either way), nothing written behind the signal's end (8-byte chunks for int16 pixels)."""
    import torch
    import recfilter_amd as rfa
    tdt = torch.int32 if dtype == np.int32 else torch.int16
    scans = [(0, True, [1.0, 1.0]), (0, True, [2.0, -1.0, 1.0])]
    big_in = rc.cuda_image((n + 64,), dtype, 1805, lo=-50, hi=50)
    fused = torch.full((n + 64,), -7, dtype=tdt, device="cuda")
    plain = torch.full((n + 64,), -7, dtype=tdt, device="cuda")
    with rfa.Plan((n,), scans, dtype=dtype) as pf, rfa.Plan((n,), scans, dtype=dtype, path=1) as pu:
        assert pf.path_name == "tiled_fused"
        pf.execute([big_in[:n]], [fused[:n]])
        pu.execute([big_in[:n]], [plain[:n]])
        torch.cuda.synchronize()
    assert torch.equal(fused, plain)
    np.testing.assert_array_equal(fused[:n].cpu().numpy(), oracle.apply_filter(big_in[:n].cpu().numpy(), scans, False))


# ---- clamped 1-D signals: the zero-border fused plan plus the border corrections (plan_clamp1d.h) ----------------------
@pytest.mark.parametrize("n", [10_000, 65_536, 100_001, 1_000_003])
@pytest.mark.parametrize("pattern", ["c", "ca", "acca", "ccccc"])
def test_clamped_1d_signals_on_the_fused_kernels(n, pattern):
    rng = np.random.default_rng(len(pattern) * 1000 + n % 997)
    scans = []
    for ch in pattern:
        k = int(rng.integers(1, 4))
        co = [float(rng.uniform(0.3, 1.2))] + [float(-c) for c in np.poly(rng.uniform(-0.85, 0.85, size=k))[1:]]
        scans.append((0, ch == "c", co))
    planes = 2 if n < 200_000 else 1
    imgs, outs, (path, _) = _run((n,), scans, clamped=True, planes=planes, seed=n % 13)
    assert path == 3
    _check(imgs, outs, scans, True)
    imgs, outs, (path, _) = _run((n,), scans, clamped=True, planes=1, seed=3, inplace=True)
    assert path == 3
    _check(imgs, outs, scans, True)


def test_clamped_1d_fallbacks():
    """What the border corrections do not cover stays where it was: filters that do not decay (a running sum), integer
    pixels, signals below 8192 samples, a prologue."""
    import recfilter_amd as rfa
    for kw in (dict(scans=[(0, True, [1.0, 1.0])]), dict(scans=[(0, True, [0.5, 0.4])], dtype=np.int32), dict(scans=[(0, True, [0.5, 0.4])], prologue=(0.5, 0.1))):
        scans = kw.pop("scans")
        with rfa.Plan((100_000,), scans, clamped=True, **kw) as plan:
            assert plan.path != 3
    imgs, outs, (path, _) = _run((5000,), [(0, True, [0.5, 0.4]), (0, False, [0.5, 0.4])], clamped=True)
    assert path != 3
    _check(imgs, outs, [(0, True, [0.5, 0.4]), (0, False, [0.5, 0.4])], True)
    imgs, outs, (path, _) = _run((100_000,), [(0, True, [1.0, 1.0])], clamped=True)
    _check(imgs, outs, [(0, True, [1.0, 1.0])], True)


def test_clamped_1d_high_order_scans():
    """Orders above 3 under a clamped border: the border corrections are built from the scans as given (any order), the
    zero-border plan underneath runs them as sections on the fused kernels."""
    o5 = _from_poles([0.8, 0.5 + 0.3j, 0.5 - 0.3j, -0.2 + 0.6j, -0.2 - 0.6j])
    o8 = _from_poles([0.7, -0.6, 0.5 + 0.4j, 0.5 - 0.4j, -0.3 + 0.5j, -0.3 - 0.5j, 0.2 + 0.7j, 0.2 - 0.7j])
    for scans in ([(0, True, o5), (0, False, o5)], [(0, False, o8)], [(0, True, o8), (0, True, rc.GAUSS3), (0, False, o5)]):
        imgs, outs, (path, _) = _run((300_000,), scans, clamped=True)
        # (ONE scan of order above 3 takes the matrix path since round 5; asked for by name the fused kernels still take it)
        assert path == (capi.RF_PATH_TILED_MATRIX if len(scans) == 1 else 3)
        _check(imgs, outs, scans, True)
        if len(scans) == 1:
            imgs, outs, (path, _) = _run((300_000,), scans, clamped=True, path=3)
            assert path == 3
            _check(imgs, outs, scans, True)
