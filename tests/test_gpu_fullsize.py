"""Full-size parity: every BASELINE.json configuration at the size the bench runs it, HIP path (C ABI) against the
CPU oracle in f64 with the STRICT metric of SURVEY 8d, max |out-ref| / max(|ref|, 1e-6) <= 1e-4 (bit-exact for the
integer summed-area table).

The oracle runs on all host cores (OpenMP over independent lines); 16384^2 takes seconds.  2048^3 (32 GiB per
volume) does not fit a host-side f64 reference, so it is pinned through inputs whose exact result is cheap:
separable (rank-1 / rank-2) volumes, for which the 3-D filter factors into three 1-D filters the oracle evaluates
exactly, with index ranges beyond 2^32, plus equality with the z-sharded execution."""
import os

import numpy as np
import pytest

import oracle
import ref_cases as rc

pytestmark = pytest.mark.gpu

TOL = 1e-4


def _threads():
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, oracle.max_threads(), 64))


def _strict_err(out, want, rows=512):
    """rel_err_strict, evaluated in row blocks (the temporaries of a 16384^2 f64 comparison are 2 GiB each)."""
    out2 = out.reshape(-1, out.shape[-1])
    want2 = want.reshape(-1, want.shape[-1])
    worst = 0.0
    for r in range(0, out2.shape[0], rows):
        worst = max(worst, rc.rel_err_strict(out2[r:r + rows], want2[r:r + rows]))
    return worst


def _floor_err(out, want, rows=512):
    out2 = out.reshape(-1, out.shape[-1])
    want2 = want.reshape(-1, want.shape[-1])
    peak = float(np.max(np.abs(want2)))
    worst = 0.0
    for r in range(0, out2.shape[0], rows):
        d = np.abs(out2[r:r + rows].astype(np.float64) - want2[r:r + rows])
        worst = max(worst, float(np.max(d / np.maximum(np.abs(want2[r:r + rows]), 1e-2 * peak))))
    return worst


def _gpu(shape, scans, clamped, img, dtype=np.float32, flags=None):
    import torch
    import recfilter_amd as rfa
    dev = torch.from_numpy(img).cuda()
    with rfa.Plan(shape, scans, dtype=dtype, clamped=clamped, flags=flags) as plan:
        assert plan.path == 3, plan.path_name                 # the fused kernels, i.e. what bench.py times
        out = plan.execute([dev])[0]
        torch.cuda.synchronize()
    got = out.cpu().numpy()
    del out, dev
    torch.cuda.empty_cache()
    return got


def test_cfg2_summed_table_8192_f32_strict():
    c = rc.BASELINE_CONFIGS["cfg2_summed_table"]
    img = rc.random_image(c["shape"], np.float32, 2)
    got = _gpu(c["shape"], c["scans"], c["clamped"], img)
    want = oracle.apply_filter(img.astype(np.float64), c["scans"], c["clamped"], threads=_threads())
    assert not rc.has_zero_crossings(want)
    assert _strict_err(got, want) < TOL


def test_cfg2_summed_table_8192_int32_bit_exact():
    c = rc.BASELINE_CONFIGS["cfg2_summed_table"]
    img = rc.random_image(c["shape"], np.int32, 3)            # [0, 255]: the table wraps around 2^31 -- still bit-exact
    got = _gpu(c["shape"], c["scans"], c["clamped"], img, dtype=np.int32)
    want = oracle.apply_filter(img, c["scans"], c["clamped"], threads=_threads())
    np.testing.assert_array_equal(got, want)
    np.testing.assert_array_equal(got, img.astype(np.int64).cumsum(0).cumsum(1).astype(np.int32))


def test_cfg3_gaussian_16384_strict():
    """The headline configuration at the size bench.py times it, every pixel against the f64 oracle."""
    c = rc.BASELINE_CONFIGS["cfg3_gaussian2_xy"]
    img = rc.random_image(c["shape"], np.float32, 4)
    got = _gpu(c["shape"], c["scans"], c["clamped"], img)
    want = oracle.apply_filter(img.astype(np.float64), c["scans"], c["clamped"], threads=_threads())
    assert not rc.has_zero_crossings(want)
    assert _strict_err(got, want) < TOL


def test_cfg4b_gaussian3_16384_one_plane_strict():
    c = rc.BASELINE_CONFIGS["cfg4b_gaussian3_rgb"]
    img = rc.random_image(c["shape"], np.float32, 5)
    got = _gpu(c["shape"], c["scans"], c["clamped"], img)
    want = oracle.apply_filter(img.astype(np.float64), c["scans"], c["clamped"], threads=_threads())
    assert not rc.has_zero_crossings(want)
    assert _strict_err(got, want) < TOL


def test_cfg4a_bicubic_16384_one_plane_highpass_floor_metric():
    """The B-spline prefilter is a high-pass: its result on noise crosses zero, where a pointwise relative error is
    meaningless; this is the ONE configuration checked with the floored metric (1 % of the peak), and it also has to
    meet the strict metric on the pixels that stay away from zero."""
    c = rc.BASELINE_CONFIGS["cfg4a_bicubic_rgb"]
    img = rc.random_image(c["shape"], np.float32, 6)
    got = _gpu(c["shape"], c["scans"], c["clamped"], img)
    want = oracle.apply_filter(img.astype(np.float64), c["scans"], c["clamped"], threads=_threads())
    assert rc.has_zero_crossings(want)
    assert _floor_err(got, want) < TOL
    far = np.abs(want) > 0.05
    assert float(np.max(np.abs(got[far] - want[far]) / np.abs(want[far]))) < TOL


def test_cfg4_three_planes_16384_batched_equals_single_plane():
    """cfg4's three planes ride in one launch per step: every plane must equal the single-plane result bit for bit
    (same kernels, same tables), which the two tests above pin against the oracle."""
    import torch
    import recfilter_amd as rfa
    c = rc.BASELINE_CONFIGS["cfg4b_gaussian3_rgb"]
    g = torch.Generator(device="cuda").manual_seed(11)
    planes = [torch.rand(c["shape"], device="cuda", generator=g) for _ in range(3)]
    with rfa.Plan(c["shape"], c["scans"], clamped=True, planes=3) as p3, rfa.Plan(c["shape"], c["scans"], clamped=True) as p1:
        outs = p3.execute(planes)
        for i in range(3):
            one = p1.execute([planes[i]])[0]
            assert torch.equal(one, outs[i])
            del one


def _separable_volume(n, seed, terms):
    """sum over `terms` of a(z) b(y) c(x) with factors that are multiples of 1/64 in [0.5, 1.5): every product is
    exact in f32, so the f32 volume IS the separable function and F(volume) = sum Fz(a) Fy(b) Fx(c) exactly."""
    rng = np.random.default_rng(seed)
    return [[(rng.integers(32, 96, size=n) / 64.0) for _ in range(3)] for _ in range(terms)]


def _check_separable_3d(n, terms, tol=TOL):
    import torch
    import recfilter_amd as rfa
    scans = rc.BASELINE_CONFIGS["cfg5_generic_xyz"]["scans"]
    fac = _separable_volume(n, 17 + n, terms)
    vol = torch.zeros((n, n, n), device="cuda")
    for a, b, c in fac:
        ta, tb, tc = (torch.from_numpy(v.astype(np.float32)).cuda() for v in (a, b, c))
        vol += ta[:, None, None] * tb[None, :, None] * tc[None, None, :]
    with rfa.Plan((n, n, n), scans) as plan:
        assert plan.path == 3
        out = plan.execute([vol])[0]
        torch.cuda.synchronize()
    # the oracle filters the 1-D factors (dimension d of the volume filter = the scans of dimension d, as dim 0 of a line)
    want = []
    for a, b, c in fac:
        fz = oracle.apply_filter(a.astype(np.float64), [(0, s[1], s[2]) for s in scans if s[0] == 2])
        fy = oracle.apply_filter(b.astype(np.float64), [(0, s[1], s[2]) for s in scans if s[0] == 1])
        fx = oracle.apply_filter(c.astype(np.float64), [(0, s[1], s[2]) for s in scans if s[0] == 0])
        want.append(tuple(torch.from_numpy(v).cuda() for v in (fz, fy, fx)))
    worst = 0.0
    step = max(1, (1 << 27) // (n * n))                      # z slabs of <= 1 GiB of f64
    for z0 in range(0, n, step):
        ref = torch.zeros((min(step, n - z0), n, n), device="cuda", dtype=torch.float64)
        for fz, fy, fx in want:
            ref += fz[z0:z0 + step, None, None] * fy[None, :, None] * fx[None, None, :]
        err = ((out[z0:z0 + step].double() - ref).abs() / ref.abs().clamp_min(1e-6)).max()
        worst = max(worst, float(err))
        del ref
    assert worst < tol, f"separable {n}^3, {terms} term(s): strict rel err {worst}"
    return vol, out


def test_cfg5_1024_cubed_against_the_oracle_strict():
    """cfg5's filter at 1024^3 (1 Gi samples) against the f64 oracle on a uniform random volume."""
    import torch
    import recfilter_amd as rfa
    import psutil
    n = 1024
    if psutil.virtual_memory().available < 30 * 2 ** 30:
        pytest.skip("the f64 oracle of 1024^3 needs ~24 GiB of host memory")
    scans = rc.BASELINE_CONFIGS["cfg5_generic_xyz"]["scans"]
    img = np.random.default_rng(8).random((n, n, n), dtype=np.float32)
    got = _gpu((n, n, n), scans, False, img)
    want = img.astype(np.float64)
    del img
    oracle.apply_filter(want, scans, False, threads=_threads(), inplace=True)         # in place: no second 8 GiB copy
    assert _strict_err(got, want, rows=4096) < TOL


def test_cfg5_2048_cubed_separable_volumes_and_sharded_equality():
    """cfg5 at its full size, 2048^3 = 8.6 G samples (indices beyond 2^32): (i) rank-2 separable volume against the
    exact factorised reference, every sample, strict metric; (ii) the z-sharded execution over 8 emulated ranks
    (stepping API, the exchange as bench.py --gpus 8 --strong drives it) reproduces the single-GPU result."""
    import torch
    import recfilter_amd as rfa
    n = 2048
    free, _ = torch.cuda.mem_get_info()
    if free < 150 * 2 ** 30:
        pytest.skip("needs ~130 GiB of device memory")
    scans = rc.BASELINE_CONFIGS["cfg5_generic_xyz"]["scans"]
    vol, out = _check_separable_3d(n, terms=2)
    world = 8
    nz = n // world
    plans = [rfa.Plan((nz, n, n), scans, shard_rank=r, shard_world=world) for r in range(world)]
    outs = [torch.empty((nz, n, n), device="cuda") for _ in range(world)]
    for r in range(world):
        plans[r].begin([vol[r * nz:(r + 1) * nz]], [outs[r]])
    for e in range(plans[0].num_exchanges):
        nbytes = plans[0].exchange_bytes(e)
        gathered = torch.empty(world * nbytes, dtype=torch.uint8, device="cuda")
        for r in range(world):
            plans[r].exchange_local(e, gathered.data_ptr() + r * nbytes)
        for r in range(world):
            plans[r].exchange_apply(e, gathered.data_ptr())
    for r in range(world):
        plans[r].finish()
    torch.cuda.synchronize()
    for r in range(world):
        d = ((outs[r] - out[r * nz:(r + 1) * nz]).abs() / out[r * nz:(r + 1) * nz].abs().clamp_min(1e-6)).max()
        assert float(d) < 1e-5, f"slab {r}: sharded vs single-GPU {float(d)}"
    for p in plans:
        p.close()


def test_volume_between_the_stages_changes_nothing_but_the_workspace():
    """Volumes of 2^28 samples and more: the x/y stage writes a plan-owned intermediate volume that the z stage reads (a final z
    pass that reads and writes the same addresses is 4 % slower: tools/microbench/zpass_shape.hip); RF_PLAN_INPLACE_Z keeps the
    z stage in the output planes.  Same kernels either way: bit-identical results, in place (in == out) too; the workspace
    differs by one volume per plane.  (test_cfg5_1024_cubed_against_the_oracle_strict holds the default against the oracle.)"""
    import torch
    import recfilter_amd as rfa
    from recfilter_amd import capi
    shape = (256, 1024, 1024)                            # 2^28 samples: the smallest volume that takes it
    scans = rc.BASELINE_CONFIGS["cfg5_generic_xyz"]["scans"]
    x = rc.cuda_image(shape, np.float32, 41)
    with rfa.Plan(shape, scans) as plan, rfa.Plan(shape, scans, flags=capi.RF_PLAN_INPLACE_Z) as inplace:
        assert plan.path == 3 and inplace.path == 3
        assert plan.workspace_bytes - inplace.workspace_bytes == x.numel() * 4
        a = plan.execute([x])[0]
        b = inplace.execute([x])[0]
        torch.cuda.synchronize()
        assert torch.equal(a, b)
        y = x.clone()
        c = plan.execute([y], [y])[0]                    # in == out: the intermediate volume is what makes this an out-of-place z pass
        torch.cuda.synchronize()
        assert torch.equal(a, c)


def test_cfg5_256_cubed_separable_matches_direct_oracle():
    """Sanity of the separable reference itself at a size where the direct oracle is cheap: both must agree."""
    import torch
    n = 256
    vol, out = _check_separable_3d(n, terms=2)
    scans = rc.BASELINE_CONFIGS["cfg5_generic_xyz"]["scans"]
    want = oracle.apply_filter(vol.cpu().numpy().astype(np.float64), scans, False, threads=_threads())
    assert rc.rel_err_strict(out.cpu().numpy(), want) < TOL


@pytest.mark.parametrize("name", ["gauss2_clamped", "gauss3_clamped", "generic_xy_zero", "sat", "x_only", "y_only", "single_tile"])
def test_streaming_pass1_on_small_shapes(name):
    """The streaming pass 1 (kernels_stream.hip: LDS-DMA ring, matrix-core contractions) normally takes images of
    >= 2048 tiles; lowered to one tile here so that border variants, few-tile walkers (fewer slots than the ring is
    deep) and every scan mix go through it at sizes the oracle checks in full (RF_PLAN_STREAM_PASS1, on the 64-row tiles
    the kernel is written for)."""
    import torch
    import recfilter_amd as rfa
    from recfilter_amd import capi
    flags = capi.RF_PLAN_TILED_ONLY | capi.RF_PLAN_STREAM_PASS1 | capi.RF_PLAN_TILE_ROWS(64)
    case = rc.FUSED_CASES[name]
    for shape in (case["shape"], (192, 1024), (64 * 5, 256 * 3)):
        if shape[0] % 64 or shape[1] % 256:
            continue
        img = rc.random_image(shape, np.float32, 21)
        got = _gpu(shape, case["scans"], case["clamped"], img, flags=flags)
        want = oracle.apply_filter(img.astype(np.float64), case["scans"], case["clamped"])
        assert rc.rel_err(got, want) < TOL, (name, shape)


def test_streaming_pass1_3d_and_planes():
    import torch
    import recfilter_amd as rfa
    from recfilter_amd import capi
    # volumes, plane batches and > 4 tails per dimension are staged by default
    flags = capi.RF_PLAN_TILED_ONLY | capi.RF_PLAN_STREAM_PASS1 | capi.RF_PLAN_TILE_ROWS(64)
    scans = rc.BASELINE_CONFIGS["cfg5_generic_xyz"]["scans"]
    img = rc.random_image((24, 128, 512), np.float32, 22)
    got = _gpu((24, 128, 512), scans, False, img, flags=flags)
    assert rc.rel_err(got, oracle.apply_filter(img.astype(np.float64), scans, False)) < TOL
    c = rc.BASELINE_CONFIGS["cfg4b_gaussian3_rgb"]
    imgs = [rc.random_image((128, 768), np.float32, 30 + p) for p in range(3)]
    with rfa.Plan((128, 768), c["scans"], clamped=True, planes=3, flags=flags) as plan:
        outs = plan.execute([torch.from_numpy(i).cuda() for i in imgs])
        torch.cuda.synchronize()
    for im, o in zip(imgs, outs):
        assert rc.rel_err(o.cpu().numpy(), oracle.apply_filter(im.astype(np.float64), c["scans"], True)) < TOL


@pytest.mark.parametrize("mode", ["mfma", "staged"])
@pytest.mark.parametrize("name", ["gauss2_clamped", "gauss3_clamped", "generic_xy_zero", "sat", "x_only", "y_only", "single_tile"])
def test_both_staged_pass1_kernels_on_whole_tiles(name, mode):
    """Pass 1 of f32 images made of whole tiles has two register-staged kernels: mfma_tails_kernel (x tails on the matrix
    cores, per-row LDS swizzle; default for orders 2 and 3) and fused_tails_kernel (vector ALU; order 1 and everything the
    other one does not take).  RF_PLAN_MFMA_PASS1 / RF_PLAN_STAGED_PASS1 send every order to one or the other: all three
    tile heights, every scan mix of the fused cases, border variants (one tile, a tile row, a tile column), 3 planes."""
    import torch
    import recfilter_amd as rfa
    from recfilter_amd import capi
    pick = capi.RF_PLAN_MFMA_PASS1 if mode == "mfma" else capi.RF_PLAN_STAGED_PASS1
    case = rc.FUSED_CASES[name]
    for ty in (32, 64, 128):
        flags = capi.RF_PLAN_TILED_ONLY | pick | capi.RF_PLAN_TILE_ROWS(ty)
        for shape in ((ty, 256), (ty * 3, 256), (ty, 1024), (ty * 2, 768)):
            img = rc.random_image(shape, np.float32, 23)
            got = _gpu(shape, case["scans"], case["clamped"], img, flags=flags)
            want = oracle.apply_filter(img.astype(np.float64), case["scans"], case["clamped"])
            assert rc.rel_err(got, want) < TOL, (name, mode, ty, shape)
    # Tuple planes in one launch, and a volume (z = the grid's batch dimension)
    c = rc.BASELINE_CONFIGS["cfg4b_gaussian3_rgb"]
    flags = capi.RF_PLAN_TILED_ONLY | pick | capi.RF_PLAN_TILE_ROWS(128)
    imgs = [rc.random_image((256, 768), np.float32, 40 + p) for p in range(3)]
    with rfa.Plan((256, 768), c["scans"], clamped=True, planes=3, flags=flags) as plan:
        outs = plan.execute([torch.from_numpy(i).cuda() for i in imgs])
        torch.cuda.synchronize()
    for im, o in zip(imgs, outs):
        assert rc.rel_err(o.cpu().numpy(), oracle.apply_filter(im.astype(np.float64), c["scans"], True)) < TOL
    scans = rc.BASELINE_CONFIGS["cfg5_generic_xyz"]["scans"]
    vol = rc.random_image((24, 128, 512), np.float32, 24)
    got = _gpu((24, 128, 512), scans, False, vol, flags=capi.RF_PLAN_TILED_ONLY | pick | capi.RF_PLAN_TILE_ROWS(64))
    assert rc.rel_err(got, oracle.apply_filter(vol.astype(np.float64), scans, False)) < TOL


@pytest.mark.parametrize("case", ["rows64_both", "tall_both", "odd_width", "planes_int32", "volume"])
def test_final_pass_split_into_whole_tiles_and_edge_strips(case):
    """Large images with partial tiles, every sample against the oracle: on 128-row tiles the final pass is up to three
    launches -- the whole tiles on the lean kernel, the last tile column and the last tile row on the EDGE variant
    (FusedArgs::tx0 ..; kernels_fused_tall.hip) --, on 64- and 32-row tiles one launch of the EDGE variant (mid-size images
    whose height is not a multiple of 64 take 32-row tiles)."""
    import torch
    import recfilter_amd as rfa
    if case == "rows64_both":
        shape, scans, clamped, planes, dtype, ty = (8292, 8228), rc.xy_pm(rc.GAUSS2), True, 1, np.float32, 64
    elif case == "tall_both":
        shape, scans, clamped, planes, dtype, ty = (16380, 8188), rc.xy_pm(rc.GAUSS3), True, 1, np.float32, 128
    elif case == "odd_width":
        shape, scans, clamped, planes, dtype, ty = (4200, 8193), rc.REFERENCE_TESTS["test_generic_xy"]["scans"], False, 1, np.float32, 32
    elif case == "planes_int32":
        shape, scans, clamped, planes, dtype, ty = (2100, 4200), [(0, True, [1.0, 1.0]), (0, False, [1.0, 1.0, -1.0]), (1, True, [1.0, 2.0, -1.0])], False, 3, np.int32, 32
    else:
        shape, scans, clamped, planes, dtype, ty = (40, 1030, 1040), rc.BASELINE_CONFIGS["cfg5_generic_xyz"]["scans"], False, 1, np.float32, 64
    rng = np.random.default_rng(5)
    if dtype == np.int32:
        imgs = [rng.integers(0, 4, size=shape).astype(np.int32) for _ in range(planes)]
    else:
        imgs = [rc.random_image(shape, np.float32, 50 + p) for p in range(planes)]
    dev = [torch.from_numpy(im).cuda() for im in imgs]
    with rfa.Plan(shape, scans, dtype=dtype, clamped=clamped, planes=planes) as plan:
        assert plan.path == 3 and list(plan.tiles)[1] == ty, plan.tiles
        outs = [o.cpu().numpy() for o in plan.execute(dev)]
        torch.cuda.synchronize()
    for im, out in zip(imgs, outs):
        if dtype == np.int32:
            np.testing.assert_array_equal(out, oracle.apply_filter(im, scans, clamped, threads=_threads()))
        else:
            want = oracle.apply_filter(im.astype(np.float64), scans, clamped, threads=_threads())
            assert _floor_err(out, want) < TOL


# ---- the reference's high-order audio sweep at bench size: 10,000,000 samples, orders 1, 3, .. 29 in their direct form --------
def test_audio_high_order_sweep_10m_samples():
    """apps/audio/audio_filter_high_order.cpp:38-73 at the size the audio apps are quoted on: every order of the sweep, one causal
    scan with the app's taps, against the f64 oracle on ALL samples (78125 tiles of 128: every level of the carry chain)."""
    import torch
    import recfilter_amd as rfa
    from recfilter_amd import capi
    n = 10_000_000
    sig = rc.random_image((n,), np.float32, 21)
    dev = torch.from_numpy(sig).cuda()
    for order in range(1, 30, 2):
        scans = [(0, True, [1.0] + [0.01] * order)]
        with rfa.Plan((n,), scans) as plan:
            assert plan.path == (capi.RF_PATH_TILED_MATRIX if order > 3 else 3), (order, plan.path_name)
            got = plan.execute([dev])[0].cpu().numpy()
        want = oracle.apply_filter(sig.astype(np.float64), scans, False)
        assert rc.rel_err(got, want) < TOL, order


def test_order_12_and_32_images_4096_against_oracle():
    """2-D causal + anticausal x / y filters of order 12 (clamped) and 32 (zero border) at 4096^2 on the matrix path."""
    import torch
    import recfilter_amd as rfa
    from recfilter_amd import capi
    rng = np.random.default_rng(5)
    for order, clamped in ((12, True), (32, False)):
        a = rng.standard_normal(order) * np.exp(-0.15 * np.arange(order))
        a *= 0.85 / np.abs(a).sum()
        c = [0.4] + [float(np.float32(v)) for v in a]
        scans = [(0, True, c), (0, False, c), (1, True, c), (1, False, c)]
        img = rc.random_image((4096, 4096), np.float32, 3)
        dev = torch.from_numpy(img).cuda()
        with rfa.Plan(img.shape, scans, clamped=clamped) as plan:
            # order 12: a pair stage per dimension (tiles of 128: pass 1, chain + apply of the causal scan, cross term and its clamped
            # border, chain + apply of the anticausal scan, ONE final pass); order 32: tiles of 256 in both dimensions, one stage per
            # scan, the y stage with its own pass 1
            assert plan.path == capi.RF_PATH_TILED_MATRIX and plan.tiles[:2] == ((128, 128) if order == 12 else (256, 256))
            assert plan.num_kernels == (2 * (1 + 2 + 2 + 2 + 1) if order == 12 else 1 + 2 + 2 + 1 + 2 + 2)
            got = plan.execute([dev])[0].cpu().numpy()
        want = oracle.apply_filter(img.astype(np.float64), scans, clamped, threads=_threads())
        assert _floor_err(got, want) < TOL, (order, clamped)


def test_matrix_path_268m_samples_index_ranges():
    """A 1-D signal of 2^28 + 4 samples (1 GiB; 2^21 tiles: three chain levels, element offsets beyond 2^31 bytes, a last tile no
    tile width divides) through a causal scan of order 29 and an anticausal one of order 12, against the f64 oracle on all samples."""
    import torch
    import recfilter_amd as rfa
    from recfilter_amd import capi
    n = (1 << 28) + 4
    rng = np.random.default_rng(8)
    a = rng.standard_normal(12) * np.exp(-0.15 * np.arange(12))
    a *= 0.85 / np.abs(a).sum()
    scans = [(0, True, [1.0] + [0.01] * 29), (0, False, [0.4] + [float(np.float32(v)) for v in a])]
    sig = rng.random(n, dtype=np.float32)
    dev = torch.from_numpy(sig).cuda()
    with rfa.Plan((n,), scans, flags=capi.RF_PLAN_NO_OVERLAP) as plan:
        assert plan.path == capi.RF_PATH_TILED_MATRIX
        got = plan.execute([dev])[0].cpu().numpy()
    del dev
    torch.cuda.empty_cache()
    want = oracle.apply_filter(sig.astype(np.float64), scans, False)
    worst = 0.0
    for lo in range(0, n, 1 << 24):
        worst = max(worst, rc.rel_err(got[lo:lo + (1 << 24)], want[lo:lo + (1 << 24)]))
    assert worst < TOL


@pytest.mark.parametrize("clamped", [False, True])
def test_matrix_path_16384_lines_chain_in_one_go(clamped):
    """16384 lines fill the chip by themselves: their 40 tiles are chained in ONE launch per scan (no chunks, no propagation:
    plan_matrix.cpp, kMxTopWide) -- a pair stage along x of a 16384 x 5120 image against the f64 oracle on every pixel."""
    import torch
    import recfilter_amd as rfa
    from recfilter_amd import capi
    rng = np.random.default_rng(6)
    def coeff(order):
        a = rng.standard_normal(order) * np.exp(-0.15 * np.arange(order))
        return [0.4] + [float(np.float32(v)) for v in a * 0.85 / np.abs(a).sum()]
    scans = [(0, True, coeff(12)), (0, False, coeff(9))]
    img = rc.random_image((16384, 5120), np.float32, 4)
    dev = torch.from_numpy(img).cuda()
    with rfa.Plan(img.shape, scans, clamped=clamped) as plan:
        assert plan.path == capi.RF_PATH_TILED_MATRIX and plan.tiles[0] == 128
        # pass 1, chain (causal), chain (anticausal, the cross term and its clamped border on the way), the pair's final pass
        assert plan.num_kernels == 4
        got = plan.execute([dev])[0].cpu().numpy()
    del dev
    want = oracle.apply_filter(img.astype(np.float64), scans, clamped, threads=_threads())
    assert _floor_err(got, want) < TOL
