"""GPU parity of pass 1 in one read of a 3-D volume (recfilter_amd/csrc/kernels_tails_walk.hip): the plan whose first pass
forms the x, y AND z tails (the z operators commuted in front of the x/y filter) against the CPU oracle and against the
plan that keeps the two first passes (RF_PLAN_STAGED_PASS1), through the C ABI.  Reference semantics: the scan loops of
lib/recfilter.cpp:302-343 (the oracle), tiling algebra lib/split.cpp:256-499, 1008-1130."""
import numpy as np
import pytest

import oracle
import ref_cases as rc

pytestmark = pytest.mark.gpu

XYZ = rc.REFERENCE_TESTS["test_generic_xyz"]["scans"]
ORDER1 = [(0, True, [0.4, 0.6]), (0, False, [0.5, 0.5]), (1, True, [0.3, 0.7]), (1, False, [0.45, 0.55]),
          (2, True, [0.5, 0.5]), (2, False, [0.6, 0.4])]
ONE_EACH = [(0, True, [1.0, 0.5, 0.25]), (1, False, [1.0, 0.5, 0.125]), (2, True, [1.0, 0.5, 0.0625])]
Z_MIXED = XYZ[:4] + [(2, True, [1.0, 0.5, 0.25]), (2, False, [0.7, 0.3])]
Z_ORDER1 = XYZ[:4] + [(2, True, [0.5, 0.5]), (2, False, [0.7, 0.3])]
XY_ORDER1_Z2 = ORDER1[:4] + XYZ[4:]
G3 = rc.GAUSS3                     # order 3 along x and y (the one-read pass takes it since round 5), order <= 2 along z
XY_ORDER3 = [(0, True, G3), (0, False, G3), (1, True, G3), (1, False, G3)] + XYZ[4:]
XY_ORDER3_ONE_EACH = [(0, False, G3), (1, True, G3), (2, True, [0.5, 0.5])]
XY_ORDERS_3_AND_1 = [(0, True, G3), (0, False, [0.5, 0.5]), (1, False, [0.6, 0.3, 0.1]), (2, True, [1.0, 0.5, 0.25]), (2, False, [0.7, 0.3])]
TWO_X_ONE_Y = XYZ[:3] + XYZ[4:]
ONE_X_TWO_Y = XYZ[1:]


def _cases():
    from recfilter_amd import capi
    return {
        "xyz_zero": ((64, 64, 256), XYZ, False, 0),
        "xyz_clamped_two_tile_columns": ((64, 96, 512), XYZ, True, 0),
        "two_patches_per_y_tile": ((64, 128, 256), XYZ, False, capi.RF_PLAN_TILE_ROWS(64)),
        "four_patches_per_y_tile_clamped": ((32, 256, 512), XYZ, True, capi.RF_PLAN_TILE_ROWS(128)),
        "two_z_tiles_clamped": ((128, 64, 256), XYZ, True, capi.RF_PLAN_TILE_PLANES(64)),
        "three_z_tiles_of_32": ((96, 64, 256), ORDER1, True, 0),
        "z_tile_128": ((256, 32, 256), XYZ, True, capi.RF_PLAN_TILE_PLANES(128)),
        "one_scan_per_dimension": ((64, 64, 512), ONE_EACH, False, 0),
        "z_orders_2_and_1": ((64, 64, 256), Z_MIXED, True, 0),
        "z_order_1_under_xy_order_2": ((64, 64, 256), Z_ORDER1, True, 0),
        "z_order_2_over_xy_order_1": ((64, 64, 256), XY_ORDER1_Z2, False, 0),
        "two_x_scans_one_y_scan": ((64, 64, 512), TWO_X_ONE_Y, True, 0),
        "one_x_scan_two_y_scans": ((64, 128, 256), ONE_X_TWO_Y, True, capi.RF_PLAN_TILE_ROWS(64)),
        "xy_order_3_clamped": ((64, 128, 512), XY_ORDER3, True, capi.RF_PLAN_TILE_ROWS(64)),
        "xy_order_3_zero_four_patches": ((32, 128, 256), XY_ORDER3, False, capi.RF_PLAN_TILE_ROWS(128)),
        "xy_order_3_one_scan_each": ((64, 64, 256), XY_ORDER3_ONE_EACH, True, 0),
        "xy_orders_3_and_1_mixed": ((64, 64, 512), XY_ORDERS_3_AND_1, True, 0),
    }


CASE_NAMES = ["xyz_zero", "xyz_clamped_two_tile_columns", "two_patches_per_y_tile", "four_patches_per_y_tile_clamped",
              "two_z_tiles_clamped", "three_z_tiles_of_32", "z_tile_128", "one_scan_per_dimension", "z_orders_2_and_1",
              "z_order_1_under_xy_order_2", "z_order_2_over_xy_order_1", "two_x_scans_one_y_scan", "one_x_scan_two_y_scans",
              "xy_order_3_clamped", "xy_order_3_zero_four_patches", "xy_order_3_one_scan_each", "xy_orders_3_and_1_mixed"]


def _run(shape, scans, clamped, flags, img, in_place=False):
    import torch
    import recfilter_amd as rfa
    from recfilter_amd import capi
    x = torch.from_numpy(img).cuda()
    out = x if in_place else torch.empty_like(x)
    with rfa.Plan(shape, scans, clamped=clamped, flags=flags | (0 if flags & capi.RF_PLAN_STAGED_PASS1 else capi.RF_PLAN_WALK_PASS1),
                  path=capi.RF_PATH_TILED_FUSED) as plan:
        _, timed = plan.execute_timed([x], [out])
        torch.cuda.synchronize()
        if in_place:
            x = torch.from_numpy(img).cuda()
            out = x
        plan.execute([x], [out])
        torch.cuda.synchronize()
    return out.cpu().numpy(), [k for k, _ in timed]


@pytest.mark.parametrize("name", CASE_NAMES)
def test_one_read_pass1_against_oracle_and_staged_plan(name):
    from recfilter_amd import capi
    shape, scans, clamped, flags = _cases()[name]
    img = np.random.default_rng(11).random(shape, dtype=np.float32)
    want = oracle.apply_filter(img.astype(np.float64), scans, clamped)
    got, steps = _run(shape, scans, clamped, flags, img)
    assert "walk_tails" in steps and "carry_planes_xy" in steps and "strided_pass1_z" not in steps, steps
    staged, steps_staged = _run(shape, scans, clamped, flags | capi.RF_PLAN_STAGED_PASS1, img)
    assert "walk_tails" not in steps_staged and "strided_pass1_z" in steps_staged, steps_staged
    # tolerance: f32 arithmetic in a different summation order than the oracle's f64 loops (1e-4 strict; here the
    # looser-to-fail max-norm bar at 2e-6, what the staged plan itself reaches)
    # (order 3: the sigma-5 Gaussian's third-order recurrence is itself worth 6e-6 in f32, on either plan)
    scale = np.abs(want).max()
    bar = 2e-5 if name.startswith("xy_order") else 2e-6
    assert np.abs(got - want).max() / scale < bar
    assert np.abs(staged - want).max() / scale < bar
    assert np.abs(got - want).max() < 3.0 * max(np.abs(staged - want).max(), 1e-6 * scale)      # no worse than two first passes
    assert rc.rel_err_strict(got, want) < 1e-4


def test_one_read_pass1_in_place():
    shape, scans, clamped, flags = _cases()["four_patches_per_y_tile_clamped"]
    img = np.random.default_rng(12).random(shape, dtype=np.float32)
    want = oracle.apply_filter(img.astype(np.float64), scans, clamped)
    got, steps = _run(shape, scans, clamped, flags, img, in_place=True)
    assert "walk_tails" in steps
    assert np.abs(got - want).max() / np.abs(want).max() < 2e-6


@pytest.mark.parametrize("clamped", [False, True])
def test_one_read_pass1_tuple_planes(clamped):
    """A Tuple volume (three planes, order 3 along x / y -- the shape of BASELINE config 4 with a third dimension): every plane
    through the one-read pass 1, its own z tails, parts and carry planes (round 5)."""
    import torch
    import recfilter_amd as rfa
    from recfilter_amd import capi
    shape, planes = (64, 128, 512), 3
    rng = np.random.default_rng(17)
    imgs = [rng.random(shape, dtype=np.float32) for _ in range(planes)]
    xs = [torch.from_numpy(im).cuda() for im in imgs]
    outs = [torch.empty_like(x) for x in xs]
    with rfa.Plan(shape, XY_ORDER3, clamped=clamped, planes=planes, flags=capi.RF_PLAN_WALK_PASS1 | capi.RF_PLAN_TILE_ROWS(64),
                  path=capi.RF_PATH_TILED_FUSED) as plan:
        _, timed = plan.execute_timed(xs, outs)
        torch.cuda.synchronize()
    steps = [k for k, _ in timed]
    assert "walk_tails" in steps and "carry_planes_xy" in steps and "strided_pass1_z" not in steps, steps
    for im, out in zip(imgs, outs):
        want = oracle.apply_filter(im.astype(np.float64), XY_ORDER3, clamped)
        assert np.abs(out.cpu().numpy() - want).max() / np.abs(want).max() < 2e-5


@pytest.mark.parametrize("clamped", [False, True])
def test_one_read_pass1_with_a_prologue(clamped):
    """A pointwise prologue x' = 0.5 x + 0.25 (define's RHS expression, lib/recfilter.cpp:197-238) is applied to the samples as the
    one-read pass 1 loads them: the x / y tails AND the z tails are those of the transformed volume (round 5)."""
    import torch
    import recfilter_amd as rfa
    from recfilter_amd import capi
    shape = (64, 64, 512)
    rng = np.random.default_rng(19)
    img = rng.random(shape, dtype=np.float32)
    x = torch.from_numpy(img).cuda()
    out = torch.empty_like(x)
    with rfa.Plan(shape, XYZ, clamped=clamped, flags=capi.RF_PLAN_WALK_PASS1, path=capi.RF_PATH_TILED_FUSED, prologue=(0.5, 0.25)) as plan:
        _, timed = plan.execute_timed([x], [out])
        torch.cuda.synchronize()
    steps = [k for k, _ in timed]
    assert "walk_tails" in steps and "strided_pass1_z" not in steps, steps
    want = oracle.apply_filter(img.astype(np.float64) * 0.5 + 0.25, XYZ, clamped)
    assert np.abs(out.cpu().numpy() - want).max() / np.abs(want).max() < 2e-6


@pytest.mark.parametrize("shape", [(64, 80, 256), (64, 64, 300), (32, 100, 520), (64, 33, 260), (32, 160, 1020),
                                   (64, 64, 258), (32, 100, 521), (32, 77, 1023), (64, 40, 255), (32, 64, 1)])
@pytest.mark.parametrize("clamped", [False, True])
def test_one_read_pass1_partial_tiles(shape, clamped):
    """Heights that are not whole tile rows and widths that are not whole tiles: what does not exist loads as zeros and is never
    stored (round 5; the staged pass 1 has always done so, lib/split.cpp:503-665 takes any extent the tile divides).  Round 6:
    widths that are not multiples of four as well (rows only element-aligned: 4-byte loads, a partial last chunk, z tails
    stored sample by sample) -- 258, 521, 1023, 255 and a volume one sample wide."""
    import torch
    import recfilter_amd as rfa
    from recfilter_amd import capi
    rng = np.random.default_rng(23)
    img = rng.random(shape, dtype=np.float32)
    x = torch.from_numpy(img).cuda()
    out = torch.empty_like(x)
    for scans, pro in ((XYZ, None), (XY_ORDER3, (0.5, 0.25))):
        kw = dict(prologue=pro) if pro else {}
        with rfa.Plan(shape, scans, clamped=clamped, flags=capi.RF_PLAN_WALK_PASS1, path=capi.RF_PATH_TILED_FUSED, **kw) as plan:
            _, timed = plan.execute_timed([x], [out])
            torch.cuda.synchronize()
        steps = [k for k, _ in timed]
        assert "walk_tails" in steps and "strided_pass1_z" not in steps, steps
        src = img.astype(np.float64) * pro[0] + pro[1] if pro else img.astype(np.float64)
        want = oracle.apply_filter(src, scans, clamped)
        assert np.abs(out.cpu().numpy() - want).max() / np.abs(want).max() < (2e-5 if scans is XY_ORDER3 else 2e-6), (shape, clamped, pro)


@pytest.mark.parametrize("shape", [(32, 200, 444), (32, 128, 356), (32, 129, 388), (64, 256, 512), (32, 70, 132), (32, 250, 640),
                                   (32, 200, 445), (32, 129, 387), (32, 140, 131)])
@pytest.mark.parametrize("clamped", [False, True])
def test_one_read_pass1_tall_patches(shape, clamped):
    """128-row y tiles, orders <= 2 along x / y: the pass runs on patches of 128 columns x 64 rows (round 6) -- two parts of
    the combined rows per y tile instead of four, the x tails of a tile in two parts (its column halves) that the carry scan
    along x adds up.  Whole tiles; a last tile whose right half is partial, missing, or four columns wide; a last tile row of
    72, 1 and 6 rows; one scan per dimension and order 1 as well."""
    import torch
    import recfilter_amd as rfa
    from recfilter_amd import capi
    rng = np.random.default_rng(29)
    img = rng.random(shape, dtype=np.float32)
    x = torch.from_numpy(img).cuda()
    out = torch.empty_like(x)
    flags = capi.RF_PLAN_WALK_PASS1 | capi.RF_PLAN_TILE_ROWS(128)
    for scans, pro in ((XYZ, None), (ONE_EACH, (0.5, 0.25)), (ORDER1, None), (TWO_X_ONE_Y, None)):
        kw = dict(prologue=pro) if pro else {}
        with rfa.Plan(shape, scans, clamped=clamped, flags=flags, path=capi.RF_PATH_TILED_FUSED, **kw) as plan:
            assert list(plan.tiles)[:2] == [256, 128]
            _, timed = plan.execute_timed([x], [out])
            torch.cuda.synchronize()
        steps = [k for k, _ in timed]
        assert "walk_tails" in steps and "strided_pass1_z" not in steps, steps
        src = img.astype(np.float64) * pro[0] + pro[1] if pro else img.astype(np.float64)
        want = oracle.apply_filter(src, scans, clamped)
        assert np.abs(out.cpu().numpy() - want).max() / np.abs(want).max() < 2e-6, (shape, clamped, len(scans))


@pytest.mark.parametrize("what", ["int32", "u8_input"])   # (asked for, refused by the shape rules)
def test_volumes_the_one_read_pass_does_not_take_keep_two_first_passes(what):
    """Integer pixels (the pass contracts on the f32 matrix cores; with the products on the vector ALU it ran as long as the two
    passes it replaces: NOTES round 4) and 8-bit input: the z stage runs its own first pass.  (Widths that are no multiples of
    four take the one-read pass since round 6: test_one_read_pass1_partial_tiles.)"""
    import torch
    import recfilter_amd as rfa
    from recfilter_amd import capi
    kw = dict(clamped=False, path=capi.RF_PATH_TILED_FUSED)
    shape, dtype, planes = (64, 64, 256), np.float32, 1
    if what == "int32":
        dtype = np.int32
        scans = [(0, True, [1, 1]), (1, True, [1, 1]), (2, True, [1, 1])]
    else:
        scans = XYZ
    if what == "u8_input":
        kw["input_dtype"] = np.uint8
    rng = np.random.default_rng(13)
    if what == "u8_input":
        imgs = [rng.integers(0, 256, shape).astype(np.uint8)]
    else:
        imgs = [(rng.integers(0, 5, shape).astype(dtype) if dtype == np.int32 else rng.random(shape, dtype=np.float32)) for _ in range(planes)]
    xs = [torch.from_numpy(im).cuda() for im in imgs]
    outs = [torch.empty(shape, dtype=torch.int32 if dtype == np.int32 else torch.float32, device="cuda") for _ in xs]
    with rfa.Plan(shape, scans, dtype=dtype, planes=planes, flags=capi.RF_PLAN_WALK_PASS1, **kw) as plan:
        _, timed = plan.execute_timed(xs, outs)
        torch.cuda.synchronize()
    steps = [k for k, _ in timed]
    assert "walk_tails" not in steps and "strided_pass1_z" in steps, steps
    for im, out in zip(imgs, outs):
        if dtype == np.int32:
            np.testing.assert_array_equal(out.cpu().numpy(), oracle.apply_filter(im, scans, False))
        else:
            want = oracle.apply_filter(im.astype(np.float64), scans, False)
            assert np.abs(out.cpu().numpy() - want).max() / np.abs(want).max() < 2e-6


def test_one_read_pass1_is_the_default_from_256_patch_columns_on():
    """Default choice (no flag): 128 x 512 x 512 has 2 x 16 x 2 = 64 patch columns -> two first passes; 512 x 512 x 512 has
    256 -> one read, with either border.  Each against the f64 oracle, strict metric (1e-4)."""
    import os
    import torch
    import recfilter_amd as rfa
    rng = np.random.default_rng(14)
    threads = min(16, os.cpu_count() or 1)
    # (a width that is no multiple of four: the pass exists for it -- RF_PLAN_WALK_PASS1 -- but two first passes are faster there)
    for shape, clamped, expect in (((128, 512, 512), False, False), ((512, 512, 512), False, True), ((512, 512, 512), True, True),
                                   ((512, 512, 514), False, False)):
        x = torch.from_numpy(rng.random(shape, dtype=np.float32)).cuda()
        with rfa.Plan(shape, XYZ, clamped=clamped) as plan:
            out, timed = plan.execute_timed([x])
            torch.cuda.synchronize()
        steps = [k for k, _ in timed]
        assert ("walk_tails" in steps) == expect, (shape, steps)
        want = oracle.apply_filter(x.cpu().numpy().astype(np.float64), XYZ, clamped, threads=threads)
        assert rc.rel_err_strict(out[0].cpu().numpy(), want) < 1e-4, (shape, clamped)


def test_one_read_pass1_with_an_epilogue():
    """out = w_f * F(x) + w_i * x + b behind the z stage (rf_pointwise_desc, the unsharp-mask form): the plan still reads the
    volume once for its tails; the helper that filters the z carry planes applies F alone."""
    import torch
    import recfilter_amd as rfa
    from recfilter_amd import capi
    shape, scans, clamped, flags = _cases()["two_z_tiles_clamped"]
    img = np.random.default_rng(15).random(shape, dtype=np.float32)
    x = torch.from_numpy(img).cuda()
    for epi in ((0.5, 0.0, 0.25), (-1.0, 2.0, 0.0)):
        with rfa.Plan(shape, scans, clamped=clamped, flags=flags | capi.RF_PLAN_WALK_PASS1, path=capi.RF_PATH_TILED_FUSED,
                      epilogue=epi) as plan:
            out, timed = plan.execute_timed([x])
            torch.cuda.synchronize()
        assert "walk_tails" in [k for k, _ in timed]
        f = oracle.apply_filter(img.astype(np.float64), scans, clamped)
        want = epi[0] * f + epi[1] * img.astype(np.float64) + epi[2]
        # (the second form cancels: the bar is relative to the filtered signal's magnitude, as tests/test_harness.py does)
        assert np.abs(out[0].cpu().numpy() - want).max() / np.abs(f).max() < 4e-6


def test_one_read_plan_executes_concurrently_on_distinct_streams():
    """executes of one plan on distinct streams may overlap.  The one-read plan carries a helper plan (the x/y filter
    over the z carry planes) and a workspace of parts: three volumes through ONE plan on three busy streams, then from three
    host threads at once; every result against the oracle."""
    import threading
    import torch
    import recfilter_amd as rfa
    from recfilter_amd import capi
    shape, scans, clamped, flags = _cases()["four_patches_per_y_tile_clamped"]
    rng = np.random.default_rng(16)
    imgs = [rng.random(shape, dtype=np.float32) for _ in range(3)]
    wants = [oracle.apply_filter(im.astype(np.float64), scans, clamped) for im in imgs]
    dev = [torch.from_numpy(im).cuda() for im in imgs]
    streams = [torch.cuda.Stream() for _ in range(3)]
    with rfa.Plan(shape, scans, clamped=clamped, flags=flags | capi.RF_PLAN_WALK_PASS1, path=capi.RF_PATH_TILED_FUSED) as plan:
        outs = [torch.empty_like(d) for d in dev]
        for st in streams:
            with torch.cuda.stream(st):
                torch.cuda._sleep(30_000_000)
        for rep in range(2):
            for i in range(3):
                plan.execute([dev[i]], [outs[i]], stream=streams[i])
        torch.cuda.synchronize()
        assert plan.num_instances == 3, plan.num_instances
        for o, w in zip(outs, wants):
            assert np.abs(o.cpu().numpy() - w).max() / np.abs(w).max() < 2e-6
        outs2 = [torch.empty_like(d) for d in dev]
        errors = []

        def worker(i):
            try:
                for _ in range(3):
                    plan.execute([dev[i]], [outs2[i]], stream=streams[i])
            except Exception as exc:      # pragma: no cover
                errors.append(exc)
        threads = [threading.Thread(target=worker, args=(i,)) for i in range(3)]
        for th in threads:
            th.start()
        for th in threads:
            th.join()
        torch.cuda.synchronize()
        assert not errors, errors
        for o, w in zip(outs2, wants):
            assert np.abs(o.cpu().numpy() - w).max() / np.abs(w).max() < 2e-6


@pytest.mark.parametrize("clamped", [False, True])
def test_one_read_pass1_partial_tiles_in_z_slabs(clamped):
    """z slabs (two and three emulated ranks, the early exchange whose begin step the one-read pass then is) of a volume whose
    width and height are not whole tiles: every rank's result against the unsharded oracle."""
    import torch
    import recfilter_amd as rfa
    from recfilter_amd import capi
    for world, extents in ((2, None), (3, [64, 32, 64])):
        nz = sum(extents) if extents else 64 * world
        shape = (nz, 72, 300)
        ext = extents or [64] * world
        lo = [sum(ext[:r]) for r in range(world)]
        img = np.random.default_rng(29).random(shape, dtype=np.float32)
        plans = [rfa.Plan((ext[r],) + shape[1:], XYZ, clamped=clamped, path=capi.RF_PATH_TILED_FUSED, flags=capi.RF_PLAN_WALK_PASS1,
                          shard_rank=r, shard_world=world, shard_extents=extents) for r in range(world)]
        ins = [[torch.from_numpy(np.ascontiguousarray(img[lo[r]:lo[r] + ext[r]])).cuda()] for r in range(world)]
        outs = [[torch.empty_like(ins[r][0])] for r in range(world)]
        for r in range(world):
            plans[r].begin(ins[r], outs[r])
        nex = plans[0].num_exchanges
        for e in range(nex):
            nbytes = plans[0].exchange_bytes(e)
            gathered = torch.empty(world * nbytes, dtype=torch.uint8, device="cuda")
            for r in range(world):
                plans[r].exchange_local(e, gathered.data_ptr() + r * nbytes)
            for r in range(world):
                plans[r].exchange_apply(e, gathered.data_ptr())
        for r in range(world):
            plans[r].finish()
        torch.cuda.synchronize()
        assert plans[0].table("H_z").size > 0            # (the z tails' impulse responses: only the one-read pass 1 has them)
        got = np.concatenate([outs[r][0].cpu().numpy() for r in range(world)], axis=0)
        for p in plans:
            p.close()
        want = oracle.apply_filter(img.astype(np.float64), XYZ, clamped)
        assert np.abs(got - want).max() / np.abs(want).max() < 2e-6, (world, clamped)
