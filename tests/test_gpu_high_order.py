"""Scans of order 9..32 in their direct form through the C ABI (RecFilter::add_filter takes any order,
/root/reference/lib/recfilter.cpp:260-343; the reference's own sweep runs orders 1, 3, .. 29,
apps/audio/audio_filter_high_order.cpp:14,38-42) against the CPU oracle."""
import numpy as np
import pytest

import oracle
import ref_cases as rc

pytestmark = pytest.mark.gpu

from recfilter_amd import capi

TOL = 1e-4     # north_star: 1e-4 relative for floating point; integers bit-exact


def stable_coeff(order, seed, b=0.4, mass=0.85):
    """[b, a1..a_order] with sum |a| = mass < 1 (bounded-input bounded-output whatever the signs)."""
    rng = np.random.default_rng(seed)
    a = rng.standard_normal(order) * np.exp(-0.15 * np.arange(order))
    a *= mass / np.abs(a).sum()
    return [b] + [float(np.float32(v)) for v in a]


def audio_coeff(order):
    return [1.0] + [0.01] * order        # apps/audio/audio_filter_high_order.cpp:41-42


def int_coeff(order, seed):
    rng = np.random.default_rng(seed)
    return [1.0] + [float(v) for v in rng.integers(-2, 3, order)]


def _run(shape, scans, dtype=np.float32, clamped=False, planes=1, tile=None, path=0, seed=77, inplace=False, flags=None):
    import torch
    import recfilter_amd as rfa
    imgs = [rc.random_image(shape, dtype, seed + i) for i in range(planes)]
    dev = [torch.from_numpy(im).cuda() for im in imgs]
    with rfa.Plan(shape, scans, dtype=dtype, clamped=clamped, planes=planes, tile=tile, path=path, flags=flags) as plan:
        outs = plan.execute(dev, dev if inplace else None)
        torch.cuda.synchronize()
        info = (plan.path, plan.tiles)
    return imgs, [o.cpu().numpy() for o in outs], info


def _check(imgs, outs, scans, clamped, tol=TOL):
    for im, out in zip(imgs, outs):
        if np.issubdtype(im.dtype, np.integer):
            np.testing.assert_array_equal(out, oracle.apply_filter(im, scans, clamped))
        else:
            want = oracle.apply_filter(im.astype(np.float64), scans, clamped)
            err = rc.rel_err(out, want)
            assert err < tol, f"rel err {err}"


@pytest.mark.parametrize("path", [1, 2], ids=["untiled", "tiled_generic"])
@pytest.mark.parametrize("order", [9, 12, 16, 23, 29, 32])
@pytest.mark.parametrize("clamped", [False, True])
def test_direct_form_always_correct_paths_f32(order, clamped, path):
    """The always-correct paths (one recurrence per line; per-dimension tiles with a run-time width) with a 32-deep window."""
    scans = [(0, True, stable_coeff(order, order)), (1, False, stable_coeff(order, 100 + order)), (0, False, audio_coeff(order))]
    imgs, outs, (got, tiles) = _run((96, 160), scans, clamped=clamped, path=path, tile=[32, 32] if path == 2 else None)
    assert got == path
    _check(imgs, outs, scans, clamped)


@pytest.mark.parametrize("path", [1, 2], ids=["untiled", "tiled_generic"])
@pytest.mark.parametrize("dtype", [np.float64, np.int32, np.int16])
def test_direct_form_other_pixel_types(dtype, path):
    if np.issubdtype(dtype, np.integer):
        scans = [(0, True, int_coeff(17, 1)), (1, False, int_coeff(32, 2)), (1, True, int_coeff(9, 3))]
    else:
        scans = [(0, True, stable_coeff(17, 1)), (1, False, stable_coeff(32, 2)), (1, True, stable_coeff(9, 3))]
    for clamped in (False, True):
        imgs, outs, (got, _) = _run((64, 96), scans, dtype, clamped=clamped, planes=2, path=path, tile=[32, 32] if path == 2 else None)
        assert got == path
        _check(imgs, outs, scans, clamped, tol=1e-10 if dtype == np.float64 else TOL)


def test_direct_form_3d_and_1d_generic():
    scans = [(0, True, stable_coeff(11, 5)), (1, False, stable_coeff(13, 6)), (2, True, stable_coeff(20, 7)), (2, False, audio_coeff(29))]
    imgs, outs, (got, _) = _run((40, 64, 96), scans, path=2, tile=[32, 32, 20], inplace=True)
    assert got == 2
    _check(imgs, outs, scans, False)
    s1 = [(0, True, audio_coeff(29)), (0, False, stable_coeff(10, 8))]
    for clamped in (False, True):
        sig, out, (got, _) = _run((4096,), s1, path=2, tile=[64], clamped=clamped)
        assert got == 2
        _check(sig, out, s1, clamped)


def test_orders_above_32_are_refused_and_32_is_accepted():
    import recfilter_amd as rfa
    with pytest.raises(Exception):
        rfa.Plan((256,), [(0, True, [1.0] + [0.001] * 33)])
    with rfa.Plan((256,), [(0, True, [1.0] + [0.001] * 32)]) as plan:
        assert plan.path in (1, 2, 5)


# ---- the matrix path (RF_PATH_TILED_MATRIX, kernels_matrix.hip): every stage a GEMM on the matrix cores ----------------
MX = capi.RF_PATH_TILED_MATRIX


@pytest.mark.parametrize("clamped", [False, True])
@pytest.mark.parametrize("order", [1, 2, 5, 8, 9, 12, 16, 17, 24, 29, 32])
def test_matrix_path_1d_every_order(order, clamped):
    """One causal scan, as apps/audio/audio_filter_high_order.cpp:57-66 has it, and an anticausal one behind it."""
    scans = [(0, True, audio_coeff(order)), (0, False, stable_coeff(order, order))]
    sig, out, (path, tiles) = _run((1 << 16,), scans, clamped=clamped, path=MX)
    assert path == MX and tiles[0] == 128
    _check(sig, out, scans, clamped)


@pytest.mark.parametrize("n", [32, 64, 96, 160, 32 * 25, 32 * 1031, 128 * 4099, 1 << 22])
def test_matrix_path_1d_lengths_and_chain_levels(n):
    """Tile widths 32 / 64 / 96 / 128, sequences short enough for one chain launch and long enough for three levels,
    partial last chunks (a prime number of tiles)."""
    scans = [(0, False, stable_coeff(32, 9)), (0, True, stable_coeff(20, 10))]
    for clamped in (False, True):
        sig, out, (path, _) = _run((n,), scans, clamped=clamped, path=MX)
        assert path == MX
        _check(sig, out, scans, clamped)


@pytest.mark.parametrize("clamped", [False, True])
@pytest.mark.parametrize("shape", [(96, 160), (200, 256), (33, 128), (1024, 2048), (7, 96), (128, 32)])
def test_matrix_path_2d_order_12_x_and_y(shape, clamped):
    """The 2-D order-12 causal + anticausal x/y filter: lane = line along x (or lane = tile when the image has fewer than
    32 rows), lane = column along y; line counts that are not multiples of the 128 a workgroup takes."""
    c = stable_coeff(12, 3)
    scans = [(0, True, c), (0, False, c)]
    if shape[0] % 32 == 0:
        scans += [(1, True, c), (1, False, c)]
    imgs, outs, (path, _) = _run(shape, scans, clamped=clamped, path=MX, planes=2)
    assert path == MX
    _check(imgs, outs, scans, clamped)


def test_matrix_path_3d_inplace_and_auto():
    scans = [(2, False, stable_coeff(9, 1)), (0, True, stable_coeff(17, 2)), (1, False, stable_coeff(32, 4)), (0, False, stable_coeff(4, 5))]
    for clamped in (False, True):
        imgs, outs, (path, tiles) = _run((64, 96, 160), scans, clamped=clamped, inplace=True)
        assert path == MX and list(tiles) == [32, 96, 64]         # automatic for orders above 8
        _check(imgs, outs, scans, clamped)
    # widths whose rows are 16-byte aligned but not a multiple of 128 columns (partial column blocks along y / z)
    imgs, outs, (path, _) = _run((32, 64, 36), [(1, True, stable_coeff(10, 6)), (2, False, stable_coeff(11, 7))], clamped=True)
    assert path == MX
    _check(imgs, outs, [(1, True, stable_coeff(10, 6)), (2, False, stable_coeff(11, 7))], True)


def test_matrix_path_integrators_and_growing_filters():
    """Poles on the unit circle: a summed-area table written as one order-2 scan per direction, integer-valued weights --
    the impulse-response blocks grow linearly and every product is exact in f32 while the sums stay below 2^24."""
    scans = [(0, True, [1.0, 2.0, -1.0]), (1, True, [1.0, 1.0])]
    import torch
    import recfilter_amd as rfa
    img = np.random.default_rng(3).integers(0, 3, (64, 96)).astype(np.float32)
    with rfa.Plan(img.shape, scans, path=MX) as plan:
        out = plan.execute([torch.from_numpy(img).cuda()])[0].cpu().numpy()
    want = oracle.apply_filter(img.astype(np.float64), scans, False)
    assert want.max() < 2 ** 24
    np.testing.assert_array_equal(out, want.astype(np.float32))


@pytest.mark.parametrize("clamped", [False, True])
@pytest.mark.parametrize("shape", [(77, 300), (20,), (50, 33, 68), (1000, 1004), (130, 4100), (10_000_004,), (3, 36), (129, 128)])
def test_matrix_path_extents_no_tile_divides(shape, clamped):
    """Tiles need not divide the extent: the padding (zeros, never stored) lies where each scan leaves the image, so the tile where
    it enters is whole -- causal and anticausal scans of one dimension then tile differently.  Widths multiples of 4, any height /
    depth; shorter than a tile, shorter than the order."""
    nd = len(shape)
    scans = [(0, True, stable_coeff(12, 3)), (0, False, stable_coeff(29, 4))]
    if nd >= 2:
        scans += [(1, True, stable_coeff(9, 5)), (1, False, stable_coeff(30, 8))]
    if nd == 3:
        scans += [(2, False, stable_coeff(17, 2)), (2, True, audio_coeff(11))]
    imgs, outs, (path, tiles) = _run(shape, scans, clamped=clamped, path=MX, planes=2 if nd == 2 else 1)
    assert path == MX
    _check(imgs, outs, scans, clamped)


def _emulate_ranks(local_shape, world, scans, clamped, planes=1, path=0):
    """`world` slabs of one image through the stepping protocol on ONE device: one plan per rank, the all-gather a rank-major
    device buffer (tools/rehearse_n8.py does the same at bench size).  Returns the inputs, the concatenated result, the paths."""
    import torch
    import recfilter_amd as rfa
    global_shape = (local_shape[0] * world,) + tuple(local_shape[1:])
    rng = np.random.default_rng(31)
    whole = [torch.from_numpy(rng.random(global_shape, dtype=np.float32)).cuda() for _ in range(planes)]
    outs = [torch.empty_like(w) for w in whole]
    ins_r = [[w.split(local_shape[0])[r] for w in whole] for r in range(world)]
    outs_r = [[o.split(local_shape[0])[r] for o in outs] for r in range(world)]
    plans = [rfa.Plan(local_shape, scans, clamped=clamped, planes=planes, shard_rank=r, shard_world=world, path=path) for r in range(world)]
    for r in range(world):
        plans[r].begin(ins_r[r], outs_r[r])
    for e in range(plans[0].num_exchanges):
        nbytes = plans[0].exchange_bytes(e)
        gathered = torch.zeros(world * nbytes, dtype=torch.uint8, device="cuda")
        for r in range(world):
            plans[r].exchange_local(e, gathered.data_ptr() + r * nbytes)
        for r in range(world):
            plans[r].exchange_apply(e, gathered.data_ptr())
    for r in range(world):
        plans[r].finish()
    torch.cuda.synchronize()
    paths = [p.path for p in plans]
    n_ex = plans[0].num_exchanges
    for p in plans:
        p.close()
    return [w.cpu().numpy() for w in whole], [o.cpu().numpy() for o in outs], paths, n_ex


@pytest.mark.parametrize("clamped", [False, True])
@pytest.mark.parametrize("case", ["rows_3_ranks", "rows_8_ranks_slow_decay", "z_slabs_2_ranks", "planes", "rows_ragged_width", "z_slabs_ragged_rows",
                                  "rows_wide_one_chain"])
def test_matrix_path_sharded_slabs(case, clamped):
    """High-order filters over the GPUs of a node: slabs of the outermost dimension, one exchange of the k-row exit
    carries per scan along it -- every rank chains the gathered exits with A^M on the matrix cores, propagates its entering carry
    through its tiles with A^1 .. A^M, and the final pass takes it in the first tile.  Emulated ranks on one device against the
    oracle on the whole image."""
    planes = 1
    if case == "rows_3_ranks":
        local, world = (128, 256), 3
        scans = [(0, True, stable_coeff(9, 1)), (0, False, stable_coeff(12, 2)), (1, True, stable_coeff(12, 3)), (1, False, stable_coeff(17, 4)), (1, True, audio_coeff(29))]
    elif case == "rows_8_ranks_slow_decay":       # poles close to 1: carries that cross several slabs
        local, world = (32, 160), 8
        scans = [(1, True, [0.02, 0.98]), (1, False, [0.05, 1.6, -0.65]), (1, True, stable_coeff(9, 5, mass=0.97))]
    elif case == "z_slabs_2_ranks":
        local, world = (64, 40, 128), 2
        scans = [(2, True, stable_coeff(10, 6)), (0, False, stable_coeff(9, 7)), (1, True, stable_coeff(11, 8)), (2, False, stable_coeff(13, 9))]
    elif case == "rows_wide_one_chain":        # 16384 columns fill the chip: the 29 tiles (of 32 rows) of a slab are chained in one go (no
        local, world = (32 * 29, 16384), 2     # propagation) before the exchange; the x scans are a pair stage
        scans = [(0, True, stable_coeff(9, 22)), (0, False, stable_coeff(9, 23)), (1, True, stable_coeff(12, 24)), (1, False, stable_coeff(12, 25))]
    elif case == "rows_ragged_width":          # the slab-local x scans on tiles that do not divide the width
        local, world = (64, 300), 2
        scans = [(0, True, stable_coeff(9, 12)), (0, False, stable_coeff(14, 13)), (1, True, stable_coeff(10, 14)), (1, False, stable_coeff(10, 15))]
    elif case == "z_slabs_ragged_rows":
        local, world = (32, 50, 68), 3
        scans = [(1, True, stable_coeff(9, 16)), (1, False, stable_coeff(9, 17)), (2, False, stable_coeff(12, 18)), (0, True, stable_coeff(9, 19))]
    else:
        local, world, planes = (96, 128), 2, 3
        scans = [(1, False, stable_coeff(20, 10)), (0, True, stable_coeff(9, 11))]
    imgs, outs, paths, n_ex = _emulate_ranks(local, world, scans, clamped, planes, path=MX)
    assert all(p == MX for p in paths)
    assert n_ex == sum(1 for s in scans if s[0] == len(local) - 1)          # one exchange per scan along the sharded dimension
    for im, out in zip(imgs, outs):
        want = oracle.apply_filter(im.astype(np.float64), scans, clamped)
        assert rc.rel_err(out, want) < TOL


def test_matrix_path_with_pointwise_stages():
    """A defining expression `scale * in + bias` and a pointwise consumer `w_f * F + w_i * x' + bias` (rf_pointwise_desc) around a
    high-order filter: stand-alone elementwise launches in front of the first stage and behind the last one."""
    import torch
    import recfilter_amd as rfa
    c = stable_coeff(12, 3)
    scans = [(0, True, c), (1, False, c)]
    img = rc.random_image((96, 256), np.float32, 9)
    dev = torch.from_numpy(img).cuda()
    with rfa.Plan(img.shape, scans, clamped=True, prologue=(0.5, 0.25), epilogue=(2.0, -1.0, 0.125), path=MX) as plan:
        assert plan.path == MX
        out = plan.execute([dev])[0].cpu().numpy()
    want, scale = rc.pointwise_want(img, scans, True, (0.5, 0.25), (2.0, -1.0, 0.125))
    assert rc.rel_err(out, want, scale=scale) < TOL        # judged against the terms' magnitudes: 2 F - x' may cancel


PAIR_CASES = {
    # shape (.., y, x), scans, planes: a causal scan directly followed by an anticausal one along a dimension = one PAIR stage
    "1d_x1_levels": ((128 * 300,), [(0, True, stable_coeff(9, 31)), (0, False, stable_coeff(16, 32))], 1),       # lane = tile; 300 tiles: two chain levels
    "1d_few_lines": ((7, 96 * 20), [(0, True, stable_coeff(12, 33)), (0, False, stable_coeff(12, 34))], 1),      # tiles of 96, lane = tile of 7 lines
    "2d_xy": ((384, 640), [(0, True, stable_coeff(12, 35)), (0, False, stable_coeff(11, 36)),
                           (1, True, stable_coeff(7, 37)), (1, False, stable_coeff(5, 38))], 2),                  # tiles 128 x 128, 5 and 3 of them, two planes
    "2d_one_tile_each": ((128, 96), [(0, True, stable_coeff(8, 39)), (0, False, stable_coeff(8, 40)),
                                     (1, True, stable_coeff(16, 41)), (1, False, stable_coeff(9, 42))], 1),       # first tile = last tile
    "2d_pair_and_singles": ((256, 512), [(0, True, stable_coeff(6, 43)), (0, False, stable_coeff(6, 44)), (0, True, stable_coeff(20, 45)),
                                         (1, False, stable_coeff(4, 46)), (1, True, stable_coeff(4, 47)), (1, False, stable_coeff(3, 48))], 1),
    "3d_z_pair": ((256, 40, 64), [(2, True, stable_coeff(10, 49)), (2, False, stable_coeff(10, 50)), (0, False, stable_coeff(5, 51))], 1),
    "2d_partial_blocks": ((200, 416), [(0, True, stable_coeff(12, 52)), (0, False, stable_coeff(12, 53)),
                                       (1, True, stable_coeff(12, 54)), (1, False, stable_coeff(12, 55))], 1),    # 200 lines / 416 columns: partial unit blocks; y ragged -> single stages
}


@pytest.mark.parametrize("clamped", [False, True])
@pytest.mark.parametrize("case", sorted(PAIR_CASES))
def test_matrix_path_pair_stages(case, clamped):
    """Causal + anticausal scans of one dimension in ONE final pass (MxPassArgs::pair): pass 1 forms both scans' tails in one
    contraction, the causal carry entering a tile feeds the anticausal tails (lib/split.cpp:912-1004), the final pass walks the
    tile forward and backward with the causal result in registers."""
    import recfilter_amd as rfa
    shape, scans, planes = PAIR_CASES[case]
    imgs, outs, (path, _) = _run(shape, scans, clamped=clamped, planes=planes, path=capi.RF_PATH_TILED_MATRIX)
    assert path == capi.RF_PATH_TILED_MATRIX
    _check(imgs, outs, scans, clamped)
    # in place: a wave has read its whole tile by the time its backward walk stores the first sub-block
    imgs, outs, _ = _run(shape, scans, clamped=clamped, planes=planes, path=capi.RF_PATH_TILED_MATRIX, inplace=True)
    _check(imgs, outs, scans, clamped)
    with rfa.Plan(shape, scans, clamped=clamped, planes=planes, path=capi.RF_PATH_TILED_MATRIX, device=capi.RF_DEVICE_HOST_ONLY) as host:
        heads = 0
        for i in range(len(scans)):                         # (the plan's order: grouped by dimension)
            try:
                heads += host.table(f"mx_pair_{i}").size
            except Exception:
                pass
        assert heads >= 1


def test_matrix_path_executes_concurrently_on_distinct_streams():
    """executes of one plan on distinct streams may overlap -- every instance has its own tails, chain levels and
    (pair stages) second set of tails.  Three images through ONE plan of pair stages and a single stage on three busy streams, then
    from three host threads at once; every result against the oracle."""
    import threading
    import torch
    import recfilter_amd as rfa
    shape, scans, _ = PAIR_CASES["2d_pair_and_singles"]
    imgs = [rc.random_image(shape, np.float32, 91 + i) for i in range(3)]
    wants = [oracle.apply_filter(im.astype(np.float64), scans, True) for im in imgs]
    dev = [torch.from_numpy(im).cuda() for im in imgs]
    streams = [torch.cuda.Stream() for _ in range(3)]
    with rfa.Plan(shape, scans, clamped=True, path=capi.RF_PATH_TILED_MATRIX) as plan:
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
            assert rc.rel_err(o.cpu().numpy(), w) < TOL
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
            assert rc.rel_err(o.cpu().numpy(), w) < TOL
