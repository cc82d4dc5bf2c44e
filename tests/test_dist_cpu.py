"""world_size-2/3 gloo tests (CPU) of the sharded driver recfilter_amd.dist.ShardedFilter.

The driver, its buffer handling and the exchange protocol are the product's; the per-slab engine is the
numpy stand-in of tests/dist_engine.py (same algebra as kernels_generic.hip, tables from the product's
host-side plan).  The result must equal the untiled oracle run on the UN-sharded image."""
import os
import socket
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, case, result_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    import torch
    import torch.distributed as dist
    import oracle
    import ref_cases as rc
    from dist_engine import NumpySlabEngine
    from recfilter_amd.dist import ShardedFilter

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        shape, scans, clamped, planes, tile = case["shape"], case["scans"], case["clamped"], case["planes"], case["tile"]
        full = [rc.random_image(shape, np.float32, 77 + p) for p in range(planes)]
        extents = case.get("extents")              # slabs of different extents (rf_filter_desc.shard_extents)
        if extents is None:
            extents = [shape[0] // world] * world
        lo = sum(extents[:rank])
        n = extents[rank]
        local_shape = (n,) + tuple(shape[1:])
        inputs = [torch.from_numpy(np.ascontiguousarray(f[lo:lo + n])) for f in full]
        outputs = [torch.empty_like(t) for t in inputs]
        engine = NumpySlabEngine(local_shape, scans, clamped, planes, rank, world, tile=tile,
                                 slab_extents=extents if case.get("extents") else None, early=case.get("early", False),
                                 force_exchange=case.get("force_exchange", False))
        filt = ShardedFilter(local_shape, scans, clamped=clamped, planes=planes, rank=rank, world=world, engine=engine,
                             force_exchange=case.get("force_exchange", False))
        assert engine.has_interior == bool(case.get("early", False))
        filt.execute(inputs, outputs)
        filt.execute(inputs, outputs)          # a second execute reuses the exchange buffers
        for p in range(planes):
            want = oracle.apply_filter(full[p].astype(np.float64), scans, clamped)[lo:lo + n]
            err = rc.rel_err(outputs[p].numpy(), want)
            assert err < 1e-5, f"rank {rank} plane {p}: rel err {err}"
        n_outer = sum(1 for s in scans if s[0] == len(shape) - 1)
        assert engine.num_exchanges == (1 if engine.merged else n_outer)
        assert engine.merged == ((world > 1 or case.get("force_exchange", False)) and 1 <= n_outer <= 4)       # every case here has order <= 3
        if case.get("extents"):
            assert engine.tiles[len(shape) - 1] == case["tile"][len(shape) - 1]     # every rank tiles alike
        open(os.path.join(result_dir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


CASES = {
    "gauss2_xy_clamped": dict(shape=(48, 40), planes=1, clamped=True, tile=[8, 8]),
    "generic_xy_zero_2planes": dict(shape=(32, 24), planes=2, clamped=False, tile=[4, 4]),
    "generic_xyz": dict(shape=(24, 8, 12), planes=1, clamped=False, tile=[4, 4, 4]),
    "y_only_mixed": dict(shape=(36, 16), planes=1, clamped=True, tile=[0, 6]),
}


def _scans(name):
    import ref_cases as rc
    if name == "gauss2_xy_clamped":
        return rc.xy_pm(rc.GAUSS2)
    if name == "generic_xy_zero_2planes":
        return rc.REFERENCE_TESTS["test_generic_xy"]["scans"]
    if name == "generic_xyz":
        return rc.REFERENCE_TESTS["test_generic_xyz"]["scans"]
    return [(1, False, [0.6, 0.5, -0.1]), (1, True, [0.6, 0.5, -0.1]), (1, False, [1.0, 0.25])]


@pytest.mark.parametrize("world", [2, 3, 4])
@pytest.mark.parametrize("name", sorted(CASES))
def test_sharded_filter_over_gloo(name, world, tmp_path):
    import torch.multiprocessing as mp
    case = dict(CASES[name])
    case["scans"] = _scans(name)
    if case["shape"][0] % world or (case["shape"][0] // world) % max(case["tile"][len(case["shape"]) - 1], 1):
        pytest.skip("slab is not a whole number of tiles")
    port = _free_port()
    mp.spawn(_worker, args=(world, port, case, str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / f"ok{r}") for r in range(world))


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("name,clamped", [("generic_xyz", False), ("generic_xyz", True), ("gauss2_xy_clamped", True)])
def test_early_exchange_over_gloo(name, clamped, world, tmp_path):
    """The early exchange of a sharded outermost dimension (recfilter_amd/csrc/plan_strided.h; what a z-sharded volume
    does on the GPU): the slabs exchange the carries of their RAW data, filter the inner dimensions beside the
    all-gather (ShardedFilter calls `interior` between issuing the collective and waiting for it), and filter the
    completed carry planes along the inner dimensions afterwards.  Same result as the oracle on the whole image -- the
    operators of the outermost dimension commute with the filter of the inner ones, clamped borders included."""
    import torch.multiprocessing as mp
    case = dict(CASES[name])
    case["scans"], case["clamped"], case["early"] = _scans(name), clamped, True
    port = _free_port()
    mp.spawn(_worker, args=(world, port, case, str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / f"ok{r}") for r in range(world))


@pytest.mark.parametrize("early", [False, True])
def test_one_rank_with_forced_exchange_over_gloo(early, tmp_path):
    """world == 1 with the exchange structure forced (RF_PLAN_FORCE_EXCHANGE): the one rank still runs begin /
    exchange_local / all-gather / interior / exchange_apply / finish; the all-gather of one rank is the identity."""
    import torch.multiprocessing as mp
    case = dict(CASES["generic_xyz"])
    case["scans"], case["early"], case["force_exchange"] = _scans("generic_xyz"), early, True
    port = _free_port()
    mp.spawn(_worker, args=(1, port, case, str(tmp_path)), nprocs=1, join=True)
    assert os.path.exists(tmp_path / "ok0")


UNEQUAL = {       # name -> (base case, {world: slab extents}): whole tiles, different counts per rank
    "gauss2_xy_clamped": {2: [32, 16], 3: [24, 8, 16], 4: [8, 16, 8, 16]},
    "generic_xyz": {2: [16, 8], 3: [4, 12, 8], 4: [4, 8, 4, 8]},
    "y_only_mixed": {2: [24, 12], 3: [6, 18, 12], 4: [6, 12, 6, 12]},
    "y_five_scans": {2: [24, 12], 3: [6, 18, 12]},          # more than four scans: one exchange per scan, A^(tiles of a slab)
}


@pytest.mark.parametrize("name,world", [(n, w) for n in sorted(UNEQUAL) for w in sorted(UNEQUAL[n])])
def test_sharded_filter_unequal_slabs_over_gloo(name, world, tmp_path):
    """Slabs of different extents (rf_filter_desc.shard_extents): the tile width comes from their common divisor, the
    exit transfer of every slab from its own tile count."""
    import torch.multiprocessing as mp
    if name == "y_five_scans":
        case = dict(shape=(36, 16), planes=1, clamped=True, tile=[0, 6])
        case["scans"] = [(1, bool(i % 2), [0.5, 0.4 - 0.05 * i]) for i in range(5)]
    else:
        case = dict(CASES[name])
        case["scans"] = _scans(name)
    case["extents"] = UNEQUAL[name][world]
    assert sum(case["extents"]) == case["shape"][0]
    port = _free_port()
    mp.spawn(_worker, args=(world, port, case, str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / f"ok{r}") for r in range(world))


def test_split_extent_whole_tiles():
    from recfilter_amd.dist import split_extent
    assert split_extent(16384, 8) == [2048] * 8
    assert split_extent(64 * 11, 4) == [192, 192, 192, 128]
    assert split_extent(96, 3, granule=32) == [32, 32, 32]
    with pytest.raises(ValueError):
        split_extent(100, 2)
    with pytest.raises(ValueError):
        split_extent(64, 2)


def test_single_rank_driver_is_plain_execute(tmp_path):
    sys.path.insert(0, HERE)
    import torch
    import oracle
    import ref_cases as rc
    from dist_engine import NumpySlabEngine
    from recfilter_amd.dist import ShardedFilter
    scans = rc.xy_pm(rc.GAUSS2)
    img = rc.random_image((32, 24))
    eng = NumpySlabEngine((32, 24), scans, True, 1, 0, 1, tile=[8, 8])
    out = [torch.empty(32, 24)]
    ShardedFilter((32, 24), scans, clamped=True, engine=eng).execute([torch.from_numpy(img)], out)
    assert rc.rel_err(out[0].numpy(), oracle.apply_filter(img.astype(np.float64), scans, True)) < 1e-5
