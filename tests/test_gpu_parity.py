"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs."""
import numpy as np
import pytest

import oracle
import ref_cases as rc

pytestmark = pytest.mark.gpu

TOL = 1e-4     # north_star: 1e-4 relative for floating point; integers bit-exact


def _run(shape, scans, dtype=np.float32, clamped=False, planes=1, tile=None, path=0, seed=1234, inplace=False):
    import torch
    import recfilter_amd as rfa
    imgs = [rc.random_image(shape, dtype, seed + i) for i in range(planes)]
    dev = [torch.from_numpy(im).cuda() for im in imgs]
    with rfa.Plan(shape, scans, dtype=dtype, clamped=clamped, planes=planes, tile=tile, path=path) as plan:
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


@pytest.mark.parametrize("path", [1, 2], ids=["untiled", "tiled_generic"])
@pytest.mark.parametrize("name", sorted(rc.REFERENCE_TESTS))
def test_reference_tests(name, path):
    """Every configuration of /root/reference/tests/test_*.cpp, literal shapes and tile widths."""
    case = rc.REFERENCE_TESTS[name]
    nd = len(case["shape"])
    tile = [case["tile"] if any(s[0] == d for s in case["scans"]) else 0 for d in range(nd)]
    imgs, outs, (got_path, tiles) = _run(case["shape"], case["scans"], case["dtype"], case["clamped"],
                                         tile=tile if path == 2 else None, path=path)
    assert got_path == path
    if path == 2:
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


def test_timed_execute_reports_every_kernel():
    import torch
    import recfilter_amd as rfa
    scans = rc.xy_pm(rc.GAUSS2)
    img = torch.from_numpy(rc.random_image((128, 128))).cuda()
    with rfa.Plan((128, 128), scans, clamped=True, path=2) as plan:
        outs, times = plan.execute_timed([img])
        assert len(times) == plan.num_kernels == 2 * 4   # per dimension: pass1, carry x2, pass2
        assert all(ms >= 0 for _, ms in times)
