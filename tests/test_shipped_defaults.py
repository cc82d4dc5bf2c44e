"""The defaults a user gets (rf_filter_desc.flags = 0, RF_PATH_AUTO): the small shapes the rest of the suite pins to the
TILED kernels (tests/conftest.py: RF_PLAN_TILED_ONLY) run here as shipped -- images up to 1024^2 on the line-parallel
untiled kernels, pointwise stages on the tiled passes they are fused into, cascades whose stages choose their own path --
against the oracle, whatever path the plan resolves to."""
import numpy as np
import pytest

import oracle
import ref_cases as rc

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("shipped_defaults")]

TOL = 1e-4


def _run(shape, scans, dtype=np.float32, clamped=False, planes=1, tile=None, seed=1234, inplace=False, **kw):
    import torch
    import recfilter_amd as rfa
    import recfilter_amd.plan as rp
    assert rp.DEFAULT_FLAGS == 0
    if np.issubdtype(np.dtype(dtype), np.integer):
        imgs = [np.random.default_rng(seed + i).integers(0, 4, size=shape).astype(dtype) for i in range(planes)]
    else:
        imgs = [rc.random_image(shape, dtype, seed + i) for i in range(planes)]
    dev = [torch.from_numpy(im).cuda() for im in imgs]
    with rfa.Plan(shape, scans, dtype=dtype, clamped=clamped, planes=planes, tile=tile, **kw) as plan:
        outs = plan.execute(dev, dev if inplace else None)
        torch.cuda.synchronize()
        path = plan.path_name
    return imgs, [o.cpu().numpy() for o in outs], path


def _check(imgs, outs, scans, clamped):
    for im, out in zip(imgs, outs):
        if np.issubdtype(im.dtype, np.integer):
            np.testing.assert_array_equal(out, oracle.apply_filter(im, scans, clamped))
        else:
            err = rc.rel_err(out, oracle.apply_filter(im.astype(np.float64), scans, clamped))
            assert err < TOL, f"rel err {err}"


@pytest.mark.parametrize("name", sorted(rc.FUSED_CASES))
def test_fused_cases_as_shipped(name):
    case = rc.FUSED_CASES[name]
    imgs, outs, _ = _run(case["shape"], case["scans"], clamped=case["clamped"])
    _check(imgs, outs, case["scans"], case["clamped"])


@pytest.mark.parametrize("name", sorted(rc.REFERENCE_TESTS))
def test_reference_tests_as_shipped(name):
    """Every tests/test_*.cpp configuration of the reference with its own split() widths, automatic path."""
    t = rc.REFERENCE_TESTS[name]
    # (split() along the dimensions the tile divides: the 20 x 1 tests are tiled along x only)
    tile = [t["tile"] if n % t["tile"] == 0 else 0 for n in reversed(t["shape"])] if "tile" in t else None
    imgs, outs, _ = _run(t["shape"], t["scans"], dtype=t.get("dtype", np.float32), clamped=t.get("clamped", False), tile=tile)
    _check(imgs, outs, t["scans"], t.get("clamped", False))


@pytest.mark.parametrize("dtype", [np.float32, np.float64, np.int32, np.int16])
@pytest.mark.parametrize("shape", [(512, 512), (192, 320), (1024, 768), (75, 464), (48, 64, 256)])
def test_pixel_types_planes_and_in_place_as_shipped(dtype, shape):
    integer = np.issubdtype(np.dtype(dtype), np.integer)
    if len(shape) == 3:
        scans = ([(0, True, [1.0, 1.0]), (1, True, [1.0, 2.0, -1.0]), (2, False, [1.0, 1.0])] if integer
                 else rc.BASELINE_CONFIGS["cfg5_generic_xyz"]["scans"])
    else:
        scans = [(0, True, [1.0, 1.0]), (0, False, [1.0, 1.0, -1.0]), (1, True, [1.0, 2.0, -1.0])] if integer else rc.xy_pm(rc.GAUSS3)
    clamped = not integer and len(shape) == 2
    imgs, outs, path = _run(shape, scans, dtype=dtype, clamped=clamped, planes=3)
    _check(imgs, outs, scans, clamped)
    imgs, outs, path2 = _run(shape, scans, dtype=dtype, clamped=clamped, planes=2, inplace=True, seed=77)
    assert path2 == path
    _check(imgs, outs, scans, clamped)


def test_small_images_take_the_line_kernels_and_pointwise_stages_the_tiled_passes():
    import torch
    import recfilter_amd as rfa
    scans = rc.xy_pm(rc.GAUSS2)
    imgs, outs, path = _run((512, 512), scans, clamped=True)
    assert path == "untiled"
    _check(imgs, outs, scans, True)
    # uint8 input + prologue + unsharp-mask epilogue: fused into the tiled passes, as shipped
    img8 = rc.cuda_image((300, 768), np.uint8, 84)
    w = 0.7
    with rfa.Plan((300, 768), scans, clamped=True, prologue=(1.0 / 255.0, 0.0), epilogue=(-w, 1.0 + w, 0.0), input_dtype=np.uint8) as plan:
        assert plan.path_name == "tiled_fused"
        out = plan.execute([img8])[0].cpu().numpy()
    x = img8.cpu().numpy().astype(np.float64) / 255.0
    assert np.abs(out - ((1.0 + w) * x - w * oracle.apply_filter(x, scans, True))).max() < 2e-5


@pytest.mark.parametrize("shape", [(256, 512), (100_000,), (1_000_000,), (640, 1280)])
def test_cascades_as_shipped(shape):
    """More than four scans per dimension: an in-plan cascade whose stages choose their own path under the defaults
    (line kernels for the small 2-D image, fused stages otherwise)."""
    bq = [0.05, 1.6, -0.7]
    if len(shape) == 1:
        scans = [(0, True, bq)] * 3 + [(0, False, bq)] * 3
    else:
        scans = [(0, True, [0.5, 0.5])] * 6 + [(1, False, [0.6, 0.4])] + [(1, True, [0.3, 0.5, 0.1])] * 5
    imgs, outs, _ = _run(shape, scans, planes=2 if len(shape) == 2 else 1)
    _check(imgs, outs, scans, False)


def test_front_end_as_shipped():
    """The RecFilter front-end (recfilter_amd/filter.py) on the reference's test_trivial shape and a Gaussian cascade."""
    import torch
    import recfilter_amd as rfa
    img = rc.random_image((256, 384), np.float32, 5)
    x, y = rfa.RecFilterDim("x", 384), rfa.RecFilterDim("y", 256)
    f = rfa.RecFilter("F")
    f.set_clamped_image_border()
    f[x, y] = torch.from_numpy(img).cuda()
    w = rfa.gaussian_weights(5.0, 3)
    for d in (+x, -x, +y, -y):
        f.add_filter(d, w)
    parts = f.cascade_by_dimension()
    for p in parts:
        p.split_all_dimensions(32)
    out = parts[-1].realize()[0].cpu().numpy()
    scans = [(0, True, w), (0, False, w), (1, True, w), (1, False, w)]
    assert rc.rel_err(out, oracle.apply_filter(img.astype(np.float64), scans, True)) < TOL
