"""Committed golden vectors (tests/golden/vectors.npz, written by tests/golden/make_golden.py): the oracle must keep
reproducing them on the CPU, and the HIP path (through the C ABI) must reproduce them on the GPU."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden as mg
import oracle
import ref_cases as rc

VECTORS = np.load(os.path.join(HERE, "golden", "vectors.npz"))


@pytest.mark.parametrize("name", sorted(mg.CASES))
def test_oracle_reproduces_golden_vectors(name):
    shape, dtype, scans, clamped, seed = mg.CASES[name]
    img, want = VECTORS[name + "/input"], VECTORS[name + "/expected"]
    assert img.shape == tuple(shape) and img.dtype == np.dtype(dtype)
    np.testing.assert_array_equal(img, rc.random_image(shape, dtype, seed))       # the generator is part of the fixture
    if np.issubdtype(dtype, np.integer):
        np.testing.assert_array_equal(oracle.apply_filter(img, scans, clamped), want)
    else:
        got = oracle.apply_filter(img.astype(np.float64), scans, clamped)
        assert rc.rel_err(got, want.astype(np.float64)) < 1e-6


def test_survey_anchor_is_in_the_fixture():
    first, last, centre, total = rc.CFG3_RANDOM64          # SURVEY.md 8(c), untiled f32 run of the reference's operator
    want = VECTORS["cfg3_gauss2_xy_64/expected"]
    assert abs(want[0, 0] - first) < 2e-6 and abs(want[-1, -1] - last) < 2e-6 and abs(want[32, 32] - centre) < 2e-6
    assert abs(float(want.sum(dtype=np.float64)) - total) < 2e-2


@pytest.mark.gpu
@pytest.mark.parametrize("path", [0, 1, 2], ids=["auto", "untiled", "tiled_generic"])
@pytest.mark.parametrize("name", sorted(mg.CASES))
def test_gpu_reproduces_golden_vectors(name, path):
    import torch
    import recfilter_amd as rfa
    shape, dtype, scans, clamped, seed = mg.CASES[name]
    img, want = VECTORS[name + "/input"], VECTORS[name + "/expected"]
    with rfa.Plan(shape, scans, dtype=dtype, clamped=clamped, path=path) as plan:
        out = plan.execute([torch.from_numpy(img).cuda()])[0].cpu().numpy()
    if np.issubdtype(dtype, np.integer):
        np.testing.assert_array_equal(out, want)
    else:
        assert rc.rel_err(out, want.astype(np.float64)) < 1e-4
