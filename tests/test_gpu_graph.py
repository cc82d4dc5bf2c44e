"""rf_plan_execute is capturable in a HIP graph: no allocation, no host synchronisation, every launch on the stream it is
given (include/recfilter_amd.h; the task's launch-bound loops -- small images, the four launches of a 1-D high-order scan --
are what a caller would capture).  A replayed graph must give what a direct execute gives, bit for bit."""
import numpy as np
import pytest

import ref_cases as rc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", ["fused_2d", "matrix_1d_order_13", "matrix_2d_order_12", "walk_3d", "overlapped_2d"])
def test_execute_captured_in_a_graph_replays_bit_identically(case):
    import torch
    import recfilter_amd as rfa
    from recfilter_amd import capi
    kw = {}
    if case == "fused_2d":
        shape, scans, clamped, path = (1024, 1024), rc.xy_pm(rc.GAUSS2), True, capi.RF_PATH_TILED_FUSED
    elif case == "matrix_1d_order_13":
        shape, scans, clamped, path = (1 << 20,), [(0, True, [1.0] + [0.01] * 13)], False, capi.RF_PATH_TILED_MATRIX
    elif case == "matrix_2d_order_12":
        a = np.random.default_rng(5).standard_normal(12) * np.exp(-0.15 * np.arange(12))
        c = [0.4] + [float(np.float32(v)) for v in a * 0.85 / np.abs(a).sum()]
        shape, scans, clamped, path = (512, 1024), [(0, True, c), (0, False, c), (1, True, c), (1, False, c)], True, capi.RF_PATH_TILED_MATRIX
    elif case == "walk_3d":
        shape, scans, clamped, path = (64, 64, 512), rc.REFERENCE_TESTS["test_generic_xyz"]["scans"], False, capi.RF_PATH_TILED_FUSED
        kw["flags"] = capi.RF_PLAN_WALK_PASS1
    else:
        shape, scans, clamped, path = (256, 256), rc.REFERENCE_TESTS["test_generic_xy"]["scans"], False, capi.RF_PATH_TILED_OVERLAPPED
        kw["tile"] = [32, 32]
    x = torch.from_numpy(rc.random_image(shape, np.float32, 41)).cuda()
    ref, out = torch.empty_like(x), torch.empty_like(x)
    with rfa.Plan(shape, scans, clamped=clamped, path=path, **kw) as plan:
        assert plan.path == path
        plan.execute([x], [ref])
        torch.cuda.synchronize()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            plan.execute([x], [out])                  # (first launches on this stream: one-time kernel attributes are set here)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            plan.execute([x], [out])
        for _ in range(3):
            out.zero_()
            g.replay()
            torch.cuda.synchronize()
            assert torch.equal(out, ref)
        del g
