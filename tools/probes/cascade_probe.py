#!/usr/bin/env python3
"""In-plan cascades (more than four scans per dimension, padded 1-D signals with mixed causality) against the oracle."""
import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import recfilter_amd as rfa, oracle, ref_cases as rc
def check(shape, scans, clamped=False, dtype=np.float32, planes=1, tol=1e-4, **kw):
    imgs = [rc.random_image(shape, dtype, 7 + p) for p in range(planes)]
    with rfa.Plan(shape, scans, dtype=dtype, clamped=clamped, planes=planes, **kw) as plan:
        dev = [torch.from_numpy(im).cuda() for im in imgs]
        outs, timed = plan.execute_timed(dev)
        names = [n for n, _ in timed]
        tot = sum(v for _, v in timed)
        outs = [o.cpu().numpy() for o in outs]
        path = plan.path_name
    worst = 0.0
    for im, o in zip(imgs, outs):
        if np.issubdtype(np.dtype(dtype), np.integer):
            assert np.array_equal(o, oracle.apply_filter(im, scans, clamped))
        else:
            x = im.astype(np.float64)
            if "prologue" in kw: x = kw["prologue"][0] * x + kw["prologue"][1]
            want = oracle.apply_filter(x, scans, clamped)
            if "epilogue" in kw: want = kw["epilogue"][0] * want + kw["epilogue"][1] * x + kw["epilogue"][2]
            worst = max(worst, rc.rel_err(o, want))
    print(shape, len(scans), "scans", path, len(names), "kernels", f"{tot*1e3:.1f} us", f"err {worst:.2e}", [n for n in names if n.startswith("stage")][:2])
    assert worst < tol
bq = [0.05, 1.6, -0.7]
check((1_000_000,), [(0, True, bq)] * 5)
check((1_000_000,), [(0, True, bq)] * 9)
check((100_000,), [(0, True, bq), (0, False, bq)])                      # padded 1-D, anticausal after causal
check((123_456,), [(0, True, bq), (0, False, bq), (0, True, [0.5, 0.5]), (0, False, [0.5, 0.5])])
check((300, 1024), [(0, True, [0.5, 0.5])] * 3 + [(0, False, [0.5, 0.5])] * 3 + [(1, True, [0.6, 0.4])] * 2, clamped=True)
check((300, 1024), [(0, True, [1.0, 1.0])] * 5 + [(1, True, [1.0, 1.0])], dtype=np.int32)
check((64, 96, 512), [(0, True, [0.5, 0.5])] * 5 + [(2, True, [0.6, 0.4])] * 2, planes=1)
check((256, 512), [(0, True, [0.5, 0.5])] * 6 + [(1, False, [0.6, 0.4])], planes=3, clamped=True)
check((256, 2048), [(0, True, [0.3, 0.9, -0.5, 0.2, 0.05, -0.02])] * 2 + [(1, True, [0.5, 0.5])])      # two order-6 scans: sections need two stages
check((256, 512), [(0, True, [0.5, 0.5])] * 5, prologue=(2.0, 0.5), epilogue=(0.5, 0.0, 1.0))
print("ok")
