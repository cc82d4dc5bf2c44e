#!/usr/bin/env python3
"""What each phase of the final pass costs when the other is absent: cfg3's filter, its x scans alone, its y scans alone, no
scans at all in one dimension (16384^2 f32, clamped).  Per-kernel HIP-event times of the plan's steps."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import recfilter_amd as rfa
import ref_cases as rc
from recfilter_amd import capi
n = 16384
img = torch.rand((n, n), device="cuda"); out = torch.empty_like(img)
for order, co in ((2, rc.GAUSS2), (3, rc.GAUSS3), (1, [1.3, -0.3])):
    cases = {"x+ x- y+ y-": rc.xy_pm(co), "x+ x-": [(0, True, co), (0, False, co)], "y+ y-": [(1, True, co), (1, False, co)],
             "x+ y+": [(0, True, co), (1, True, co)], "x+": [(0, True, co)], "y+": [(1, True, co)]}
    for name, scans in cases.items():
        with rfa.Plan((n, n), scans, clamped=True, flags=capi.RF_PLAN_TILED_ONLY | capi.RF_PLAN_TILE_ROWS(128)) as plan:
            for _ in range(5): plan.execute([img], [out])
            acc = {}
            for _ in range(15):
                _, tm = plan.execute_timed([img], [out])
                for k, v in tm: acc.setdefault(k, []).append(v)
            res = {k: round(float(np.median(v)) * 1e3, 1) for k, v in acc.items()}
            print(f"order {order}  {name:12s} tiles {list(plan.tiles)}  " + "  ".join(f"{k}={v}" for k, v in res.items()), flush=True)
