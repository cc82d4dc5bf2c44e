#!/usr/bin/env python3
"""Orders 4..6 with a CLAMPED border (f32): the plan's sections behind border modifications on the fused kernels against the
scans as given (RF_PLAN_NO_SECTIONS: the generic path).  16384^2 and 4096^2, x/y causal + anticausal."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import recfilter_amd as rfa
from recfilter_amd import capi

def from_poles(poles, b=0.25):
    p = np.poly(poles).real
    return [b] + [float(-v) for v in p[1:]]
cases = {"order4": [0.7, 0.6, 0.3 + 0.5j, 0.3 - 0.5j], "order5": [0.8, 0.5 + 0.3j, 0.5 - 0.3j, -0.2 + 0.6j, -0.2 - 0.6j],
         "order6": [0.85, 0.1, 0.4 + 0.4j, 0.4 - 0.4j, -0.5 + 0.2j, -0.5 - 0.2j]}
for n in (4096, 16384):
    img = torch.rand((n, n), device="cuda"); out = torch.empty_like(img)
    for name, poles in cases.items():
        co = from_poles(poles)
        scans = [(0, True, co), (0, False, co), (1, True, co), (1, False, co)]
        row = []
        for label, flags in (("sections", capi.RF_PLAN_TILED_ONLY), ("as given", capi.RF_PLAN_TILED_ONLY | capi.RF_PLAN_NO_SECTIONS)):
            with rfa.Plan((n, n), scans, clamped=True, flags=flags) as plan:
                for _ in range(3): plan.execute([img], [out])
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10): plan.execute([img], [out])
                e1.record(); torch.cuda.synchronize()
                row.append(f"{label}: {plan.path_name} {e0.elapsed_time(e1) / 10:.3f} ms")
        print(f"{n}^2 {name} clamped  " + "   ".join(row), flush=True)
