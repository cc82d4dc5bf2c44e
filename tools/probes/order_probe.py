#!/usr/bin/env python3
"""Orders 4..6 (zero border, f32): the plan's split into sections on the fused kernels against the scans as given
(`python tools/order_probe.py given`: RF_PLAN_NO_SECTIONS, generic path), 4096^2 and 16384^2."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import recfilter_amd as rfa

def from_poles(poles, b=0.3):
    p = np.poly(poles).real
    return [b] + [float(-v) for v in p[1:]]

cases = {"order 4": [0.7, 0.6, 0.3 + 0.5j, 0.3 - 0.5j], "order 5": [0.8, 0.5 + 0.3j, 0.5 - 0.3j, -0.2 + 0.6j, -0.2 - 0.6j],
         "order 6": [0.85, 0.1, 0.4 + 0.4j, 0.4 - 0.4j, -0.5 + 0.2j, -0.5 - 0.2j]}
for n in (4096, 16384):
    img = torch.rand((n, n), device="cuda"); out = torch.empty_like(img)
    for name, poles in cases.items():
        co = from_poles(poles)
        scans = [(0, True, co), (0, False, co), (1, True, co), (1, False, co)]
        plan = rfa.Plan((n, n), scans, clamped=False, flags=rfa.capi.RF_PLAN_NO_SECTIONS if "given" in sys.argv[1:] else 0)
        for _ in range(3): plan.execute([img], [out])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): plan.execute([img], [out])
        e1.record(); torch.cuda.synchronize()
        print(f"{n}^2 {name}: {e0.elapsed_time(e1) / 10:8.3f} ms  path={plan.path_name} tiles={list(plan.tiles)}", flush=True)
        plan.close()
