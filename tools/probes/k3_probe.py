#!/usr/bin/env python3
"""Order-3 filters on the fused path (tuning aid): cfg4b and the single-plane gaussian_3xy of the reference, per kernel."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import recfilter_amd as rfa
import ref_cases as rc

w3 = list(rfa.gaussian_weights(5.0, 3))
scans = rc.xy_pm(w3)
for n, planes in ((16384, 3), (16384, 1), (4096, 1)):
    imgs = [torch.rand((n, n), device="cuda") for _ in range(planes)]
    outs = [torch.empty_like(i) for i in imgs]
    plan = rfa.Plan((n, n), scans, clamped=True, planes=planes, path=3)
    for _ in range(3): plan.execute(imgs, outs)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): plan.execute(imgs, outs)
    e1.record(); torch.cuda.synchronize()
    _, times = plan.execute_timed(imgs, outs)
    print(f"{planes} x {n}^2 order 3: {e0.elapsed_time(e1) / 20:7.3f} ms  " + " ".join(f"{k}={v*1e3:.0f}us" for k, v in times), flush=True)
    plan.close()
