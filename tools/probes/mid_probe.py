#!/usr/bin/env python3
"""Per-kernel times of the order-3 Gaussian (gaussian_3xy of the reference's sweep) at mid sizes (tuning aid).
usage: mid_probe.py [order] sizes..."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import recfilter_amd as rfa
import ref_cases as rc
order = int(sys.argv[1]) if len(sys.argv) > 1 else 3
coeff = {1: rc.BICUBIC_COEFF, 2: rc.GAUSS2, 3: rc.GAUSS3}[order]
for n in [int(a) for a in sys.argv[2:]] or [2048, 2112]:
    scans = rc.xy_pm(coeff)
    with rfa.Plan((n, n), scans, clamped=True, path=3) as plan:
        img = torch.rand((n, n), device="cuda"); out = torch.empty_like(img)
        for _ in range(50): plan.execute([img], [out])
        acc = {}
        for _ in range(50):
            _, timed = plan.execute_timed([img], [out])
            for k, v in timed: acc.setdefault(k, []).append(v)
        tot = sum(np.mean(v) for v in acc.values())
        print(n, plan.tiles, f"total {1e3*tot:.1f} us:", " ".join(f"{k}={1e3*np.mean(v):.1f}" for k, v in acc.items()))
