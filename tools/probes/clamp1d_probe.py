#!/usr/bin/env python3
"""Clamped 1-D signals: the fused plan with border corrections against the generic path (what they ran on before round 3)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import recfilter_amd as rfa, ref_cases as rc

bq = [0.05, 1.6, -0.7]
for n in (1_000_000, 10_000_000):
    x = torch.rand((n,), device="cuda"); y = torch.empty_like(x)
    for name, scans in (("1 biquad", [(0, True, bq)]), ("causal + anticausal", [(0, True, rc.GAUSS2), (0, False, rc.GAUSS2)]),
                        ("4 biquads", [(0, True, bq)] * 4)):
        res = []
        for path in (0, 2):
            with rfa.Plan((n,), scans, clamped=True, path=path) as p:
                for _ in range(3): p.execute([x], [y])
                torch.cuda.synchronize(); t0 = time.perf_counter()
                reps = 20 if path == 0 else 3
                for _ in range(reps): p.execute([x], [y])
                torch.cuda.synchronize()
                res.append(f"{p.path_name}: {(time.perf_counter() - t0) / reps * 1e3:.3f} ms ({p.num_kernels} launches)")
        print(n, name, " | ".join(res), flush=True)
