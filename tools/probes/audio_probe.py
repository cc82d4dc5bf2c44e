#!/usr/bin/env python3
"""Per-kernel times of a 1-D signal through n causal biquads (apps/audio): where a long signal's time goes (tuning aid).
usage: audio_probe.py [samples] [n ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import recfilter_amd as rfa
n_samples = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
for n in [int(a) for a in sys.argv[2:]] or [1, 2, 3, 4]:
    scans = [(0, True, [0.05, 1.6, -0.7])] * n
    with rfa.Plan((n_samples,), scans) as plan:
        x = torch.rand((n_samples,), device="cuda"); out = torch.empty_like(x)
        for _ in range(30): plan.execute([x], [out])
        acc = {}
        for _ in range(30):
            _, timed = plan.execute_timed([x], [out])
            for k, v in timed: acc.setdefault(k, []).append(v)
        tot = sum(np.mean(v) for v in acc.values())
        print(n, plan.path_name, plan.tiles, f"total {1e3*tot:.1f} us:", " ".join(f"{k}={1e3*np.mean(v):.1f}" for k, v in acc.items()))
