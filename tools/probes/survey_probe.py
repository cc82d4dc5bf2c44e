#!/usr/bin/env python3
"""Survey of common image sizes x filters x planes on the automatic path: ms per filter, algorithmic GB/s (8 B per f32 sample)
and the per-kernel times -- to spot cliffs away from the headline shape.  python tools/survey_probe.py"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import recfilter_amd as rfa, ref_cases as rc

FILTERS = {"sat": ([(0, True, [1.0, 1.0]), (1, True, [1.0, 1.0])], False), "gauss2": (rc.xy_pm(rc.GAUSS2), True), "gauss3": (rc.xy_pm(rc.GAUSS3), True)}
SIZES = [(1080, 1920), (1440, 2560), (2160, 3840), (4096, 4096), (5000, 5000), (4000, 6000), (6000, 8000), (8192, 8192), (9000, 12000), (16384, 16384)]
for planes in (1, 3):
    for name, (scans, clamped) in FILTERS.items():
        for shape in SIZES:
            if planes * shape[0] * shape[1] * 8 > 20e9:
                continue
            xs = [torch.rand(shape, device="cuda") for _ in range(planes)]; ys = [torch.empty_like(x) for x in xs]
            with rfa.Plan(shape, scans, clamped=clamped, planes=planes) as p:
                for _ in range(5): p.execute(xs, ys)
                acc = {}
                for _ in range(6):
                    _, tm = p.execute_timed(xs, ys)
                    for k, v in tm: acc.setdefault(k, []).append(v)
                reps = 30 if shape[0] * shape[1] < 3e7 else 10
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(reps): p.execute(xs, ys)
                torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / reps * 1e3
                px = shape[0] * shape[1] * planes
                print(f"x{planes} {name:6s} {str(shape):15s} {p.path_name:11s} tiles={list(p.tiles)} {ms:8.4f} ms {8 * px / ms / 1e6:6.0f} GB/s ",
                      {k: round(float(np.median(v)), 4) for k, v in acc.items()}, flush=True)
            del xs, ys
