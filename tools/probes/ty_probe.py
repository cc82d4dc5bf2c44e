#!/usr/bin/env python3
"""Tile height of the fused path: the automatic choice against RF_PLAN_TILE_ROWS(32 / 64 / 128) on shapes with partial tiles.
python tools/ty_probe.py"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import recfilter_amd as rfa, ref_cases as rc
from recfilter_amd import capi

def t(shape, scans, flags, planes=1):
    xs = [torch.rand(shape, device="cuda") for _ in range(planes)]; ys = [torch.empty_like(x) for x in xs]
    with rfa.Plan(shape, scans, clamped=True, planes=planes, flags=flags) as p:
        for _ in range(5): p.execute(xs, ys)
        reps = 30 if shape[0] * shape[1] < 5e7 else 12
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps): p.execute(xs, ys)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3, list(p.tiles)

for name, scans in (("gauss2", rc.xy_pm(rc.GAUSS2)), ("gauss3", rc.xy_pm(rc.GAUSS3))):
    for shape in [(4320, 7680), (4000, 6000), (6000, 8000), (5000, 5000), (9000, 12000), (8192, 8192), (12000, 12000), (8640, 15360)]:
        for planes in (1, 3):
            if shape[0] * shape[1] * planes > 3.3e8: continue
            res = []
            for ty in (0, 32, 64, 128):
                ms, tiles = t(shape, scans, capi.RF_PLAN_TILE_ROWS(ty) if ty else 0, planes)
                res.append(f"{'auto' if ty == 0 else ty}:{tiles[1]}={ms:.4f}")
            print(name, shape, f"x{planes}", "  ".join(res), flush=True)
