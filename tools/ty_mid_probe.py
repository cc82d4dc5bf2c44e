"""Tile height (automatic / 32 / 64 rows) on mid-size images: python tools/ty_mid_probe.py [HxW ...]"""
import os, sys, time
import numpy as np, torch
ROOT = os.getcwd()
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import recfilter_amd as rfa, ref_cases as rc
from recfilter_amd import capi
def t(shape, scans, flags, planes=1):
    xs = [torch.rand(shape, device="cuda") for _ in range(planes)]; ys = [torch.empty_like(x) for x in xs]
    with rfa.Plan(shape, scans, clamped=True, planes=planes, flags=flags) as p:
        for _ in range(10): p.execute(xs, ys)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(60): p.execute(xs, ys)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / 60 * 1e3, list(p.tiles)
for name, scans in (("gauss2", rc.xy_pm(rc.GAUSS2)), ("gauss3", rc.xy_pm(rc.GAUSS3)), ("sat", [(0, True, [1.0, 1.0]), (1, True, [1.0, 1.0])])):
    for shape in ([tuple(int(v) for v in a.split('x')) for a in sys.argv[1:]] or [(2048, 2048), (2176, 3840), (2048, 4096), (3072, 3072), (3072, 4096), (4096, 4096), (4096, 6144), (6144, 6144)]):
        res = []
        for ty in (0, 32, 64):
            ms, tiles = t(shape, scans, capi.RF_PLAN_TILE_ROWS(ty) if ty else 0)
            res.append(f"{'auto' if ty == 0 else ty}:{tiles[1]}={ms*1e3:.1f}us")
        n64 = ((shape[1] + 255) // 256) * (shape[0] // 64)
        print(name, shape, f"tiles64={n64}", "  ".join(res), flush=True)
