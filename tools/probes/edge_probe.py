#!/usr/bin/env python3
"""Final-pass cost of partial tiles: whole-tile images against neighbours with a partial last tile column / row, with a row
pitch that is / is not a multiple of 128 bytes.  python tools/edge_probe.py [shapes as HxW ...]"""
import sys, time
import numpy as np
import torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import recfilter_amd as rfa, ref_cases as rc

def t(shape, scans, planes=1, reps=20):
    xs = [torch.rand(shape, device="cuda") for _ in range(planes)]; ys = [torch.empty_like(x) for x in xs]
    with rfa.Plan(shape, scans, clamped=True, planes=planes) as p:
        for _ in range(5): p.execute(xs, ys)
        acc = {}
        for _ in range(10):
            _, tm = p.execute_timed(xs, ys)
            for k, v in tm: acc.setdefault(k, []).append(v)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps): p.execute(xs, ys)
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / reps * 1e3
        px = np.prod(shape) * planes
        print(f"{str(shape):16s} x{planes} {p.path_name} tiles={list(p.tiles)} {ms:.4f} ms  {8 * px / ms / 1e6:.0f} GB/s alg ",
              {k: round(float(np.median(v)), 4) for k, v in acc.items()}, flush=True)

g2 = rc.xy_pm(rc.GAUSS2)
shapes = [tuple(int(v) for v in s.split("x")) for s in sys.argv[1:]] or \
    [(16384, 16384), (16384, 16380), (16380, 16384), (16380, 16380), (16384, 16383), (16384, 16256), (16256, 16256), (8192, 8192),
     (8190, 8192), (8192, 8188), (8192, 8064), (2160, 3840), (2176, 3840), (1080, 1920), (1152, 2048), (4320, 7680), (4352, 7680)]
for shape in shapes:
    t(shape, g2)
