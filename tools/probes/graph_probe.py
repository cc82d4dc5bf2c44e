#!/usr/bin/env python3
"""Does capturing one execute in a HIP graph lower the fixed cost of a small image?  (plan.execute is capturable:
no allocation, no host synchronisation, every launch on the stream it is given.)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import recfilter_amd as rfa
import ref_cases as rc

scans = rc.xy_pm(rc.GAUSS2)
for n in (256, 1024, 2048, 4096, 16384):
    plan = rfa.Plan((n, n), scans, clamped=True)
    img = torch.rand((n, n), device="cuda"); out = torch.empty_like(img); ref = torch.empty_like(img)
    plan.execute([img], [ref])
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): plan.execute([img], [out])
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        plan.execute([img], [out])
    out.zero_()
    g.replay(); torch.cuda.synchronize()
    ok = bool(torch.equal(out, ref))
    iters = 300
    def timed(fn):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(iters): fn()
        torch.cuda.synchronize(); return 1e6 * (time.perf_counter() - t0) / iters
    direct = timed(lambda: plan.execute([img], [out]))
    graph = timed(g.replay)
    print(f"n={n}: direct {direct:.1f} us/exec, graph replay {graph:.1f} us/exec, identical={ok}")
    del g
    plan.close()
