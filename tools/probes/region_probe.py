#!/usr/bin/env python3
"""Where a short timed region's fixed cost sits: wall clock between synchronize() calls against HIP events around the
same K executions (cfg3).  usage: region_probe.py [size]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import recfilter_amd as rfa
sys.path.insert(0, ROOT)
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
cfg = bench.workload("cfg3", n if n != 16384 else None)
shape = cfg["shape"]
x = [torch.rand(shape, device="cuda")]
o = [torch.empty_like(x[0])]
plan = rfa.Plan(shape, cfg["scans"], clamped=cfg["clamped"])
for _ in range(5): plan.execute(x, o)
torch.cuda.synchronize()
for K in (1, 2, 5, 10, 20, 50, 100, 20, 5, 1):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0.record()
    for _ in range(K): plan.execute(x, o)
    t_sub = time.perf_counter()
    e1.record()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    print(f"K={K:4d} wall {1e3*(t1-t0):8.3f} ms ({1e3*(t1-t0)/K:.4f}/step)  events {e0.elapsed_time(e1):8.3f} ms ({e0.elapsed_time(e1)/K:.4f}/step)  "
          f"submit {1e3*(t_sub-t0):7.3f} ms")
