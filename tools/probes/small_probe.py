#!/usr/bin/env python3
"""Where the fixed cost of a small image goes: CPU enqueue time per execute against the GPU's steady rate, and the
per-kernel times (tuning aid)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import recfilter_amd as rfa
import ref_cases as rc

order = int(sys.argv[1]) if len(sys.argv) > 1 else 2
coeff = {1: [0.5, 0.5], 2: rc.GAUSS2, 3: [0.2, 0.9, -0.3, 0.05]}[order]
scans = rc.xy_pm(coeff)
for n in (256, 512, 1024, 2048, 4096):
    plan = rfa.Plan((n, n), scans, clamped=True)
    img = torch.rand((n, n), device="cuda"); out = torch.empty_like(img)
    for _ in range(20): plan.execute([img], [out])
    torch.cuda.synchronize()
    iters = 500
    t0 = time.perf_counter()
    for _ in range(iters): plan.execute([img], [out])
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    _, timed = plan.execute_timed([img], [out])
    print(f"n={n}: enqueue {1e6*(t1-t0)/iters:.1f} us/exec, total {1e6*(t2-t0)/iters:.1f} us/exec; kernels "
          + " ".join(f"{k}={1e3*v:.1f}us" for k, v in timed))
    plan.close()
