#!/usr/bin/env python3
"""Step time of every execution path on one image (tuning aid): which path serves which pixel type / size how fast."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import recfilter_amd as rfa
import ref_cases as rc

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
scans = rc.xy_pm(rc.GAUSS2)
for dtype, tdt in ((np.float32, torch.float32), (np.float64, torch.float64), (np.int16, torch.int16)):
    img = (torch.rand((n, n), device="cuda") * 100).to(tdt)
    out = torch.empty_like(img)
    sc = scans if dtype != np.int16 else [(0, True, [1.0, 1.0]), (0, False, [1.0, 1.0]), (1, True, [1.0, 1.0]), (1, False, [1.0, 1.0])]
    for name, kw in (("untiled(lines)", dict(path=1)), ("generic T=64", dict(path=2, tile=[64, 64])), ("generic T=128", dict(path=2, tile=[128, 128])),
                     ("overlapped 64x64", dict(path=4, tile=[64, 64])), ("overlapped 256x16", dict(path=4, tile=[256, 16])),
                     ("overlapped 128x32", dict(path=4, tile=[128, 32])), ("fused", dict(path=3))):
        try:
            plan = rfa.Plan((n, n), sc, dtype=dtype, clamped=True, **kw)
        except Exception as e:
            print(f"{np.dtype(dtype).name:8s} {name:18s} n/a ({str(e)[:60]})"); continue
        for _ in range(3): plan.execute([img], [out])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): plan.execute([img], [out])
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"{np.dtype(dtype).name:8s} {name:18s} {ms:8.3f} ms  {n * n / ms / 1e6:8.1f} Gpixel/s  path={plan.path_name}", flush=True)
        plan.close()
