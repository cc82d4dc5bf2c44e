#!/usr/bin/env python3
"""Fused path per pixel type (tuning aid): step time and per-kernel times of a summed-area table and of an order-2
4-scan integer filter, f32 against int32 / int16."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import recfilter_amd as rfa

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
sat = [(0, True, [1.0, 1.0]), (1, True, [1.0, 1.0])]
k2 = [(0, True, [1.0, 2.0, -1.0]), (0, False, [1.0, 2.0, -1.0]), (1, True, [1.0, 2.0, -1.0]), (1, False, [1.0, 2.0, -1.0])]
for label, scans in (("sat k=1 +x+y", sat), ("k=2 x4", k2)):
    for dtype, tdt in ((np.float32, torch.float32), (np.int32, torch.int32), (np.int16, torch.int16)):
        img = (torch.rand((n, n), device="cuda") * 3).to(tdt)
        out = torch.empty_like(img)
        plan = rfa.Plan((n, n), scans, dtype=dtype, clamped=False, path=3)
        for _ in range(3): plan.execute([img], [out])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): plan.execute([img], [out])
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        _, times = plan.execute_timed([img], [out])
        print(f"{label:14s} {np.dtype(dtype).name:8s} {ms:7.3f} ms  " + " ".join(f"{k}={v*1e3:.0f}us" for k, v in times), flush=True)
        plan.close()
