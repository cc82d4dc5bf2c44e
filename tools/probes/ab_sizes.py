#!/usr/bin/env python3
"""Times a spread of sizes/configs with the library RECFILTER_AMD_LIB points at (A/B runs of kernel variants)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import recfilter_amd as rfa
import ref_cases as rc

def run(shape, scans, clamped, planes=1, iters=30):
    plan = rfa.Plan(shape, scans, clamped=clamped, planes=planes)
    ins = [torch.rand(shape, device="cuda") for _ in range(planes)]
    outs = [torch.empty_like(t) for t in ins]
    for _ in range(3): plan.execute(ins, outs)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): plan.execute(ins, outs)
    e1.record(); torch.cuda.synchronize()
    plan.close()
    return e0.elapsed_time(e1) / iters

g2 = rc.xy_pm(rc.GAUSS2)
res = {}
for n in (1024, 2048, 4096, 8192, 16384):
    res[f"gauss2_{n}"] = run((n, n), g2, True)
res["cfg2_sat_8192"] = run((8192, 8192), rc.BASELINE_CONFIGS["cfg2_sat"]["scans"] if "cfg2_sat" in rc.BASELINE_CONFIGS else [(0, True, [1.0, 1.0]), (1, True, [1.0, 1.0])], False)
a = 2 - 3 ** 0.5
res["cfg4a_bicubic_3x16384"] = run((16384, 16384), rc.xy_pm([1 + a, -a]), True, planes=3, iters=10)
xyz = rc.BASELINE_CONFIGS["cfg5_generic_xyz"]["scans"]
for n in (256, 512, 1024):
    res[f"cfg5_{n}"] = run((n, n, n), xyz, False, iters=10)
print(os.environ.get("RECFILTER_AMD_LIB", "default"), " ".join(f"{k}={v:.4f}" for k, v in res.items()))
