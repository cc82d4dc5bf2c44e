#!/usr/bin/env python3
"""Times pass 1 (and the whole cfg3 step) under the current environment knobs; one line of JSON (tuning aid).
The knobs exist only in the A/B build: make -C recfilter_amd/csrc ab; RECFILTER_AMD_LIB=recfilter_amd/librecfilter_amd_ab.so RF_...=1 python tools/p1_probe.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import recfilter_amd as rfa
import ref_cases as rc

cfgname = sys.argv[1] if len(sys.argv) > 1 else "cfg3_gaussian2_xy"
c = rc.BASELINE_CONFIGS[cfgname]
shape = c["shape"]; planes = c.get("planes", 1)
if len(sys.argv) > 2:
    shape = tuple(int(sys.argv[2]) for _ in shape)
imgs = [torch.rand(shape, device="cuda") for _ in range(planes)]
outs = [torch.empty_like(i) for i in imgs]
with rfa.Plan(shape, c["scans"], clamped=c["clamped"], planes=planes) as plan:
    for _ in range(3):
        plan.execute(imgs, outs)
    acc = {}
    for _ in range(15):
        _, times = plan.execute_timed(imgs, outs)
        for k, ms in times:
            acc.setdefault(k, []).append(ms)
    res = {k: round(float(np.median(v)), 4) for k, v in acc.items()}
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        plan.execute(imgs, outs)
    e1.record(); torch.cuda.synchronize()
    step = e0.elapsed_time(e1) / 20
knobs = {k: v for k, v in os.environ.items() if k.startswith("RF_")}
print(json.dumps({"knobs": knobs, "step_ms": round(step, 4), "kernels": res}), flush=True)
