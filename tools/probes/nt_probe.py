#!/usr/bin/env python3
"""Non-temporal hints against plain accesses on rows that are / are not multiples of 128 bytes: cfg3's filter at 16384^2 and
16380^2, config 5's at 1024^3 and 1024 x 1020 x 1020.  Run once per library (RECFILTER_AMD_LIB)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import recfilter_amd as rfa
import ref_cases as rc

def t(shape, scans, clamped, reps):
    x = torch.rand(shape, device="cuda"); out = torch.empty_like(x)
    with rfa.Plan(shape, scans, clamped=clamped) as plan:
        for _ in range(3): plan.execute([x], [out])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): plan.execute([x], [out])
        e1.record(); torch.cuda.synchronize()
        _, timed = plan.execute_timed([x], [out])
    return e0.elapsed_time(e1) / reps, "  ".join(f"{k} {v * 1000:.0f}" for k, v in timed)

lib = os.environ.get("RECFILTER_AMD_LIB", "shipped")
for shape in ((16384, 16384), (16380, 16380)):
    ms, ks = t(shape, rc.xy_pm(rc.GAUSS2), True, 20)
    print(f"{lib} {shape}: {ms:.3f} ms  {ks}", flush=True)
for shape in ((1024, 1024, 1024), (1024, 1020, 1020)):
    ms, ks = t(shape, rc.REFERENCE_TESTS["test_generic_xyz"]["scans"], False, 10)
    print(f"{lib} {shape}: {ms:.3f} ms  {ks}", flush=True)
