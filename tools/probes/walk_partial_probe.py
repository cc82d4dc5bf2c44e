#!/usr/bin/env python3
"""A volume whose width and height are not whole tiles (1024 x 1020 x 1020, z x y x x): the one-read pass 1 (partial patches as
an EDGE body, round 5) against the two first passes it replaces; config-5 scans."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import recfilter_amd as rfa
from recfilter_amd import capi
import ref_cases as rc

shape = tuple(int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (1024, 1020, 1020)
scans = rc.REFERENCE_TESTS["test_generic_xyz"]["scans"]
x = torch.rand(shape, device="cuda")
out = torch.empty_like(x)
for _ in range(2):
    for name, flags in (("staged", capi.RF_PLAN_STAGED_PASS1), ("default", 0)):
        with rfa.Plan(shape, scans, flags=flags) as plan:
            plan.execute([x], [out])
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                plan.execute([x], [out])
            e1.record(); torch.cuda.synchronize()
            _, timed = plan.execute_timed([x], [out])
            print(f"{shape} {name}: {e0.elapsed_time(e1) / 10:.3f} ms per step  " + "  ".join(f"{k} {v * 1000:.0f}" for k, v in timed), flush=True)
