#!/usr/bin/env python3
"""One-read pass 1 against two first passes on volumes of odd extents (profiles/r6/walk_odd_widths.txt):
    python tools/walk_sizes.py 1021 1022 1020 1024"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import recfilter_amd as rfa
from recfilter_amd import capi
import ref_cases as rc

scans = rc.BASELINE_CONFIGS["cfg5_generic_xyz"]["scans"]
if "--int32" in sys.argv:
    # a summed-volume table of int32 samples (bit-exact ring arithmetic): integer pixels keep two first passes -- the one-read pass
    # contracts on the f32 matrix cores, and with the products on the vector ALU it took as long as the passes it replaces
    sys.argv.remove("--int32")
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    sat = [(d, True, [1.0, 1.0]) for d in range(3)]
    x = torch.randint(0, 4, (n, n, n), device="cuda", dtype=torch.int32)
    out = torch.empty_like(x)
    with rfa.Plan((n, n, n), sat, dtype=np.int32) as plan:
        for _ in range(3):
            plan.execute([x], [out])
        acc = {}
        for _ in range(8):
            _, times = plan.execute_timed([x], [out])
            for k, ms in times:
                acc.setdefault(k, []).append(ms)
        med = {k: float(np.median(v)) for k, v in acc.items()}
    print(f"int32 summed-volume table {n}^3 ({plan.path_name}): {sum(med.values()):.3f} ms (" + ", ".join(f"{k} {v:.3f}" for k, v in med.items()) + ")")
    sys.exit(0)
for n in [int(a) for a in sys.argv[1:]] or [1021, 1020, 1024]:
    shape = (n, n, n) if n % 64 == 0 else (1024, n, n)        # (the depth stays whole z tiles: what the one-read pass needs)
    x = torch.rand(shape, device="cuda")
    out = torch.empty_like(x)
    row = []
    for name, flags in (("one read", 0), ("two first passes", capi.RF_PLAN_STAGED_PASS1)):
        with rfa.Plan(shape, scans, flags=flags) as plan:
            for _ in range(3):
                plan.execute([x], [out])
            acc = {}
            for _ in range(8):
                _, times = plan.execute_timed([x], [out])
                for k, ms in times:
                    acc.setdefault(k, []).append(ms)
            med = {k: float(np.median(v)) for k, v in acc.items()}
            row.append(f"{name}: {sum(med.values()):.3f} ms (" + ", ".join(f"{k} {v:.3f}" for k, v in med.items() if v > 0.15 * max(med.values())) + ")")
    print(f"{'x'.join(map(str, shape))}: " + "; ".join(row), flush=True)
    del x, out
    torch.cuda.empty_cache()
