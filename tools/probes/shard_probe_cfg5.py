#!/usr/bin/env python3
"""One rank's step of config 5 --strong on ONE GPU: the slab plan of rank world/2 (n/world x n x n) through the stepping calls,
the all-gather a device copy.  One-read pass 1 (default) against the two first passes (RF_PLAN_STAGED_PASS1).
  shard_probe_cfg5.py [world=8] [n=2048]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import recfilter_amd as rfa
from recfilter_amd import capi
import ref_cases as rc

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
scans = rc.BASELINE_CONFIGS["cfg5_generic_xyz"]["scans"]
shape = (n // world, n, n)
img = torch.rand(shape, device="cuda"); out = torch.empty_like(img)
res = {}
for label, fl in (("two first passes", capi.RF_PLAN_STAGED_PASS1), ("one read", 0)) * 2:
    plan = rfa.Plan(shape, scans, shard_rank=world // 2, shard_world=world, flags=fl)
    nex = plan.num_exchanges
    bufs = [(torch.zeros(plan.exchange_bytes(e), dtype=torch.uint8, device="cuda"),
             torch.zeros(plan.exchange_bytes(e) * world, dtype=torch.uint8, device="cuda")) for e in range(nex)]
    def step():
        plan.begin([img], [out])
        for e in range(nex):
            send, gath = bufs[e]
            plan.exchange_local(e, send.data_ptr())
            gath.view(world, -1).copy_(send.view(1, -1).expand(world, -1))      # stands in for the all-gather
            if e == nex - 1 and plan.has_interior: plan.interior()
            plan.exchange_apply(e, gath.data_ptr())
        plan.finish()
    for _ in range(3): step()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): step()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 20
    res[label] = out.clone() if label not in res else res[label]
    print(f"cfg5 {n}^3 / {world}: slab {shape}, {label}: {ms:.3f} ms per step, {nex} exchange(s), interior {plan.has_interior}", flush=True)
    plan.close()
d = (res["one read"] - res["two first passes"]).abs().max() / res["two first passes"].abs().max()
print(f"one read vs two first passes: {float(d):.2e}")
