#!/usr/bin/env python3
"""Times the pieces of a sharded step on ONE GPU: `world` slab plans of cfg3 size emulate the ranks, the all-gather is a
device copy.  Shows what the exchange costs each rank besides the collective itself (tuning aid)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import recfilter_amd as rfa
import ref_cases as rc

world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
rank = world // 2
scans = rc.xy_pm(rc.GAUSS2)
plan = rfa.Plan((n, n), scans, clamped=True, shard_rank=rank, shard_world=world)
img = torch.rand((n, n), device="cuda"); out = torch.empty_like(img)
nex = plan.num_exchanges
bufs = [(torch.zeros(plan.exchange_bytes(e), dtype=torch.uint8, device="cuda"),
         torch.zeros(plan.exchange_bytes(e) * world, dtype=torch.uint8, device="cuda")) for e in range(nex)]

def ev():
    e = torch.cuda.Event(enable_timing=True); e.record(); return e

acc = {}
for it in range(12):
    marks = [("start", ev())]
    plan.begin([img], [out]); marks.append(("begin", ev()))
    for e in range(nex):
        send, gath = bufs[e]
        plan.exchange_local(e, send.data_ptr()); marks.append((f"local{e}", ev()))
        for r in range(world):
            gath[r * send.numel():(r + 1) * send.numel()].copy_(send)          # stands in for the all-gather
        marks.append((f"gather{e}(copy)", ev()))
        plan.exchange_apply(e, gath.data_ptr()); marks.append((f"apply{e}", ev()))
    plan.finish(); marks.append(("finish", ev()))
    torch.cuda.synchronize()
    if it >= 2:
        for (_, a), (name, b) in zip(marks[:-1], marks[1:]):
            acc.setdefault(name, []).append(a.elapsed_time(b))
res = {k: round(float(np.median(v)), 4) for k, v in acc.items()}
print(f"world={world} rank={rank} n={n}: total {sum(res.values()):.4f} ms", res)

# whole steps back to back, no events inside (what a rank's stream sees between two collectives)
def step():
    plan.begin([img], [out])
    for e in range(nex):
        send, gath = bufs[e]
        plan.exchange_local(e, send.data_ptr())
        gath.view(world, -1).copy_(send.view(1, -1).expand(world, -1))          # one copy kernel stands in for the all-gather
        plan.exchange_apply(e, gath.data_ptr())
    plan.finish()
for _ in range(5): step()
torch.cuda.synchronize()
a = ev()
for _ in range(40): step()
b = ev(); torch.cuda.synchronize()
print(f"back to back: {a.elapsed_time(b) / 40:.4f} ms per step")
