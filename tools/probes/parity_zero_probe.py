#!/usr/bin/env python3
"""Is a sharded result bit-identical to the unsharded one?  (rehearse_n8.py printed sharded_parity 0 for cfg3.)  Counts the
samples that differ, per slab row range, on a small row-sharded image; compares both with the f64 oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import recfilter_amd as rfa
import ref_cases as rc
import oracle
cfg = rc.BASELINE_CONFIGS["cfg3_gaussian2_xy"]
world, slab, width = 4, 256, 1024
img = rc.random_image((world * slab, width), np.float32, 5)
dev = torch.from_numpy(img).cuda()
out_s = torch.empty_like(dev)
ins, outs = list(dev.split(slab)), list(out_s.split(slab))
plans = [rfa.Plan((slab, width), cfg["scans"], clamped=True, shard_rank=r, shard_world=world) for r in range(world)]
for r in range(world): plans[r].begin([ins[r]], [outs[r]])
nb = plans[0].exchange_bytes(0)
g = torch.empty(world * nb, dtype=torch.uint8, device="cuda")
for r in range(world): plans[r].exchange_local(0, g.data_ptr() + r * nb)
for r in range(world): plans[r].exchange_apply(0, g.data_ptr())
for r in range(world): plans[r].finish()
torch.cuda.synchronize()
out_u = torch.empty_like(dev)
with rfa.Plan((world * slab, width), cfg["scans"], clamped=True, flags=rfa.capi.RF_PLAN_TILED_ONLY) as p:
    p.execute([dev], [out_u]); torch.cuda.synchronize()
    print("unsharded path", p.path_name, p.tiles, "sharded", plans[0].path_name, plans[0].tiles)
a, b = out_s.cpu().numpy(), out_u.cpu().numpy()
want = oracle.apply_filter(img.astype(np.float64), cfg["scans"], True)
print("differing samples:", int((a != b).sum()), "of", a.size, "max abs diff", float(np.abs(a - b).max()))
print("rows with differences:", np.unique(np.nonzero(a != b)[0])[:20])
print("vs oracle: sharded", rc.rel_err(a, want), "unsharded", rc.rel_err(b, want))
g.view(torch.float32)[:16]
print("gathered exit carries (first 8 floats of rank 0):", g.view(torch.float32)[:8].cpu().numpy())
