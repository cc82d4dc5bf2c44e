import sys; sys.path.insert(0,"."); sys.path.insert(0,"tests")
import torch, recfilter_amd as rfa
from recfilter_amd import capi
import ref_cases as rc
XYZ = rc.BASELINE_CONFIGS["cfg5_generic_xyz"]["scans"]
for n in (1024, 2048):
    x = torch.rand((n,n,n), device="cuda"); out = torch.empty_like(x)
    for label, fl in (("default", 0), ("ty64", capi.RF_PLAN_TILE_ROWS(64)), ("ty32", capi.RF_PLAN_TILE_ROWS(32)), ("tz64", capi.RF_PLAN_TILE_PLANES(64)), ("default", 0)):
        with rfa.Plan((n,n,n), XYZ, clamped=True, flags=fl) as plan:
            for _ in range(2): plan.execute([x],[out])
            torch.cuda.synchronize()
            e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
            reps = 10 if n == 1024 else 4
            e0.record()
            for _ in range(reps): plan.execute([x],[out])
            e1.record(); torch.cuda.synchronize()
            _, t = plan.execute_timed([x],[out])
            print(n, label, plan.tiles, round(e0.elapsed_time(e1)/reps,3), "ms ", "  ".join(f"{k} {v*1e3:.0f}" for k,v in t), flush=True)
    del x, out
