import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch, recfilter_amd as rfa, ref_cases as rc
FLAGS = rfa.capi.RF_PLAN_NO_PLANE_BATCH if len(sys.argv) > 1 and sys.argv[1] == "separate" else 0      # python tools/planes_probe.py [separate]
def run(n, planes, scans, iters=50):
    plan = rfa.Plan((n, n), scans, clamped=True, planes=planes, flags=FLAGS)
    ins = [torch.rand((n, n), device="cuda") for _ in range(planes)]
    outs = [torch.empty_like(t) for t in ins]
    for _ in range(5): plan.execute(ins, outs)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): plan.execute(ins, outs)
    e1.record(); torch.cuda.synchronize(); plan.close()
    return e0.elapsed_time(e1) / iters
g2 = rc.xy_pm(rc.GAUSS2)
print("separate launches" if FLAGS else "batched", " ".join(f"{n}x3={run(n, 3, g2, 50 if n < 8192 else 10):.4f}" for n in (512, 1024, 2048, 4096, 8192, 16384)))
