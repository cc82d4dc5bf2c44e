#!/usr/bin/env python3
"""Per-kernel timing of filter variants on one GPU (tuning aid): which phase of the fused pass costs what."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import recfilter_amd as rfa
import ref_cases as rc

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
g = rc.GAUSS2
variants = {
    "xy(+x-x+y-y)": rc.xy_pm(g),
    "x-only(+x-x)": [(0, True, g), (0, False, g)],
    "y-only(+y-y)": [(1, True, g), (1, False, g)],
    "+x": [(0, True, g)],
    "+y": [(1, True, g)],
    "sat(+x+y,k=1)": [(0, True, [1.0, 1.0]), (1, True, [1.0, 1.0])],
    "gauss3 xy": rc.xy_pm(rc.GAUSS3),
}
img = torch.rand((n, n), device="cuda")
out = torch.empty_like(img)
# ceilings of this box for the same buffers: read-only reduction, device-to-device copy
def _time(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
gb = img.numel() * 4 / 1e9
t_copy, t_sum = _time(lambda: out.copy_(img)), _time(lambda: img.sum())
print(f"ceilings: copy {t_copy:.4f} ms = {2 * gb / t_copy:.1f} TB/s r+w, sum {t_sum:.4f} ms = {gb / t_sum:.1f} TB/s read", flush=True)
pointwise = {"xy + unsharp epilogue": dict(epilogue=(-1.0, 2.0, 0.0)), "xy + prologue": dict(prologue=(1 / 255.0, 0.0)),
             "xy + both": dict(prologue=(1 / 255.0, 0.0), epilogue=(-1.0, 2.0, 0.0))}
runs = [(k, v, {}) for k, v in variants.items()] + [(k, rc.xy_pm(g), kw) for k, kw in pointwise.items()]
for name, scans, kw in runs:
    with rfa.Plan((n, n), scans, clamped=True, **kw) as plan:
        for _ in range(3):
            plan.execute([img], [out])
        acc = {}
        for _ in range(10):
            _, times = plan.execute_timed([img], [out])
            for k, ms in times:
                acc.setdefault(k, []).append(ms)
        res = {k: round(float(np.median(v)), 4) for k, v in acc.items()}
        print(f"{name:16s} TY={plan.tiles[1]} total={sum(res.values()):.4f} {json.dumps(res)}", flush=True)
