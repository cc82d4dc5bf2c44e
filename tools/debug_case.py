"""Run one of tests/ref_cases.FUSED_CASES on the GPU and show where it differs from the oracle (debugging aid)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import recfilter_amd as rfa, oracle, ref_cases as rc
name = sys.argv[1] if len(sys.argv) > 1 else "sat"
case = rc.FUSED_CASES[name]
img = rc.random_image(case["shape"])
with rfa.Plan(case["shape"], case["scans"], clamped=case["clamped"], path=3) as plan:
    out = plan.execute([torch.from_numpy(img).cuda()])[0].cpu().numpy()
    print("tiles", plan.tiles)
want = oracle.apply_filter(img.astype(np.float64), case["scans"], case["clamped"])
err = np.abs(out - want) / np.maximum(np.abs(want), 1e-2 * np.abs(want).max())
print("max err", err.max())
bad = np.argwhere(err > 1e-4)
print("n bad", len(bad), "of", err.size)
if len(bad):
    ys, xs = bad[:, 0], bad[:, 1]
    print("bad rows range", ys.min(), ys.max(), "cols range", xs.min(), xs.max())
    print("bad rows unique (first 40)", np.unique(ys)[:40])
    print("bad cols unique (first 40)", np.unique(xs)[:40])
    print("bad cols mod 16 hist", np.bincount(xs % 16, minlength=16))
    print("bad rows mod 16 hist", np.bincount(ys % 16, minlength=16))
    y0, x0 = bad[0]
    print("first bad", y0, x0, out[y0, x0], want[y0, x0])
