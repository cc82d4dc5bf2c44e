import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import recfilter_amd as rfa, oracle, ref_cases as rc
cases = {
 "y+ k1": [(1, True, [1.0, 0.5])],
 "y- k1": [(1, False, [1.0, 0.5])],
 "x+ k1": [(0, True, [1.0, 0.5])],
 "x- k1": [(0, False, [1.0, 0.5])],
 "y+ k2": [(1, True, [1.0, 0.5, 0.1])],
 "x+y+ k1": [(0, True, [1.0, 0.5]), (1, True, [1.0, 0.5])],
}
shape = (192, 768)
img = rc.random_image(shape)
for name, scans in cases.items():
    for clamped in (False, True):
        with rfa.Plan(shape, scans, clamped=clamped, path=3) as plan:
            out = plan.execute([torch.from_numpy(img).cuda()])[0].cpu().numpy()
        want = oracle.apply_filter(img.astype(np.float64), scans, clamped)
        err = np.abs(out - want) / np.maximum(np.abs(want), 1e-2 * np.abs(want).max())
        bad = np.argwhere(err > 1e-4)
        msg = ""
        if len(bad):
            msg = f"rows {np.unique(bad[:,0]//64)} coltiles {np.unique(bad[:,1]//256)}"
        print(f"{name:8s} clamped={clamped} maxerr={err.max():.3e} nbad={len(bad)} {msg}")
