#!/usr/bin/env python3
"""Pass 1 of a 3-D plan in one read (kernels_tails_walk.hip) against the two first passes (RF_PLAN_STAGED_PASS1): results
on small volumes (against the oracle too), step times and kernel times at 1024^3 / 2048^3.
  walk_probe.py [check] [time 1024] [time 2048] [time3 1024: order 3 along x / y]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import recfilter_amd as rfa
from recfilter_amd import capi
import ref_cases as rc

XYZ = rc.REFERENCE_TESTS["test_generic_xyz"]["scans"]
ORDER1 = [(0, True, [0.4, 0.6]), (0, False, [0.5, 0.5]), (1, True, [0.3, 0.7]), (1, False, [0.45, 0.55]), (2, True, [0.5, 0.5]), (2, False, [0.6, 0.4])]
ONE_EACH = [(0, True, [1.0, 0.5, 0.25]), (1, False, [1.0, 0.5, 0.125]), (2, True, [1.0, 0.5, 0.0625])]
Z1 = XYZ[:4] + [(2, False, [0.7, 0.3])]

def check():
    import oracle
    worst = 0.0
    for name, shape, scans, clamped, flags in [
            ("xyz 64x64x256", (64, 64, 256), XYZ, False, 0),
            ("xyz clamped 64x96x512", (64, 96, 512), XYZ, True, 0),
            ("xyz ty64 64x128x256", (64, 128, 256), XYZ, False, capi.RF_PLAN_TILE_ROWS(64)),
            ("xyz ty128 32x256x512 clamped", (32, 256, 512), XYZ, True, capi.RF_PLAN_TILE_ROWS(128)),
            ("xyz tz64 128x64x256", (128, 64, 256), XYZ, True, capi.RF_PLAN_TILE_PLANES(64)),
            ("order1 96x64x256", (96, 64, 256), ORDER1, True, 0),
            ("one scan each 64x64x512", (64, 64, 512), ONE_EACH, False, 0),
            ("z order 1 64x64x256", (64, 64, 256), Z1, True, 0)]:
        rng = np.random.default_rng(7)
        img = rng.random(shape, dtype=np.float32)
        x = torch.from_numpy(img).cuda()
        outs = []
        for fl in (flags, flags | capi.RF_PLAN_STAGED_PASS1):
            plan = rfa.Plan(shape, scans, clamped=clamped, flags=fl | (0 if fl & capi.RF_PLAN_STAGED_PASS1 else capi.RF_PLAN_WALK_PASS1), path=capi.RF_PATH_TILED_FUSED)
            out = torch.empty_like(x)
            plan.execute([x], [out]); torch.cuda.synchronize()
            outs.append((out.cpu().numpy(), None))
            if fl == flags: print("   steps:", " ".join(k for k, _ in plan.execute_timed([x], [out])[1]))
        ref = oracle.apply_filter(img.astype(np.float64), scans, clamped)
        a, b = outs[0][0], outs[1][0]
        scale = np.abs(b).max()
        e_ab = np.abs(a - b).max() / scale
        msg = f"{name}: walk vs staged {e_ab:.2e}"
        if ref is not None:
            msg += f", walk vs oracle {np.abs(a - ref).max() / scale:.2e}, staged vs oracle {np.abs(b - ref).max() / scale:.2e}"
        print(msg, flush=True)
        worst = max(worst, e_ab)
    print("worst", worst)
    assert worst < 2e-5

G3 = rc.GAUSS3
XY3 = [(0, True, G3), (0, False, G3), (1, True, G3), (1, False, G3)] + XYZ[4:]      # order 3 along x / y, order 2 along z (round 5)

def timeit(n, shape=None, scans=None):
    shape = shape or (n, n, n)
    scans = scans or XYZ
    x = torch.rand(shape, device="cuda"); out = torch.empty_like(x)
    for label, fl in (("staged", capi.RF_PLAN_STAGED_PASS1), ("walk", capi.RF_PLAN_WALK_PASS1), ("staged", capi.RF_PLAN_STAGED_PASS1), ("walk", capi.RF_PLAN_WALK_PASS1)):
        plan = rfa.Plan(shape, scans, clamped=False, flags=fl)
        for _ in range(2): plan.execute([x], [out])
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        reps = 5 if shape[0] >= 2048 and shape[1] >= 2048 else 20
        n = n or "x".join(map(str, shape))
        e0.record()
        for _ in range(reps): plan.execute([x], [out])
        e1.record(); torch.cuda.synchronize()
        line = f"{n} {label}: {e0.elapsed_time(e1) / reps:.3f} ms per step"
        try:
            _, t = plan.execute_timed([x], [out])
            line += "  " + "  ".join(f"{k} {v * 1e3:.0f}" for k, v in t)
        except Exception as ex:
            line += f"  (no per-kernel times: {ex})"
        print(line, flush=True)
        del plan

if __name__ == "__main__":
    args = sys.argv[1:] or ["check"]
    i = 0
    while i < len(args):
        if args[i] == "check": check(); i += 1
        elif args[i] == "time": timeit(int(args[i + 1])); i += 2
        elif args[i] == "time3": timeit(int(args[i + 1]), scans=XY3); i += 2
        elif args[i] == "shape": timeit(0, tuple(int(v) for v in args[i + 1].split("x"))); i += 2
        else: raise SystemExit(__doc__)
