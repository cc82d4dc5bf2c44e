"""One shape, cfg3's filter, a dozen executes: something for rocprofv3 to trace (python tools/one_shape.py 16384x16256)."""
import sys, torch
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import recfilter_amd as rfa, ref_cases as rc
h, w = (int(v) for v in sys.argv[1].split("x"))
x = torch.rand((h, w), device="cuda"); y = torch.empty_like(x)
with rfa.Plan((h, w), rc.xy_pm(rc.GAUSS2), clamped=True) as p:
    for _ in range(12): p.execute([x], [y])
    torch.cuda.synchronize()
