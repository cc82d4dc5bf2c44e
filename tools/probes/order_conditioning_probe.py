import sys; sys.path.insert(0,"."); sys.path.insert(0,"tests")
import numpy as np, torch, recfilter_amd as rfa, oracle, ref_cases as rc
from recfilter_amd import capi
rng = np.random.default_rng(5)
def random_scan(dim):
    k = int(rng.integers(4, 9)); poles = []
    while len(poles) < k:
        if k - len(poles) >= 2 and rng.random() < 0.6:
            r, th = rng.uniform(0.2, 0.85), rng.uniform(0.2, 2.9); poles += [r*np.exp(1j*th), r*np.exp(-1j*th)]
        else: poles.append(rng.uniform(-0.8, 0.88))
    p = np.poly(poles).real
    return (dim, bool(rng.integers(0, 2)), [float(rng.uniform(0.1, 0.6))] + [float(-v) for v in p[1:]])
worst = {}
for case in range(24):
    shape = (64, 256)
    scans = [random_scan(0), random_scan(1)]
    clamped = bool(case % 2)
    img = rc.random_image(shape, np.float32, case)
    want = oracle.apply_filter(img.astype(np.float64), scans, clamped)
    x = torch.from_numpy(img).cuda()
    row = []
    for label, kw in (("auto", dict(flags=capi.RF_PLAN_TILED_ONLY)), ("matrix", dict(path=5)), ("untiled", dict(path=1, flags=capi.RF_PLAN_SERIAL_UNTILED))):
        with rfa.Plan(shape, scans, clamped=clamped, **kw) as p:
            got = p.execute([x])[0].cpu().numpy()
            e = rc.rel_err(got, want)
            row.append(f"{label}({p.path_name[6:]}) {e:.1e}")
            worst[label] = max(worst.get(label, 0), e)
    print(case, [len(s[2])-1 for s in scans], clamped, "  ".join(row), flush=True)
print("worst", worst)
