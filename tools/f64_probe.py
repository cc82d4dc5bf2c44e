import sys, time, os
ROOT=os.getcwd(); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,"tests"))
import numpy as np, torch
import recfilter_amd as rfa, ref_cases as rc
for dt,tdt in ((np.float64, torch.float64),(np.float32, torch.float32)):
  for n in (16384, 8192):
    x=torch.rand((n,n),device="cuda",dtype=tdt); y=torch.empty_like(x)
    with rfa.Plan((n,n), rc.xy_pm(rc.GAUSS2), dtype=dt, clamped=True) as p:
        for _ in range(5): p.execute([x],[y])
        acc={}
        for _ in range(8):
            _,tm=p.execute_timed([x],[y])
            for k,v in tm: acc.setdefault(k,[]).append(v)
        torch.cuda.synchronize(); t0=time.perf_counter()
        for _ in range(10): p.execute([x],[y])
        torch.cuda.synchronize(); ms=(time.perf_counter()-t0)/10*1e3
        print(np.dtype(dt).name, n, p.path_name, list(p.tiles), round(ms,4), {k:round(float(np.median(v)),4) for k,v in acc.items()}, flush=True)
