import sys; sys.path.insert(0,"."); sys.path.insert(0,"tests")
import numpy as np, torch, recfilter_amd as rfa, ref_cases as rc
n=16384; g=rc.GAUSS2
img8=torch.randint(0,256,(n,n),dtype=torch.uint8,device="cuda"); out=torch.empty((n,n),device="cuda")
variants={"xy":rc.xy_pm(g),"x-only":[(0,True,g),(0,False,g)],"y-only":[(1,True,g),(1,False,g)],"+x":[(0,True,g)],"+y":[(1,True,g)]}
for name,sc in variants.items():
    with rfa.Plan((n,n), sc, clamped=True, input_dtype=np.uint8) as plan:
        for _ in range(3): plan.execute([img8],[out])
        acc={}
        for _ in range(10):
            _,t=plan.execute_timed([img8],[out])
            for k,ms in t: acc.setdefault(k,[]).append(ms)
        res={k:round(float(np.median(v)),4) for k,v in acc.items()}
        print(name, round(sum(res.values()),4), res, flush=True)
