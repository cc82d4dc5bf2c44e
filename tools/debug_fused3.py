import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import recfilter_amd as rfa, oracle, ref_cases as rc
def run(shape, scans, clamped=False, reps=1):
    img = rc.random_image(shape)
    res=[]
    for _ in range(reps):
        with rfa.Plan(shape, scans, clamped=clamped, path=3) as plan:
            _, times = plan.execute_timed([torch.from_numpy(img).cuda()])
            out = plan.execute([torch.from_numpy(img).cuda()])[0].cpu().numpy()
        want = oracle.apply_filter(img.astype(np.float64), scans, clamped)
        err = np.abs(out - want) / np.maximum(np.abs(want), 1e-2 * np.abs(want).max())
        bad = np.argwhere(err > 1e-4)
        msg = f"rowtiles {np.unique(bad[:,0]//64)} coltiles {np.unique(bad[:,1]//256)}" if len(bad) else ""
        res.append(f"maxerr={err.max():.2e} nbad={len(bad)} {msg}")
    print(shape, scans, clamped, res)
sat=[(0,True,[1.0,1.0]),(1,True,[1.0,1.0])]
half=[(0,True,[1.0,0.5]),(1,True,[1.0,0.5])]
run((128,256), sat, reps=2)
run((128,256), half, reps=2)
run((192,768), sat, reps=2)
run((192,768), half, reps=2)
run((128,512), sat)
run((64,512), sat)
run((128,256), [(1,True,[1.0,1.0])])
run((128,256), [(0,True,[1.0,1.0]),(1,True,[1.0,0.5])])
run((128,256), [(0,True,[1.0,0.5]),(1,True,[1.0,1.0])])
