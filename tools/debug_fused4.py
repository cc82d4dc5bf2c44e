import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import recfilter_amd as rfa, oracle, ref_cases as rc
from recfilter_amd import capi
shape=(128,256); scans=[(1,True,[1.0,1.0])]
img = rc.random_image(shape)
def buf(plan, i, n):
    ptr=ctypes.c_void_p(); nb=ctypes.c_size_t()
    capi.check(capi.lib().rf_plan_debug_buffer(plan._h, i, ctypes.byref(ptr), ctypes.byref(nb)))
    out=torch.empty(nb.value//4, dtype=torch.float32, device="cuda")
    hip=ctypes.CDLL("libamdhip64.so")
    hip.hipMemcpy(ctypes.c_void_p(out.data_ptr()), ptr, ctypes.c_size_t(nb.value), 3)
    return out.cpu().numpy()
with rfa.Plan(shape, scans, path=3) as plan:
    d=torch.from_numpy(img).cuda(); o=torch.empty_like(d)
    plan.begin([d],[o]); torch.cuda.synchronize()
    yt=buf(plan,13,0)
    print("yt size", yt.size, "expected", 1*2*1*256)
    cs=img.astype(np.float64).cumsum(0)
    print("tile0 tail vs truth:", np.abs(yt[:256]-cs[63]).max(), " tile1 tail vs truth:", np.abs(yt[256:512]-cs[127]).max())
    print(yt[:4], cs[63][:4], yt[256:260], cs[127][:4])
    plan.finish(); torch.cuda.synchronize()
    out=o.cpu().numpy()
    print("row 63 err", np.abs(out[63]-cs[63]).max(), "row 64 err", np.abs(out[64]-cs[64]).max(), out[64][:3], cs[64][:3], img[64][:3])
