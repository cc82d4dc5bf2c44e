import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch, recfilter_amd as rfa
n = 16384
img = torch.rand((n, n), device="cuda"); out = torch.empty_like(img)
def t(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for order in ([1, 1], [2, 0], [0, 2], [1, 0], [0, 1]):
    print(order, round(t(lambda: rfa.box_difference(img, 5, order, out=out)), 4), "ms")
