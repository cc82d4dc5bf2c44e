import torch, time
def bw(nbytes, between_bytes=0, reps=30):
    x = torch.empty(nbytes // 4, device="cuda", dtype=torch.float32).normal_()
    z = torch.empty(max(between_bytes, 4) // 4, device="cuda", dtype=torch.float32) if between_bytes else None
    w = torch.empty_like(z) if between_bytes else None
    for _ in range(3):
        x.sum()
        if between_bytes: w.copy_(z)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    tot = 0.0
    for _ in range(reps):
        if between_bytes: w.copy_(z)          # traffic between two uses of x
        e0.record(); x.sum(); e1.record(); torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return nbytes / (tot / reps * 1e-3) / 1e12
for mb in (16, 32, 64, 128, 192, 256, 512, 1024):
    print(f"re-read {mb:5d} MiB: {bw(mb << 20):6.2f} TB/s   with a 64 MiB copy (128 MiB of traffic) between uses: {bw(mb << 20, 64 << 20):6.2f} TB/s", flush=True)
