#!/usr/bin/env python3
"""One rank of a z-sharded cfg5 run on ONE GPU, with a delay kernel standing in for the all-gather.

    python tools/overlap_probe.py [--planes 256] [--size 2048] [--world 8] [--rank 3] [--delays-ms 0,1.3,3.0]

The pool's boxes have one GPU, so the collective of an 8-rank run cannot be timed here; what can be shown is what the
plan's structure does with a collective of a given duration.  The stand-in runs on a side stream, as RCCL does: it waits
for the exit carries, copies this rank's contribution into its slot of `gathered`, sleeps `delay` ms (torch.cuda._sleep,
calibrated with HIP events), and the compute stream waits for it before rf_plan_exchange_apply.  Two plans of the same
slab (rank `--rank` of `--world`, 256 planes of 2048^2 by default = cfg5 `--strong` on 8 GPUs):
  late   RF_PLAN_LATE_EXCHANGE: x/y stage, z pass 1, local z carries, [exchange], apply, z pass 2 -- nothing to run beside it
  early  the default: z pass 1 on the raw input, local z carries, [exchange || x/y stage], apply + x/y filter of the carry
         planes, z pass 2 (recfilter_amd/csrc/plan_strided.h)
Expected: late = kernels + delay; early = max(kernels beside it, delay) + the rest.  Prints one JSON line per case."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--planes", type=int, default=256)
    ap.add_argument("--size", type=int, default=2048)
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--rank", type=int, default=3)
    ap.add_argument("--delays-ms", default="0,1.3,3.0")
    ap.add_argument("--steps", type=int, default=10)
    args = ap.parse_args()
    import numpy as np
    import torch
    import ref_cases as rc
    from recfilter_amd import capi
    from recfilter_amd.dist import ShardedFilter

    scans = rc.BASELINE_CONFIGS["cfg5_generic_xyz"]["scans"]
    shape = (args.planes, args.size, args.size)
    x = torch.rand(shape, device="cuda")
    out = torch.empty_like(x)
    side = torch.cuda.Stream()

    # calibrate torch.cuda._sleep: cycles per millisecond
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(1_000_000)
    torch.cuda.synchronize()
    e0.record(); torch.cuda._sleep(20_000_000); e1.record(); torch.cuda.synchronize()
    cycles_per_ms = 20_000_000 / e0.elapsed_time(e1)

    class Work:
        def wait(self):
            torch.cuda.current_stream().wait_stream(side)

    def make_collective(delay_ms, rank):
        def collective(gathered, send):
            side.wait_stream(torch.cuda.current_stream())       # behind the exit carries
            with torch.cuda.stream(side):
                n = send.numel()
                gathered[rank * n:(rank + 1) * n].copy_(send, non_blocking=True)
                if delay_ms > 0:
                    torch.cuda._sleep(int(delay_ms * cycles_per_ms))
            return Work()
        return collective

    def run(flags, delay_ms):
        filt = ShardedFilter(shape, scans, clamped=False, rank=args.rank, world=args.world, flags=flags,
                             collective=make_collective(delay_ms, args.rank))
        for _ in range(3):
            filt.execute([x], [out])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            filt.execute([x], [out])
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / args.steps
        has = filt.plan.has_interior
        nbytes = filt.plan.exchange_bytes(0)
        for p in filt.plans:
            p.close()
        return ms, has, nbytes

    delays = [float(v) for v in args.delays_ms.split(",")]
    base = {}
    for name, flags in (("late", capi.RF_PLAN_LATE_EXCHANGE), ("early", 0)):
        for d in delays:
            ms, has, nbytes = run(flags, d)
            if d == 0:
                base[name] = ms
            print(json.dumps({"exchange": name, "interior_beside_collective": has, "slab": "x".join(map(str, shape)),
                              "rank": args.rank, "world": args.world, "send_MiB": round(nbytes / 2 ** 20, 1),
                              "collective_stand_in_ms": d, "ms_per_step": round(ms, 3),
                              "kernels_only_ms": round(base[name], 3),
                              "sum_kernels_plus_collective": round(base[name] + d, 3),
                              "hidden_ms": round(base[name] + d - ms, 3)}), flush=True)


if __name__ == "__main__":
    main()
