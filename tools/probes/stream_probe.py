#!/usr/bin/env python3
"""Would running independent planes on two streams hide the latency-bound carry kernels behind the other plane's
HBM-bound passes?  Two one-plane plans, same stream vs two streams."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import recfilter_amd as rfa
import ref_cases as rc

order = int(sys.argv[1]) if len(sys.argv) > 1 else 2
coeff = {1: [0.5, 0.5], 2: rc.GAUSS2, 3: [0.0226432718, 2.29634666, -1.79971004, 0.480720282]}[order]
scans = rc.xy_pm(coeff)
n = 16384
plans = [rfa.Plan((n, n), scans, clamped=True) for _ in range(2)]
ins = [torch.rand((n, n), device="cuda") for _ in range(2)]
outs = [torch.empty_like(t) for t in ins]
streams = [torch.cuda.Stream() for _ in range(2)]

def run(two_streams, iters=20):
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        if two_streams:
            start = torch.cuda.Event(); start.record()
            done = []
            for p in range(2):
                with torch.cuda.stream(streams[p]):
                    streams[p].wait_event(start)
                    plans[p].execute([ins[p]], [outs[p]])
                    ev = torch.cuda.Event(); ev.record(); done.append(ev)
            for ev in done: torch.cuda.current_stream().wait_event(ev)
        else:
            for p in range(2): plans[p].execute([ins[p]], [outs[p]])
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

for _ in range(2):
    print(f"order {order}: two planes on one stream {run(False):.4f} ms, on two streams {run(True):.4f} ms")

# free-running: each stream runs its plane's steps back to back with no per-iteration join, the second stream started
# half a step late so that its latency-bound carry kernels fall into the other plane's HBM-bound passes
def free_run(iters=40):
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    start = torch.cuda.Event(); start.record()
    done = []
    for p in range(2):
        streams[p].wait_event(start)
    with torch.cuda.stream(streams[1]):
        plans[1].begin([ins[1]], [outs[1]]); plans[1].finish() if False else None
    for it in range(iters):
        for p in range(2):
            with torch.cuda.stream(streams[p]):
                plans[p].execute([ins[p]], [outs[p]])
    for p in range(2):
        with torch.cuda.stream(streams[p]):
            ev = torch.cuda.Event(); ev.record(); done.append(ev)
    for ev in done: torch.cuda.current_stream().wait_event(ev)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (2 * iters)

for _ in range(3):
    print(f"order {order}: free-running on two streams {free_run():.4f} ms per plane-step")
