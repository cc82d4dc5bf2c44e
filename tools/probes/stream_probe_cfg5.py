#!/usr/bin/env python3
"""Config 5 (2048^3, six scans of order 2): would two z halves on two streams hide the latency-bound kernels between the passes
(x carries, xscan_rows, y carries, z carries: 3 ms of 34) behind the other half's HBM-bound passes?  One plan on the volume
against two plans on 2048 x 2048 x 1024 halves (the same kernels and bytes; the z scans of a half stop at its faces, so the
RESULT differs -- this is a timing probe only), joined per step and free-running."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import recfilter_amd as rfa
import ref_cases as rc

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
scans = rc.REFERENCE_TESTS["test_generic_xyz"]["scans"]
full = rfa.Plan((n, n, n), scans)
halves = [rfa.Plan((n // 2, n, n), scans) for _ in range(2)]
print("paths", full.path_name, halves[0].path_name, "tiles", full.tiles, halves[0].tiles)
vol = torch.rand((n, n, n), device="cuda")
out = torch.empty_like(vol)
ins = [vol[:n // 2], vol[n // 2:]]
outs = [out[:n // 2], out[n // 2:]]
streams = [torch.cuda.Stream() for _ in range(2)]


def timed(fn, iters):
    fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def one():
    full.execute([vol], [out])


def serial_halves():
    for p in range(2):
        halves[p].execute([ins[p]], [outs[p]])


def joined():
    start = torch.cuda.Event(); start.record()
    done = []
    for p in range(2):
        with torch.cuda.stream(streams[p]):
            streams[p].wait_event(start)
            halves[p].execute([ins[p]], [outs[p]])
            ev = torch.cuda.Event(); ev.record(); done.append(ev)
    for ev in done:
        torch.cuda.current_stream().wait_event(ev)


def free(iters=10):
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    start = torch.cuda.Event(); start.record()
    for p in range(2):
        streams[p].wait_event(start)
    for _ in range(iters):
        for p in range(2):
            with torch.cuda.stream(streams[p]):
                halves[p].execute([ins[p]], [outs[p]])
    for p in range(2):
        with torch.cuda.stream(streams[p]):
            ev = torch.cuda.Event(); ev.record()
        torch.cuda.current_stream().wait_event(ev)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for _ in range(2):
    print(f"{n}^3: one plan {timed(one, 6):.3f} ms | two halves, one stream {timed(serial_halves, 6):.3f} ms | two streams, joined per step "
          f"{timed(joined, 6):.3f} ms | two streams free-running {free():.3f} ms per volume", flush=True)
