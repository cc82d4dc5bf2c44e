"""tools/matrix_bench.py -- timings of the matrix path (RF_PATH_TILED_MATRIX, kernels_matrix.hip) on one MI355X.

    python tools/matrix_bench.py audio  [samples]      orders 1, 3, .. 29 of apps/audio/audio_filter_high_order.cpp, one
                                                       causal scan of a 1-D signal (default 10,000,000 samples -> 9,999,872)
    python tools/matrix_bench.py image  [size] [order] the 2-D causal + anticausal x/y filter of a given order
    python tools/matrix_bench.py zerophase [samples] [order]  a 1-D signal through a causal and an anticausal scan of that order
    python tools/matrix_bench.py kernels ...           as above with the per-kernel HIP-event times of one execute

Every row: ms per execute (HIP events around `reps` executes on the stream), Msamples/s, bytes per sample under the 8 B
accounting (one read + one write per sample) as GB/s and as a fraction of 8 TB/s, and the max relative error against the
f64 oracle on a bounded prefix / corner (the oracle is the checker, never timed here)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def stable_coeff(order, seed, b=0.4, mass=0.85):
    rng = np.random.default_rng(seed)
    a = rng.standard_normal(order) * np.exp(-0.15 * np.arange(order))
    a *= mass / np.abs(a).sum()
    return [b] + [float(np.float32(v)) for v in a]


def time_plan(plan, dev, reps):
    import torch
    outs = plan.execute(dev)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        plan.execute(dev, outs)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, outs


def main(argv):
    import torch
    import oracle
    import ref_cases as rc
    import recfilter_amd as rfa
    from recfilter_amd import capi
    what = argv[0] if argv else "audio"
    per_kernel = what == "kernels"
    if per_kernel:
        argv = argv[1:]
        what = argv[0] if argv else "audio"
    path = int(os.environ.get("MX_PATH", capi.RF_PATH_AUTO))
    if what == "audio":
        n = (int(argv[1]) if len(argv) > 1 else 10_000_000) // 128 * 128
        sig = rc.random_image((n,), np.float32, 1)
        dev = [torch.from_numpy(sig).cuda()]
        print(f"# audio_high_order, {n} samples, one causal scan, coefficients 0.01 (apps/audio/audio_filter_high_order.cpp:38-42)")
        print("order\tpath\tms\tMsamples/s\tGB/s(8B)\tfrac_of_8TB/s\tkernels\tmax_rel_err(first 65536)")
        for order in range(1, 30, 2):
            scans = [(0, True, [1.0] + [0.01] * order)]
            with rfa.Plan((n,), scans, path=path) as plan:
                ms, outs = time_plan(plan, dev, 20)
                got = outs[0][:65536].cpu().numpy()
                want = oracle.apply_filter(sig[:65536].astype(np.float64), scans, False)
                err = rc.rel_err(got, want)
                gbs = 8.0 * n / ms / 1e6
                print(f"{order}\t{plan.path_name}\t{ms:.4f}\t{n / ms / 1e3:.0f}\t{gbs:.0f}\t{gbs / 8000:.3f}\t{plan.num_kernels}\t{err:.2e}", flush=True)
                if per_kernel and order in (1, 9, 15, 29):
                    _, times = plan.execute_timed(dev)
                    print("    " + "  ".join(f"{nm}={t * 1000:.1f}us" for nm, t in times), flush=True)
        return 0
    if what == "zerophase":
        n = (int(argv[1]) if len(argv) > 1 else 10_000_000) // 128 * 128
        order = int(argv[2]) if len(argv) > 2 else 12
        c = stable_coeff(order, 3)
        scans = [(0, True, c), (0, False, c)]
        sig = rc.random_image((n,), np.float32, 1)
        dev = [torch.from_numpy(sig).cuda()]
        with rfa.Plan((n,), scans, path=path, flags=capi.RF_PLAN_NO_OVERLAP) as plan:
            ms, outs = time_plan(plan, dev, 20)
            m = min(n, 1 << 20)
            want = oracle.apply_filter(sig.astype(np.float64), scans, False)[:m]
            err = rc.rel_err(outs[0][:m].cpu().numpy(), want)
            gbs = 8.0 * n / ms / 1e6
            print(f"{n} samples order {order} causal + anticausal: path {plan.path_name} tiles {plan.tiles} {ms:.4f} ms  {n / ms / 1e3:.0f} Msamples/s  "
                  f"{gbs:.0f} GB/s = {gbs / 8000:.3f} of 8 TB/s  kernels {plan.num_kernels}  rel err {err:.2e}", flush=True)
            if per_kernel:
                _, times = plan.execute_timed(dev)
                print("    " + "  ".join(f"{nm}={t * 1000:.1f}us" for nm, t in times), flush=True)
        return 0
    size = int(argv[1]) if len(argv) > 1 else 16384
    order = int(argv[2]) if len(argv) > 2 else 12
    c = stable_coeff(order, 3)
    scans = [(0, True, c), (0, False, c), (1, True, c), (1, False, c)]
    img = rc.random_image((size, size), np.float32, 1)
    dev = [torch.from_numpy(img).cuda()]
    for clamped in (False, True):
        with rfa.Plan((size, size), scans, clamped=clamped, path=path) as plan:
            ms, outs = time_plan(plan, dev, 5)
            # a corner is enough for the check: 512 rows and columns from the top left, and the image is filtered causally
            # and anticausally -- so compare a crop of a full-size oracle run only for small images
            err = float("nan")
            if size <= 4096:
                want = oracle.apply_filter(img.astype(np.float64), scans, clamped, threads=16)
                err = rc.rel_err(outs[0].cpu().numpy(), want)
            n = size * size
            gbs = 8.0 * n / ms / 1e6
            print(f"{size}^2 order {order} x/y +- {'clamped' if clamped else 'zero'}: path {plan.path_name} tiles {plan.tiles} {ms:.3f} ms  "
                  f"{n / ms / 1e3:.0f} Msamples/s  {gbs:.0f} GB/s = {gbs / 8000:.3f} of 8 TB/s  rel err {err:.2e}", flush=True)
            if per_kernel:
                _, times = plan.execute_timed(dev)
                print("    " + "  ".join(f"{nm}={t * 1000:.1f}us" for nm, t in times), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
