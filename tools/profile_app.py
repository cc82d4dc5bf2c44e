#!/usr/bin/env python3
"""profile_app.py -- the reference's benchmark harness on the MI355X runtime.

Mirrors `Arguments` (lib/recfilter_utils.cpp:31-112), the apps under apps/ written against the RecFilter
front-end, `RecFilter::profile` (lib/recfilter.cpp:991-1016) and the width sweep of scripts/profile_app.sh:

    python tools/profile_app.py gaussian_3xy -w 4096 -t 32 -iter 100
    python tools/profile_app.py summed_table -w 0            # sweep 64..4096 step 64 -> summed_table.ours.perflog
    python tools/profile_app.py audio_biquads -w 10485760 -iter 20
    python tools/profile_app.py --list

Every row of a .perflog is "<width>\\t<milliseconds>\\t<throughput>", throughput in the reference's unit
(lib/timing.cpp:3-5: pixels*1000 / (ms * 2^20), "MiP/s").  With one iteration and no -nocheck the result is
compared with the CPU oracle (max relative error printed, like the reference's apps print theirs).
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np


def throughput(ms, pixels):                      # lib/timing.cpp:3-5
    return pixels * 1000.0 / (ms * 1024 * 1024)


class Arguments:
    """lib/recfilter_utils.cpp:31-112: same option names and defaults."""

    def __init__(self, argv):
        ap = argparse.ArgumentParser(prog="profile_app.py", add_help=True)
        ap.add_argument("app", nargs="?")
        ap.add_argument("--list", action="store_true")
        ap.add_argument("-w", "--w", "-width", "--width", dest="width", type=int, default=4096,
                        help="image width, 0 = run all widths 64..4096 step 64 and force -nocheck [4096]")
        ap.add_argument("-t", "--t", "-tile", "--tile", dest="block", type=int, default=32,
                        help="tile width for splitting each dimension [32]")
        ap.add_argument("-iter", "--iter", dest="iterations", type=int, default=1)
        ap.add_argument("-nocheck", "--nocheck", dest="nocheck", action="store_true")
        ap.add_argument("--outdir", default=".")
        a = ap.parse_args(argv)
        self.__dict__.update(vars(a))
        if self.width and self.width % self.block:
            ap.error("Width should be a multiple of block size")
        if self.width:
            self.widths = [self.width]
        else:                                     # scripts/profile_app.sh:6-8
            self.widths = list(range(64, 4096 + 1, 64))
            self.nocheck = True
        if self.iterations > 1:
            self.nocheck = True


# ---- the apps: each returns (filter to profile, list of (dim, causal, coeff) scans for the check, clamped,
#      callable giving the expected image from the oracle's filtered image) -----------------------------------------
def _image(shape, seed=0):
    import torch
    g = torch.Generator(device="cuda").manual_seed(1234 + seed)
    return torch.rand(shape, generator=g, device="cuda", dtype=torch.float32)


def _xy_filter(rfa, name, w, coeff_sets, clamped, img):
    x, y = rfa.RecFilterDim("x", w), rfa.RecFilterDim("y", w)
    F = rfa.RecFilter(name)
    if clamped:
        F.set_clamped_image_border()
    F[x, y] = img
    for W in coeff_sets:
        F.add_filter(+x, W); F.add_filter(-x, W); F.add_filter(+y, W); F.add_filter(-y, W)
    return F, x, y


def app_summed_table(rfa, w, t):                  # apps/summed_table/summed_table.cpp
    img = _image((w, w))
    x, y = rfa.RecFilterDim("x", w), rfa.RecFilterDim("y", w)
    F = rfa.RecFilter("SAT")
    F[x, y] = img
    F.add_filter(+x, [1.0, 1.0]); F.add_filter(+y, [1.0, 1.0])
    F.split(x, t, y, t)
    return F, img, F._contents["scans"], False, None


def _gaussian(cascade):
    def build(rfa, w, t):
        img = _image((w, w))
        W1, W2, W3 = (rfa.gaussian_weights(5.0, k) for k in (1, 2, 3))
        sets = {"3xy": [W3], "1xy_2xy": [W1, W2], "1xy_1xy_1xy": [W1, W1, W1], "1xy_2x_2y": [W1, W2], "3x_3y": [W3]}[cascade]
        F, x, y = _xy_filter(rfa, "Gaussian_" + cascade, w, sets, True, img)
        scans = list(F._contents["scans"])
        if cascade == "3xy":
            stages = [F]
        elif cascade == "1xy_2xy":
            stages = F.cascade([0, 1, 2, 3], [4, 5, 6, 7])
        elif cascade == "1xy_1xy_1xy":
            stages = F.cascade([[0, 1, 2, 3], [4, 5, 6, 7], [8, 9, 10, 11]])
        elif cascade == "1xy_2x_2y":
            stages = F.cascade([[0, 1, 2, 3], [4, 5], [6, 7]])
        else:
            stages = F.cascade_by_dimension()
        for f in stages:
            f.split_all_dimensions(t)
            f.gpu_auto_schedule()
        return stages[-1], img, scans, True, None
    return build


def _bspline(coeff, cascaded):
    def build(rfa, w, t):
        img = _image((w, w))
        F, x, y = _xy_filter(rfa, "Bspline", w, [coeff], True, img)
        scans = list(F._contents["scans"])
        stages = F.cascade_by_dimension() if cascaded else [F]
        for f in stages:
            f.split_all_dimensions(t)
        return stages[-1], img, scans, True, None
    return build


_A = 2.0 - 3.0 ** 0.5
BICUBIC = [1 + _A, -_A]                            # apps/bspline/bicubic_filter.cpp:36-37
BIQUINTIC = [1 + _A, -_A, 0.1]                     # apps/bspline/biquintic_*: "inaccurate coefficients, only for measuring performance"


def _usm(optimized):
    def build(rfa, w, t):                         # apps/usm/unsharp_mask_{naive,optimized}.cpp
        img = _image((w, w))
        weight = 1.0
        B, x, y = _xy_filter(rfa, "Blur", w, [rfa.gaussian_weights(5.0, 3)], True, img)
        scans = list(B._contents["scans"])
        B.split_all_dimensions(t)
        expect = lambda blur, im: (1.0 + weight) * im - weight * blur
        if optimized:
            B.compute_at(rfa.Pointwise(w_filtered=-weight, w_input=1.0 + weight))
            return B, img, scans, True, expect

        class Naive:                              # blur, then a separate pointwise pass over the image (torch, like the
            def realize(self):                    # reference's separately scheduled Halide Func)
                return [(1.0 + weight) * img - weight * B.realize()[0]]

            def profile(self, iterations):
                import time
                import torch
                self.realize(); torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(iterations):
                    self.realize()
                torch.cuda.synchronize()
                return (time.perf_counter() - t0) * 1000.0 / iterations
        return Naive(), img, scans, True, expect
    return build


def _box(stages):
    """apps/box/box_filter_{1,3,6}.cpp: stages of box_filter_order_1 (summed-area table + xy differences) and
    box_filter_order_2 (second-order prefix sums along x, differences, then the same along y), radius 5."""
    def build(rfa, w, t):
        B = 5
        img = _image((w, w))
        pad = 3 * (B + 1) + 1
        img[:pad], img[-pad:], img[:, :pad], img[:, -pad:] = 0, 0, 0, 0
        x, y = rfa.RecFilterDim("x", w), rfa.RecFilterDim("y", w)
        I2 = rfa.integral_image_coeff(2)

        import torch
        bufs = [img] + [torch.empty_like(img) for _ in stages]       # bufs[i + 1] = result of stage i
        runs = []
        for i, kind in enumerate(stages):
            src, dst = bufs[i], bufs[i + 1]
            if kind == 1:
                F = rfa.RecFilter("Box1_Sat"); F[x, y] = src
                F.add_filter(+x, [1.0, 1.0]); F.add_filter(+y, [1.0, 1.0]); F.split(x, t, y, t)
                runs.append(lambda F=F, dst=dst: rfa.box_difference(F.realize()[0], B, [1, 1], out=dst))
            else:
                mid = torch.empty_like(img)
                Fx = rfa.RecFilter("Box2_Satx"); Fx[x, y] = src; Fx.add_filter(+x, I2); Fx.split_all_dimensions(t)
                Fy = rfa.RecFilter("Box2_Saty"); Fy[x, y] = mid; Fy.add_filter(+y, I2); Fy.split_all_dimensions(t)
                runs.append(lambda Fx=Fx, Fy=Fy, mid=mid, dst=dst: (
                    rfa.box_difference(Fx.realize()[0], B, [2, 0], out=mid),
                    rfa.box_difference(Fy.realize()[0], B, [0, 2], out=dst)))

        class Chain:
            def realize(self):
                for r in runs:
                    r()
                return [bufs[-1]]

            def profile(self, iterations):
                import time
                self.realize(); torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(iterations):
                    self.realize()
                torch.cuda.synchronize()
                return (time.perf_counter() - t0) * 1000.0 / iterations

        def expect_fn(_unused, im, filt=None):
            import ref_loops
            import oracle
            filt = filt or oracle.apply_filter          # (run_one passes the f32 oracle schedules for its f32 evaluation)
            cur = im
            for kind in stages:
                if kind == 1:
                    sat = filt(cur, [(0, True, [1.0, 1.0]), (1, True, [1.0, 1.0])], False)
                    cur = ref_loops.box_difference(sat, B, [1, 1])
                else:
                    sx = filt(cur, [(0, True, [float(v) for v in I2])], False)
                    dx = ref_loops.box_difference(sx, B, [2, 0])
                    sy = filt(dx, [(1, True, [float(v) for v in I2])], False)
                    cur = ref_loops.box_difference(sy, B, [0, 2])
            return cur
        return Chain(), img, [], False, expect_fn
    return build


def app_dog(rfa, w, t):
    """apps/DoG/diff_gauss.cpp: difference of two Gaussians (sigma 1 and 2), each approximated by three box filters:
    summed-area table -> xy differences for both radii (a Tuple from here on) -> second-order integral along x -> x
    differences -> the same along y -> difference of the two planes.  The recursive filters are RecFilters (the Tuple
    stages filter both planes in one launch per step), the difference Funcs rf_tap_filter."""
    import torch
    import ref_loops
    # three box iterations for sigma 1 and 2, what the app's comment says (diff_gauss.cpp:48-52).  Its call
    # gaussian_box_filter(sigma1, 3) binds as (iterations, sigma) = (1, 3.0) / (2, 3.0) under lib/iir_coeff.h:70 and gives
    # the SAME radius 8 twice, i.e. an output that is zero up to rounding -- nothing a check could hold on to.
    B1, B2 = rfa.gaussian_box_filter(3, 1.0), rfa.gaussian_box_filter(3, 2.0)
    img = _image((w, w))
    pad = max(3 * B1 + 3, 3 * B2 + 3)
    img[:pad], img[-pad:], img[:, :pad], img[:, -pad:] = 0, 0, 0, 0
    x, y = rfa.RecFilterDim("x", w), rfa.RecFilterDim("y", w)
    taps = ref_loops.dog_taps(B1, B2)
    box1 = [torch.empty_like(img) for _ in range(2)]
    box2x = [torch.empty_like(img) for _ in range(2)]
    out = torch.empty_like(img)
    SAT = rfa.RecFilter("SAT"); SAT[x, y] = img
    SAT.add_filter(+x, [1.0, 1.0]); SAT.add_filter(+y, [1.0, 1.0]); SAT.split_all_dimensions(t)
    SAT2x = rfa.RecFilter("SAT2x"); SAT2x[x, y] = box1
    SAT2x.add_filter(+x, [1.0, 2.0, -1.0]); SAT2x.split_all_dimensions(t)
    SAT2y = rfa.RecFilter("SAT2y"); SAT2y[x, y] = box2x
    SAT2y.add_filter(+y, [1.0, 2.0, -1.0]); SAT2y.split_all_dimensions(t)

    class Chain:
        def realize(self):
            sat = SAT.realize()[0]
            for p in range(2):
                rfa.tap_filter([sat], taps["box1"][p], out=box1[p])
            s2x = SAT2x.realize()
            for p in range(2):
                rfa.tap_filter([s2x[p]], taps["box2x"][p], out=box2x[p])
            rfa.tap_filter(SAT2y.realize(), taps["dog"], out=out)
            return [out]

        def profile(self, iterations):
            import time
            self.realize(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(iterations):
                self.realize()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) * 1000.0 / iterations

    def expect_fn(_unused, im, filt=None):
        import oracle
        filt = filt or oracle.apply_filter              # (run_one passes the f32 oracle schedules for its f32 evaluation)
        return ref_loops.dog_pipeline(im, B1, B2, lambda a, sc: filt(a, sc, False))
    return Chain(), img, [], False, expect_fn


def _audio(kind):
    def build(rfa, w, t, param):                  # apps/audio/audio_filter_{high_order,biquads}.cpp
        img = _image((w,))
        x = rfa.RecFilterDim("x", w)
        F = rfa.RecFilter("R_tiled")
        F[x] = img
        expect = None
        if kind == "high_order":
            coeff = [1.0] + [0.01] * param         # "dummy coeff, for performance comparison only"
            F.add_filter(+x, coeff)                # the direct form, as the app has it: orders up to 32 go through the C ABI
        else:
            for _ in range(param + 1):
                F.add_filter(+x, [1.0, 0.1, 0.1])
        scans = list(F._contents["scans"])
        if len(scans) > 4:
            # more sections than one fused pass overlaps: cascade them four at a time (every stage reads the previous
            # stage's device buffer); profile() times the whole chain
            groups = [list(range(i, min(i + 4, len(scans)))) for i in range(0, len(scans), 4)]
            stages = F.cascade(groups)
            for f in stages:
                f.split(x, t)
            return stages[-1], img, scans, False, expect
        F.split(x, t)
        return F, img, scans, False, expect
    return build


APPS = {
    "summed_table": app_summed_table,
    "gaussian_3xy": _gaussian("3xy"), "gaussian_1xy_2xy": _gaussian("1xy_2xy"),
    "gaussian_1xy_1xy_1xy": _gaussian("1xy_1xy_1xy"), "gaussian_1xy_2x_2y": _gaussian("1xy_2x_2y"),
    "gaussian_3x_3y": _gaussian("3x_3y"),
    "bicubic": _bspline(BICUBIC, False), "biquintic_overlapped": _bspline(BIQUINTIC, False),
    "biquintic_cascaded": _bspline(BIQUINTIC, True),
    "usm_naive": _usm(False), "usm_optimized": _usm(True),
    "box_filter_1": _box([1]), "box_filter_3": _box([1, 2]), "box_filter_6": _box([2, 2, 2]),
    "diff_gauss": app_dog,
}
SWEEP_APPS = {      # the 1-D apps sweep a filter parameter at one width instead (apps/audio/*.cpp)
    # "for (int order=1; order<MAX_ORDER; order+=2)", MAX_ORDER 30 (apps/audio/audio_filter_high_order.cpp:14,38)
    "audio_high_order": (_audio("high_order"), lambda rfa: range(1, 30, 2)),
    "audio_biquads": (_audio("biquads"), lambda rfa: range(1, 16)),
}


def run_one(rfa, build, w, args, *extra):
    import oracle
    import ref_cases as rc
    F, img, scans, clamped, expect = build(rfa, w, args.block, *extra)
    ms = F.profile(args.iterations)
    err = None
    if not args.nocheck:
        out = F.realize()[0].cpu().numpy()
        im = img.cpu().numpy()
        want = oracle.apply_filter(im.astype(np.float64), scans, clamped)
        if expect is not None:
            want = expect(want, im.astype(np.float64))
        err = rc.rel_err(out, want)
        run_one.err_f32 = None
        if expect is not None:
            # Apps whose expression cancels (unsharp mask, differences of summed-area tables): the SAME expression evaluated
            # in f32 -- the reference's arithmetic: its apps are Halide float pipelines -- against the f64 evaluation, by the
            # f32 oracle in both of the reference's schedules: the untiled recurrence and the tiled one (tile 32; carries
            # through f32 tile matrices, which is how the reference's GPU schedule sums).  The larger of the two is the
            # error the reference's own result carries; the tests bar ours against it.
            im32 = im.astype(np.float32)
            tiled_ok = all(n % 32 == 0 for n in im32.shape)
            evals = [lambda a, sc, cl=False: oracle.apply_filter(np.ascontiguousarray(a, dtype=np.float32), sc, cl)]
            if tiled_ok:
                evals.append(lambda a, sc, cl=False: oracle.apply_filter_tiled(np.ascontiguousarray(a, dtype=np.float32), sc, cl, tile=32))
            worst = 0.0
            for filt in evals:
                base32 = filt(im32, scans, clamped) if scans else im32
                try:
                    want32 = expect(base32, im32, filt)
                except TypeError:
                    want32 = expect(base32, im32)
                worst = max(worst, rc.rel_err(np.asarray(want32, dtype=np.float64), want))
            run_one.err_f32 = worst
    return ms, err


def main(argv=None):
    args = Arguments(sys.argv[1:] if argv is None else argv)
    if args.list or not args.app:
        print("\n".join(sorted(list(APPS) + list(SWEEP_APPS))))
        return 0
    import recfilter_amd as rfa
    if os.environ.get("PROFILE_APP_CHAINED_CASCADES"):      # A/B: the stages of a cascade as separate plans (the reference's structure)
        rfa.RecFilter.merge_cascades = False
    os.makedirs(args.outdir, exist_ok=True)
    if args.app in SWEEP_APPS:
        build, params = SWEEP_APPS[args.app]
        w = args.width
        log = open(os.path.join(args.outdir, f"{args.app}.tiled.perflog"), "w")
        for p in params(rfa):
            ms, err = run_one(rfa, build, w, args, p)
            row = f"{p}\t{ms:.6f}\t{throughput(ms, w):.3f}"
            log.write(row + "\n"); log.flush()
            print(row + ("" if err is None else f"\tmax rel err {err:.3e}"), flush=True)
        return 0
    if args.app not in APPS:
        print(f"unknown app {args.app}; --list shows the apps", file=sys.stderr)
        return 2
    log = open(os.path.join(args.outdir, f"{args.app}.ours.perflog"), "w") if len(args.widths) > 1 else None
    for w in args.widths:
        ms, err = run_one(rfa, APPS[args.app], w, args)
        row = f"{w}\t{ms:.6f}\t{throughput(ms, w * w):.3f}"
        if log:
            log.write(row + "\n"); log.flush()
        e32 = getattr(run_one, "err_f32", None)
        print(row + ("" if err is None else f"\tmax rel err {err:.3e}") +
              ("" if err is None or e32 is None else f"\tf32 evaluation of the same expression {e32:.3e}"), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
