#!/usr/bin/env python3
"""Random large shapes and filters: the fused path against the untiled GPU path (the literal add_filter recurrence),
both through the C ABI.  A one-off robustness sweep for index arithmetic at sizes the oracle is too slow for."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import recfilter_amd as rfa
import ref_cases as rc

ODD = "--odd" in sys.argv          # widths that are not multiples of 4 as well (4- and 8-byte pixels on the fused kernels)
if ODD:
    sys.argv.remove("--odd")
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)


def width(q):
    return 4 * q + (int(rng.integers(0, 4)) if ODD else 0)


n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 24
worst = 0.0
for case in range(0 if len(sys.argv) > 3 and sys.argv[3] in ("overlap", "big", "f64", "1d", "shard", "planes", "streams", "sections", "whole", "walk", "matrix") else n_cases):
    ndim = 2 if case % 4 else 3
    if ndim == 2:
        shape = (int(rng.integers(1, 9000)), width(int(rng.integers(1, 2400))))
    else:
        shape = (int(rng.choice([32, 64, 96])), int(rng.integers(1, 700)), width(int(rng.integers(1, 320))))
    scans = []
    for d in range(ndim):
        for _ in range(int(rng.integers(0, 3 if d < 2 else 2)) + (1 if d == 0 else 0)):
            k = int(rng.integers(1, 4))
            a = rng.uniform(-1.0, 1.0, size=k); a *= rng.uniform(0.2, 0.9) / np.sum(np.abs(a))
            scans.append((d, bool(rng.integers(0, 2)), [float(rng.uniform(0.3, 1.5))] + [float(v) for v in a]))
    clamped = bool(rng.integers(0, 2))
    dtype = torch.int32 if case % 5 == 4 else torch.float32
    if dtype == torch.int32:
        scans = [(d, c, [float(int(rng.integers(1, 3)))] + [float(int(rng.integers(-2, 3))) for _ in co[1:]]) for d, c, co in scans]
        img = torch.randint(0, 256, shape, dtype=torch.int32, device="cuda")
    else:
        img = torch.rand(shape, device="cuda")
    npdt = np.int32 if dtype == torch.int32 else np.float32
    with rfa.Plan(shape, scans, dtype=npdt, clamped=clamped) as pf, rfa.Plan(shape, scans, dtype=npdt, clamped=clamped, path=1, flags=rfa.capi.RF_PLAN_SERIAL_UNTILED) as pu:
        of, ou = pf.execute([img])[0], pu.execute([img])[0]
        torch.cuda.synchronize()
        if dtype == torch.int32:
            err = float((of != ou).sum().item())
        else:
            peak = float(ou.abs().max().item())
            err = float(((of - ou).abs() / torch.clamp(ou.abs(), min=1e-2 * peak)).max().item())
        worst = max(worst, err)
        print(f"{case:3d} {pf.path_name:13s} {str(shape):22s} scans={len(scans)} clamped={int(clamped)} {str(dtype)[6:]:8s} err={err:.3e}",
              "" if err < (1 if dtype == torch.int32 else 2e-4) else "  <-- CHECK", flush=True)
print("worst", worst)

# ---- the fully overlapped tiling (path 4) on random small shapes / tiles / pixel types against the untiled path ----
if len(sys.argv) > 3 and sys.argv[3] == "overlap":
    worst = 0.0
    for case in range(n_cases):
        ndim = 2 if case % 3 else 3
        tiles = [int(rng.choice([4, 8, 16, 32])) for _ in range(ndim)]
        while np.prod(tiles) > 4096:
            tiles[int(rng.integers(0, ndim))] //= 2
        shape = tuple(int(t * rng.integers(1, 9)) for t in reversed(tiles))          # numpy order: outermost first
        scans = []
        for d in range(ndim):
            for _ in range(int(rng.integers(1, 3))):
                k = int(rng.integers(1, min(5, tiles[d]) + 1))
                a = rng.uniform(-1.0, 1.0, size=k); a *= rng.uniform(0.2, 0.9) / np.sum(np.abs(a))
                scans.append((d, bool(rng.integers(0, 2)), [float(rng.uniform(0.3, 1.5))] + [float(v) for v in a]))
        clamped = bool(rng.integers(0, 2))
        npdt = [np.float32, np.float64, np.int32][case % 3]
        if npdt == np.int32:
            scans = [(d, c, [float(int(rng.integers(1, 3)))] + [float(int(rng.integers(-2, 3))) for _ in co[1:]]) for d, c, co in scans]
            img = torch.randint(0, 16, shape, dtype=torch.int32, device="cuda")
        else:
            img = torch.rand(shape, device="cuda", dtype=torch.float64 if npdt == np.float64 else torch.float32)
        with rfa.Plan(shape, scans, dtype=npdt, clamped=clamped, path=4, tile=tiles) as po, \
                rfa.Plan(shape, scans, dtype=npdt, clamped=clamped, path=1, flags=rfa.capi.RF_PLAN_SERIAL_UNTILED) as pu:
            oo, ou = po.execute([img])[0], pu.execute([img])[0]
            torch.cuda.synchronize()
            if npdt == np.int32:
                err = float((oo != ou).sum().item())
            else:
                peak = float(ou.abs().max().item())
                err = float(((oo - ou).abs() / torch.clamp(ou.abs(), min=1e-2 * peak)).max().item())
            worst = max(worst, err)
            print(f"{case:3d} {po.path_name:16s} {str(shape):18s} tiles={tiles} scans={len(scans)} clamped={int(clamped)} {np.dtype(npdt).name:8s} err={err:.3e}",
                  "" if err < (1 if npdt == np.int32 else 2e-4) else "  <-- CHECK", flush=True)
    print("worst (overlapped)", worst)

# ---- large images: what RF_PATH_AUTO picks there (256 x 128 tiles for order >= 2) against the untiled path ----
if len(sys.argv) > 3 and sys.argv[3] == "big":
    worst = 0.0
    for case in range(n_cases):
        rows = 128 * int(rng.integers(48, 129)) if case % 3 else int(rng.integers(6200, 16000))
        cols = 4 * int(rng.integers(2100, 4097))
        scans = []
        for d in range(2):
            for _ in range(int(rng.integers(1, 3))):
                k = int(rng.integers(2, 4)) if d == 0 or not scans else int(rng.integers(1, 4))
                a = rng.uniform(-1.0, 1.0, size=k); a *= rng.uniform(0.2, 0.9) / np.sum(np.abs(a))
                scans.append((d, bool(rng.integers(0, 2)), [float(rng.uniform(0.3, 1.5))] + [float(v) for v in a]))
        clamped = bool(rng.integers(0, 2))
        img = torch.rand((rows, cols), device="cuda")
        with rfa.Plan((rows, cols), scans, clamped=clamped) as pf, rfa.Plan((rows, cols), scans, clamped=clamped, path=1, flags=rfa.capi.RF_PLAN_SERIAL_UNTILED) as pu:
            of, ou = pf.execute([img])[0], pu.execute([img])[0]
            torch.cuda.synchronize()
            peak = float(ou.abs().max().item())
            err = float(((of - ou).abs() / torch.clamp(ou.abs(), min=1e-2 * peak)).max().item())
            worst = max(worst, err)
            print(f"{case:3d} {pf.path_name:13s} tiles={list(pf.tiles)} {str((rows, cols)):16s} scans={len(scans)} clamped={int(clamped)} err={err:.3e}",
                  "" if err < 2e-4 else "  <-- CHECK", flush=True)
    print("worst (big)", worst)

# ---- f64 pixels on the fused kernels and 1-D signals of arbitrary length, against the untiled path ----
if len(sys.argv) > 3 and sys.argv[3] in ("f64", "1d"):
    mode = sys.argv[3]
    worst = 0.0
    for case in range(n_cases):
        if mode == "f64":
            shape = (int(rng.integers(1, 3000)), width(int(rng.integers(1, 900))))
            dims = 2
        else:
            shape = (int(rng.integers(8192, 3000000)),)
            dims = 1
        scans = []
        for d in range(dims):
            for _ in range(int(rng.integers(1, 4)) if dims == 1 else int(rng.integers(0, 3)) + (1 if d == 0 else 0)):
                k = int(rng.integers(1, 4))
                a = rng.uniform(-1.0, 1.0, size=k); a *= rng.uniform(0.2, 0.9) / np.sum(np.abs(a))
                scans.append((d, bool(rng.integers(0, 2)), [float(rng.uniform(0.3, 1.5))] + [float(v) for v in a]))
        clamped = bool(rng.integers(0, 2))        # (1-D: clamped signals run the zero-border plan plus border corrections)
        tdt = torch.float64 if mode == "f64" else torch.float32
        npdt = np.float64 if mode == "f64" else np.float32
        img = torch.rand(shape, device="cuda", dtype=tdt)
        # (the reference: the literal recurrence, one thread per line)
        with rfa.Plan(shape, scans, dtype=npdt, clamped=clamped) as pf, \
             rfa.Plan(shape, scans, dtype=npdt, clamped=clamped, path=1, flags=rfa.capi.RF_PLAN_SERIAL_UNTILED) as pu:
            of, ou = pf.execute([img])[0], pu.execute([img])[0]
            torch.cuda.synchronize()
            peak = float(ou.abs().max().item())
            err = float(((of - ou).abs() / torch.clamp(ou.abs(), min=1e-2 * peak)).max().item())
            worst = max(worst, err)
            tol = 1e-10 if mode == "f64" else 2e-4
            print(f"{case:3d} {pf.path_name:13s} {str(shape):16s} scans={[(d, int(c), len(co) - 1) for d, c, co in scans]} clamped={int(clamped)} err={err:.3e}",
                  "" if err < tol else "  <-- CHECK", flush=True)
    print(f"worst ({mode})", worst)

# ---- emulated ranks with slabs of different extents against the unsharded plan (fused 2-D rows / 3-D z slabs) ----
if len(sys.argv) > 3 and sys.argv[3] == "shard":
    worst = 0.0
    for case in range(n_cases):
        world = int(rng.integers(2, 6))
        gran = int(rng.choice([32, 64, 128]))
        ext = [gran * int(rng.integers(1, 5)) for _ in range(world)]
        three = case % 3 == 0
        shape = (sum(ext), int(rng.integers(1, 6)) * 32, width(int(rng.integers(16, 200)))) if three else (sum(ext), width(int(rng.integers(16, 500))))
        nd = len(shape)
        scans = []
        for d in range(nd):
            for _ in range(int(rng.integers(1, 3))):
                k = int(rng.integers(1, 4))
                a = rng.uniform(-1.0, 1.0, size=k); a *= rng.uniform(0.2, 0.9) / np.sum(np.abs(a))
                scans.append((d, bool(rng.integers(0, 2)), [float(rng.uniform(0.3, 1.5))] + [float(v) for v in a]))
        clamped = bool(rng.integers(0, 2))
        import recfilter_amd.plan as _rp
        _rp.DEFAULT_FLAGS = rfa.capi.RF_PLAN_TILE_ROWS(128) if rng.integers(0, 2) else 0
        img = torch.rand(shape, device="cuda")
        with rfa.Plan(shape, scans, clamped=clamped) as p1:
            want = p1.execute([img])[0]
        lo = [sum(ext[:r]) for r in range(world)]
        plans = [rfa.Plan((ext[r],) + tuple(shape[1:]), scans, clamped=clamped, shard_rank=r, shard_world=world, shard_extents=ext) for r in range(world)]
        ins = [img[lo[r]:lo[r] + ext[r]].contiguous() for r in range(world)]
        outs = [torch.empty_like(t) for t in ins]
        for r in range(world): plans[r].begin([ins[r]], [outs[r]])
        for e in range(plans[0].num_exchanges):
            nb = plans[0].exchange_bytes(e)
            g = torch.empty(world * nb, dtype=torch.uint8, device="cuda")
            for r in range(world): plans[r].exchange_local(e, g.data_ptr() + r * nb)
            for r in range(world): plans[r].exchange_apply(e, g.data_ptr())
        for r in range(world): plans[r].finish()
        torch.cuda.synchronize()
        got = torch.cat(outs, dim=0)
        peak = float(want.abs().max().item())
        err = float(((got - want).abs() / torch.clamp(want.abs(), min=1e-2 * peak)).max().item())
        worst = max(worst, err)
        print(f"{case:3d} {plans[0].path_name:13s} tiles={list(plans[0].tiles)} {str(shape):18s} ext={ext} scans={len(scans)} clamped={int(clamped)} nex={plans[0].num_exchanges} err={err:.3e}",
              "" if err < 2e-4 else "  <-- CHECK", flush=True)
        for p in plans: p.close()
    print("worst (shard)", worst)

# ---- Tuple planes, pixel types, uint8 input and pointwise stages on random shapes, against the untiled path ----
if len(sys.argv) > 3 and sys.argv[3] == "planes":
    worst = 0.0
    import recfilter_amd.plan as _rp
    SERIAL = rfa.capi.RF_PLAN_SERIAL_UNTILED        # (path=1 plans below: the literal recurrence as the reference)
    for case in range(n_cases):
        kind = ["f32", "i32", "i16", "u8", "f32pw"][case % 5]
        q = int(rng.integers(1, 700))
        shape = (int(rng.integers(1, 1500)), 4 * q if kind in ("i16", "u8") else width(q))       # (2-byte pixels / byte inputs: multiples of 4)
        planes = int(rng.integers(1, 6))
        scans = []
        for d in range(2):
            for _ in range(int(rng.integers(0, 3)) + (1 if d == 0 else 0)):
                k = int(rng.integers(1, 4))
                if kind in ("i32", "i16"):
                    scans.append((d, bool(rng.integers(0, 2)), [float(int(rng.integers(1, 3)))] + [float(int(rng.integers(-2, 3))) for _ in range(k)]))
                else:
                    a = rng.uniform(-1.0, 1.0, size=k); a *= rng.uniform(0.2, 0.9) / np.sum(np.abs(a))
                    scans.append((d, bool(rng.integers(0, 2)), [float(rng.uniform(0.3, 1.5))] + [float(v) for v in a]))
        clamped = bool(rng.integers(0, 2))
        kw = {}
        if kind == "i32":
            imgs = [torch.randint(0, 64, shape, dtype=torch.int32, device="cuda") for _ in range(planes)]; npdt = np.int32
        elif kind == "i16":
            imgs = [torch.randint(0, 8, shape, dtype=torch.int16, device="cuda") for _ in range(planes)]; npdt = np.int16
        elif kind == "u8":
            imgs = [torch.randint(0, 256, shape, dtype=torch.uint8, device="cuda") for _ in range(planes)]; npdt = np.float32
            kw = dict(input_dtype=np.uint8, prologue=(1.0 / 255.0, 0.0))
        else:
            imgs = [torch.rand(shape, device="cuda") for _ in range(planes)]; npdt = np.float32
            if kind == "f32pw":
                kw = dict(prologue=(0.5, 0.25), epilogue=(float(np.float32(-0.75)), float(np.float32(1.75)) if case % 2 else 0.0, 0.125))
        with rfa.Plan(shape, scans, dtype=npdt, clamped=clamped, planes=planes, **kw) as pf, \
                rfa.Plan(shape, scans, dtype=npdt, clamped=clamped, planes=planes, path=1, flags=SERIAL, **kw) as pu:
            of, ou = pf.execute(imgs), pu.execute(imgs)
            torch.cuda.synchronize()
            err = 0.0
            for a_, b_ in zip(of, ou):
                if kind in ("i32", "i16"):
                    err = max(err, float((a_ != b_).sum().item()))
                else:
                    peak = float(b_.abs().max().item())
                    err = max(err, float(((a_ - b_).abs() / torch.clamp(b_.abs(), min=1e-2 * peak)).max().item()))
            worst = max(worst, err)
            print(f"{case:3d} {pf.path_name:13s} {kind:6s} planes={planes} {str(shape):14s} scans={len(scans)} clamped={int(clamped)} err={err:.3e}",
                  "" if err < (1 if kind in ("i32", "i16") else 2e-4) else "  <-- CHECK", flush=True)
    print("worst (planes)", worst)

# ---- one plan, many streams and host threads (execution instances, capi.cpp acquire_instance) against a serial run ----
if len(sys.argv) > 3 and sys.argv[3] == "streams":
    import threading
    worst = 0.0
    for case in range(n_cases):
        shape = (int(rng.integers(64, 3000)), width(int(rng.integers(16, 900))))
        scans = []
        for d in range(2):
            for _ in range(int(rng.integers(1, 3))):
                k = int(rng.integers(1, 4))
                a = rng.uniform(-1.0, 1.0, size=k); a *= rng.uniform(0.2, 0.9) / np.sum(np.abs(a))
                scans.append((d, bool(rng.integers(0, 2)), [float(rng.uniform(0.3, 1.5))] + [float(v) for v in a]))
        clamped = bool(rng.integers(0, 2))
        n_streams, n_threads, reps = int(rng.integers(2, 6)), int(rng.integers(1, 5)), int(rng.integers(1, 4))
        imgs = [torch.rand(shape, device="cuda") for _ in range(n_streams)]
        with rfa.Plan(shape, scans, clamped=clamped) as plan:
            want = [plan.execute([im])[0].clone() for im in imgs]
            torch.cuda.synchronize()
            streams = [torch.cuda.Stream() for _ in range(n_streams)]
            outs = [torch.zeros_like(im) for im in imgs]
            errors = []

            def worker(tid):
                try:
                    for _ in range(reps):
                        for i in range(tid, n_streams, n_threads):
                            plan.execute([imgs[i]], [outs[i]], stream=streams[i])
                except Exception as exc:
                    errors.append(repr(exc))
            threads = [threading.Thread(target=worker, args=(t,)) for t in range(n_threads)]
            for th in threads: th.start()
            for th in threads: th.join()
            torch.cuda.synchronize()
            bad = sum(int((o != w).sum().item()) for o, w in zip(outs, want))
            worst = max(worst, bad)
            print(f"{case:3d} {plan.path_name:13s} {str(shape):14s} streams={n_streams} threads={n_threads} reps={reps} instances={plan.num_instances} "
                  f"differing samples={bad} errors={errors}", "" if bad == 0 and not errors else "  <-- CHECK", flush=True)
    print("worst (streams)", worst)

# ---- orders 4..8 as sections: random stable pole sets, both borders (clamped: zero-border form behind border modifications),
# shapes the rewrite takes (width % 16 == 0, height % 32 == 0), 2-D and 3-D, against the serial untiled kernel on the SAME
# high-order coefficients ----
if len(sys.argv) > 3 and sys.argv[3] == "sections":
    worst = 0.0
    for case in range(n_cases):
        def random_scan(dim):
            k = int(rng.integers(4, 9))
            poles = []
            while len(poles) < k:
                if k - len(poles) >= 2 and rng.random() < 0.6:
                    r, th = rng.uniform(0.2, 0.85), rng.uniform(0.2, 2.9)
                    poles += [r * np.exp(1j * th), r * np.exp(-1j * th)]
                else:
                    poles.append(rng.uniform(-0.8, 0.88))
            p = np.poly(poles).real
            return (dim, bool(rng.integers(0, 2)), [float(rng.uniform(0.1, 0.6))] + [float(-v) for v in p[1:]])
        ndim = 2 if case % 3 else 3
        if ndim == 2:
            shape = (32 * int(rng.integers(1, 90)), 16 * int(rng.integers(1, 300)))
        else:
            shape = (int(rng.choice([32, 64, 128])), 32 * int(rng.integers(1, 12)), 16 * int(rng.integers(1, 40)))
        scans = []
        for d in range(ndim):
            n_here = int(rng.integers(0, 2)) + (1 if d == 0 else 0)
            for _ in range(n_here):
                scans.append(random_scan(d) if rng.random() < 0.7 else (d, bool(rng.integers(0, 2)), [0.4, 0.5, -0.1]))
        if all(len(co) - 1 <= 3 for _, _, co in scans):
            scans[0] = random_scan(scans[0][0])
        clamped = bool(rng.integers(0, 2))
        img = torch.rand(shape, device="cuda")
        with rfa.Plan(shape, scans, clamped=clamped, flags=rfa.capi.RF_PLAN_TILED_ONLY) as pf, \
             rfa.Plan(shape, scans, clamped=clamped, path=1, flags=rfa.capi.RF_PLAN_SERIAL_UNTILED) as pu:
            of, ou = pf.execute([img])[0], pu.execute([img])[0]
            torch.cuda.synchronize()
            peak = float(ou.abs().max().item())
            err = float(((of - ou).abs() / torch.clamp(ou.abs(), min=1e-2 * peak)).max().item())
            worst = max(worst, err)
            extra = ""
            if err >= 1e-4 and img.numel() <= (1 << 24):
                # which of the two f32 evaluations is off: both against the f64 oracle, and the direct form on the matrix path
                import oracle
                want = oracle.apply_filter(img.cpu().numpy().astype(np.float64), scans, clamped)
                e_f, e_u = rc.rel_err(of.cpu().numpy(), want), rc.rel_err(ou.cpu().numpy(), want)
                extra = f"  [vs f64 oracle: this plan {e_f:.2e}, untiled f32 {e_u:.2e}"
                try:
                    with rfa.Plan(shape, scans, clamped=clamped, path=5) as pm:
                        extra += f", matrix path {rc.rel_err(pm.execute([img])[0].cpu().numpy(), want):.2e}"
                except Exception:
                    pass
                extra += "]"
            print(f"{case:3d} {pf.path_name:13s} {str(shape):22s} orders={[len(co) - 1 for _, _, co in scans]} clamped={int(clamped)} err={err:.3e}",
                  ("" if err < 2e-4 else "  <-- CHECK") + extra, flush=True)
    print("worst (sections)", worst)

# ---- images of whole tiles (what the matrix-core pass 1 takes), orders 1..3, one or two scans per dimension, planes ----
if len(sys.argv) > 3 and sys.argv[3] == "whole":
    worst = 0.0
    for case in range(n_cases):
        ty = int(rng.choice([32, 64, 128]))
        shape = (ty * int(rng.integers(1, 40)), 256 * int(rng.integers(1, 24)))
        planes = int(rng.choice([1, 1, 3]))
        k = int(rng.integers(1, 4))
        scans = []
        for d in range(2):
            for _ in range(int(rng.integers(0 if d else 1, 3))):
                a = rng.uniform(-1.0, 1.0, size=k); a *= rng.uniform(0.2, 0.9) / np.sum(np.abs(a))
                scans.append((d, bool(rng.integers(0, 2)), [float(rng.uniform(0.3, 1.5))] + [float(v) for v in a]))
        clamped = bool(rng.integers(0, 2))
        imgs = [torch.rand(shape, device="cuda") for _ in range(planes)]
        flags = rfa.capi.RF_PLAN_TILED_ONLY | rfa.capi.RF_PLAN_TILE_ROWS(ty) | (rfa.capi.RF_PLAN_MFMA_PASS1 if case % 2 else 0)
        with rfa.Plan(shape, scans, clamped=clamped, planes=planes, flags=flags) as pf, \
             rfa.Plan(shape, scans, clamped=clamped, planes=planes, path=1, flags=rfa.capi.RF_PLAN_SERIAL_UNTILED) as pu:
            of, ou = pf.execute(imgs), pu.execute(imgs)
            torch.cuda.synchronize()
            err = 0.0
            for a_, b_ in zip(of, ou):
                peak = float(b_.abs().max().item())
                err = max(err, float(((a_ - b_).abs() / torch.clamp(b_.abs(), min=1e-2 * peak)).max().item()))
            worst = max(worst, err)
            print(f"{case:3d} {pf.path_name:13s} {str(shape):16s} x{planes} tiles={list(pf.tiles)} order={k} scans={len(scans)} clamped={int(clamped)} err={err:.3e}",
                  "" if err < 2e-4 else "  <-- CHECK", flush=True)
    print("worst (whole)", worst)

# ---- volumes of whole tiles through the one-read pass 1 (kernels_tails_walk.hip): random tile heights / z tiles, orders 1..2 per
# stage, one or two scans per dimension, both borders; whole volumes and z slabs (emulated ranks) against the untiled path ----
if len(sys.argv) > 3 and sys.argv[3] == "walk":
    worst = 0.0
    for case in range(n_cases):
        ty, tz = int(rng.choice([32, 64, 128])), int(rng.choice([32, 64, 128]))
        world = int(rng.choice([1, 1, 2, 3]))
        shape = (tz * int(rng.integers(1, 4)) * world, ty * int(rng.integers(1, 5)), 256 * int(rng.integers(1, 5)))
        if case % 2 == 1:          # ragged: a partial last tile row / column (round 6: tall patches, and with --odd any width)
            shape = (shape[0], int(rng.integers(1, 4 * ty + 1)), width(int(rng.integers(1, 300))))
        kxy, kz = int(rng.integers(1, 3)), int(rng.integers(1, 3))
        scans = []
        for d in range(3):
            for _ in range(int(rng.integers(1, 3))):
                k = kxy if d < 2 else kz
                a = rng.uniform(-1.0, 1.0, size=k); a *= rng.uniform(0.2, 0.9) / np.sum(np.abs(a))
                scans.append((d, bool(rng.integers(0, 2)), [float(rng.uniform(0.3, 1.5))] + [float(v) for v in a]))
        clamped = bool(rng.integers(0, 2))
        img = torch.rand(shape, device="cuda")
        flags = rfa.capi.RF_PLAN_TILED_ONLY | rfa.capi.RF_PLAN_TILE_ROWS(ty) | rfa.capi.RF_PLAN_TILE_PLANES(tz) | rfa.capi.RF_PLAN_WALK_PASS1
        with rfa.Plan(shape, scans, clamped=clamped, path=1, flags=rfa.capi.RF_PLAN_SERIAL_UNTILED) as pu:
            ou = pu.execute([img])[0]
        of = torch.empty_like(img)
        if world == 1:
            with rfa.Plan(shape, scans, clamped=clamped, flags=flags) as pf:
                _, timed = pf.execute_timed([img], [of])
                names, tiles = [k_ for k_, _ in timed], list(pf.tiles)
        else:
            nz = shape[0] // world
            plans = [rfa.Plan((nz,) + shape[1:], scans, clamped=clamped, flags=flags, shard_rank=r, shard_world=world) for r in range(world)]
            tiles, names = list(plans[0].tiles), ["walk_tails" if plans[0].has_interior else "?"]
            try:
                plans[0].table("H_z")
            except Exception:
                names = ["(two first passes)"]
            for r in range(world):
                plans[r].begin([img[r * nz:(r + 1) * nz]], [of[r * nz:(r + 1) * nz]])
            for e in range(plans[0].num_exchanges):
                nb = plans[0].exchange_bytes(e)
                gathered = torch.empty(world * nb, dtype=torch.uint8, device="cuda")
                for r in range(world):
                    plans[r].exchange_local(e, gathered.data_ptr() + r * nb)
                for r in range(world):
                    if e == plans[0].num_exchanges - 1 and plans[r].has_interior: plans[r].interior()
                    plans[r].exchange_apply(e, gathered.data_ptr())
            for r in range(world):
                plans[r].finish()
            for p_ in plans: p_.close()
        torch.cuda.synchronize()
        peak = float(ou.abs().max().item())
        err = float(((of - ou).abs() / torch.clamp(ou.abs(), min=1e-2 * peak)).max().item())
        worst = max(worst, err)
        took = "walk" if "walk_tails" in names else "NOT TAKEN " + str(names[:2])
        print(f"{case:3d} {str(shape):18s} world={world} tiles={tiles} orders={kxy}/{kz} scans={len(scans)} clamped={int(clamped)} {took} err={err:.3e}",
              "" if err < 2e-4 else "  <-- CHECK", flush=True)
    print("worst (walk)", worst)


# ---- the matrix path (path 5: orders up to 32 in their direct form, every stage a GEMM on the matrix cores) on random shapes,
# orders, directions, borders and plane counts against the serial untiled kernel (one recurrence per line, the literal operator)
if len(sys.argv) > 3 and sys.argv[3] == "matrix":
    worst = 0.0
    for case in range(n_cases):
        kind = case % 4
        ragged = case % 8 >= 4      # extents no tile divides (widths multiples of 4): padding where each scan leaves the image
        if kind == 0:        # 1-D signals: lane = tile; every tile width, several chain levels
            shape = (4 * int(rng.integers(1, 320000)),) if ragged else (32 * int(rng.integers(1, 40000)),)
        elif kind == 3:      # volumes
            shape = (int(rng.integers(1, 160)), int(rng.integers(1, 260)), 4 * int(rng.integers(1, 90))) if ragged else \
                    (32 * int(rng.integers(1, 5)), 32 * int(rng.integers(1, 8)), 4 * int(rng.integers(1, 90)))
        else:                # images: lane = line along x (lane = tile below 32 rows), lane = column along y
            shape = (int(rng.integers(1, 1500)), 4 * int(rng.integers(1, 560))) if ragged else (int(rng.integers(1, 1500)), 32 * int(rng.integers(1, 70)))
        ndim = len(shape)
        scans = []
        for d in range(ndim):
            n_d = shape[ndim - 1 - d]
            if (not ragged and n_d % 32 != 0) or (d > 0 and shape[-1] % 4 != 0):
                continue
            for _ in range(int(rng.integers(0, 3)) + (1 if d == 0 else 0)):
                k = int(rng.integers(1, 33))
                a = rng.standard_normal(k) * np.exp(-rng.uniform(0.05, 0.4) * np.arange(k))
                a *= rng.uniform(0.3, 0.95) / np.sum(np.abs(a))
                scans.append((d, bool(rng.integers(0, 2)), [float(rng.uniform(0.3, 1.5))] + [float(np.float32(v)) for v in a]))
        if shape[-1] % 32 != 0 and not ragged:
            scans = [s for s in scans if s[0] != 0] or [(ndim - 1, True, [1.0, 0.5])]
            if shape[0] % 32 != 0:
                continue
        clamped = bool(rng.integers(0, 2))
        planes = int(rng.integers(1, 4)) if kind else 1
        imgs = [torch.rand(shape, device="cuda") for _ in range(planes)]
        with rfa.Plan(shape, scans, clamped=clamped, planes=planes, path=5) as pm, \
                rfa.Plan(shape, scans, clamped=clamped, planes=planes, path=1, flags=rfa.capi.RF_PLAN_SERIAL_UNTILED) as pu:
            om, ou = pm.execute(imgs), pu.execute(imgs)
            torch.cuda.synchronize()
            err = 0.0
            for a_, b_ in zip(om, ou):
                peak = float(b_.abs().max().item())
                err = max(err, float(((a_ - b_).abs() / torch.clamp(b_.abs(), min=1e-2 * peak)).max().item()))
            worst = max(worst, err)
            orders = [len(s[2]) - 1 for s in scans]
            print(f"{case:3d} matrix {str(shape):22s} tiles={pm.tiles} orders={orders} clamped={int(clamped)} planes={planes} err={err:.3e}",
                  "" if err < 2e-4 else "  <-- CHECK", flush=True)
    print("worst", worst)
