#!/usr/bin/env python3
"""bench.py -- headline benchmark of the tiled recursive-filter hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg3] [--size S]

A "step" is one execute of the whole filter (pass 1, carry stages, pass 2) over one synthetic
image already resident in HBM.  N=1 runs BASELINE.json configs[2] ("apps/gaussian: 2D order-2
causal+anticausal x/y, 16384^2 f32") -- the configuration the metric is quoted
on.  N>1 (launched by torch.distributed.run, one rank per GPU) shards the image by rows: every rank
owns a 16384-row slab of a (N*16384) x 16384 image (weak scaling) and the ranks exchange the k-row
boundary carry of the two y scans with one RCCL all-gather per scan.

Steps are submitted round robin to `--inflight` HIP streams, each with its own plan (workspace), exchange buffers and
output planes, so that one step's latency-bound carry kernels and its all-gather run beside another step's HBM-bound
passes; `value` is throughput over the K timed steps, `ms_per_step` its inverse.  Default: ONE step at a time on one
GPU (strictly one after the other on one stream: what `kernels_ms` adds up to and what the rocprofv3 summaries under
profiles/ show), two in flight on N > 1 GPUs, where the second step keeps the all-gather's latency off the critical
path (`config.steps_in_flight` says which; on one GPU two in flight are worth 2 %).

After the timed region of a sharded run the result is CHECKED: every rank gathers all input slabs, runs the unsharded plan
on the whole image and compares its slab of that with what the sharded protocol produced (`sharded_parity`, max over ranks;
above 1e-4 the bench exits non-zero).  A default run (no --workload / --size) also measures BASELINE config 5 -- 2048^3, six
order-2 scans, sharded along z over the N GPUs, strong scaling -- and carries its line, same keys, as `configs[0]`.

Prints ONE JSON line (rank 0).  `value` = Mpixels/s over all GPUs; `roofline` prices the dominant
kernel against the 8 TB/s HBM peak with HIP-event timing of that kernel; `cpu_baseline` is the CPU
oracle (a port of the reference's scan operator, OpenMP over all host cores) on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBPS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def workload(name, size, rows=0):
    import ref_cases as rc
    if name == "order12":
        # not a BASELINE config: the matrix path's reference point -- a 16384^2 f32 image through causal +
        # anticausal scans of ORDER 12 along x and y, clamped border, coefficients with sum |a| = 0.85 (tools/matrix_bench.py)
        import numpy as np
        rng = np.random.default_rng(3)
        a = rng.standard_normal(12) * np.exp(-0.15 * np.arange(12))
        c = [0.4] + [float(np.float32(v)) for v in a * 0.85 / np.abs(a).sum()]
        n = size or 16384
        return dict(shape=(n, n), dtype=np.float32, clamped=True, planes=1, scans=[(0, True, c), (0, False, c), (1, True, c), (1, False, c)])
    cfg = dict(rc.BASELINE_CONFIGS[{"cfg2": "cfg2_summed_table", "cfg3": "cfg3_gaussian2_xy",
                                    "cfg4a": "cfg4a_bicubic_rgb", "cfg4b": "cfg4b_gaussian3_rgb",
                                    "cfg5": "cfg5_generic_xyz"}[name]])
    if size:
        cfg["shape"] = tuple(size for _ in cfg["shape"])
    if rows:
        cfg["shape"] = (rows,) + tuple(cfg["shape"][1:])          # the outermost extent alone: one rank's slab of a strong-scaling run
    cfg.setdefault("planes", 1)
    return cfg


def algorithmic_bytes_per_launch(kernel_name, samples, itemsize):
    """Compulsory HBM bytes of one launch (DESIGN.md "roofline accounting"): the final pass reads
    every sample once and writes it once; pass 1 only reads; carry kernels touch no image bytes."""
    if "pass2" in kernel_name:
        return 2 * itemsize * samples
    if "pass1" in kernel_name or "tails" in kernel_name:      # (fused_tails: whichever pass-1 kernel the plan launches)
        return itemsize * samples
    return 0


def kernel_sources_sha16():
    """Identity of the code that is running: sha256 over the kernel and plan sources (recfilter_amd/csrc/*.hip|*.h|*.cpp,
    sorted by name), first 16 hex digits.  tools/pmc_summary.py stores the same digest in the PMC summary it writes, so a
    summary collected on other kernels than the ones running is recognised without git (the GPU box has no .git)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "recfilter_amd", "csrc")
    for path in sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h")) + glob.glob(os.path.join(csrc, "*.cpp"))):
        h.update(os.path.basename(path).encode() + b"\0")
        h.update(open(path, "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(kernel_name, workload, shape):
    """HBM bytes per launch of `kernel_name` from the committed rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE
    collected separately, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for 16 B/lane reads on gfx950).
    rocprofv3 cannot run inside the bench, so this is a LOOKUP in the newest committed summary, not a measurement of
    the run that prints it -- the bench line names the file (`traffic_source`), the commit it was collected at
    (`traffic_head`) and the digest of the kernel sources it was collected on.  `traffic` is null when that digest is
    not the digest of the sources that are running (a kernel changed since: the old bytes would be a guess), and for
    every workload and size without a committed summary of its own (cfg3 at 16384^2 and config 5 at 2048^3 have one).
    Returns (traffic, info dict)."""
    info = {"traffic_source": None, "traffic_head": None, "traffic_sources_sha16": None, "running_sources_sha16": kernel_sources_sha16()}
    # one summary per workload and size: the headline's is pmc_traffic.json, every other one pmc_traffic_<workload>_<extent>.json
    # (tools/refresh_profiles.sh collects cfg3 at 16384^2 and config 5 at 2048^3)
    if workload == "cfg3" and tuple(shape) == (16384, 16384):
        fname = "pmc_traffic.json"
    elif len(set(shape)) == 1:
        fname = f"pmc_traffic_{workload}_{shape[0]}.json"
    else:
        return None, info
    for rnd in ("r6", "r5", "r4", "r3", "r2", "r1"):
        rel = os.path.join("profiles", rnd, fname)
        try:
            doc = json.load(open(os.path.join(ROOT, rel)))
            table = doc["kernels"]
        except Exception:
            continue
        # step names carry the dimension and the pass ("strided_pass2_z"): match the kernel they launch ("strided_pass_kernel")
        stem = kernel_name.rstrip("_xyz+-0123456789")
        hits = [k for k in table if k.startswith(kernel_name)] or [k for k in table if k.startswith(stem)]
        for key in hits[:1]:
            entry = table[key]
            if True:
                info.update(traffic_source=rel, traffic_head=doc.get("git_head"), traffic_sources_sha16=doc.get("kernel_sources_sha16"))
                if doc.get("kernel_sources_sha16") != info["running_sources_sha16"]:
                    info["traffic_note"] = "stale: collected on other kernel sources than the ones running"
                    return None, info
                # HBM bytes of one whole step: every kernel's bytes per launch x its launches per step (the dominant
                # kernel launches once per step)
                per_step = entry["launches"]
                info["step_traffic"] = int(sum((e["fetch_bytes_corrected"] + e["write_bytes"]) * e["launches"] / per_step
                                               for e in table.values()))
                return entry["fetch_bytes_corrected"] + entry["write_bytes"], info
        break          # the newest summary has no such kernel: older rounds' bytes are not this code's
    return None, info


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(cfg, budget_px=1 << 28):      # 16384^2 = the whole cfg3 image (1 GiB); larger workloads are cropped
    """Times the CPU counterparts of the reference's two CPU schedules (oracle/, test infrastructure) on the same
    workload, all host cores available to this process:
      untiled  oracle.apply_filter        -- lib/recfilter.cpp:302-343 with OpenMP over lines, lines adjacent in memory
                                             side by side (cpu_auto_full_schedule, lib/recfilter.cpp:586-608)
      tiled    oracle.apply_filter_tiled  -- pass 1 / carry / pass 2 per dimension, tiles in parallel, tile 32
                                             (cpu_auto_intra/inter_schedule, lib/recfilter.cpp:610-678)
    both rebuilt with -O3 -march=native for this host.  `value` is the faster of the two."""
    import numpy as np
    import oracle
    native = oracle.use_native_build()
    shape = list(cfg["shape"])
    while int(np.prod(shape)) > budget_px:
        i = int(np.argmax(shape))
        shape[i] //= 2
    img = np.random.default_rng(1).random(tuple(shape), dtype=np.float32)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    nproc = os.cpu_count() or avail
    avail = max(1, min(oracle.max_threads(), avail))
    oracle.apply_filter(img[..., :64], cfg["scans"], cfg["clamped"], threads=avail)   # warm the library
    work = np.empty_like(img)
    results = {}
    px = float(np.prod(shape))
    # the box may expose more hardware threads than physical cores: take the best thread count
    counts = sorted({avail, max(1, avail // 2), max(1, avail // 4), min(avail, 32)}, reverse=True)
    for kind, fn in (("untiled", lambda th: oracle.apply_filter(work, cfg["scans"], cfg["clamped"], threads=th, inplace=True)),
                     ("tiled", lambda th: oracle.apply_filter_tiled(work, cfg["scans"], cfg["clamped"], tile=32, threads=th,
                                                                    inplace=True))):
        best, best_threads = None, avail
        for threads in counts:
            for _ in range(2):
                np.copyto(work, img)
                t0 = time.perf_counter()
                fn(threads)
                dt = time.perf_counter() - t0
                if best is None or dt < best:
                    best, best_threads = dt, threads
        results[kind] = {"value": round(px / best / 1e6, 2), "threads": best_threads, "seconds": round(best, 4)}
    top = max(results, key=lambda k: results[k]["value"])
    return {"value": results[top]["value"], "unit": "Mpixels/s", "cores": results[top]["threads"], "kind": "port",
            "schedule": top, "untiled": results["untiled"], "tiled": results["tiled"],
            "nproc": nproc, "threads_available": avail, "cpu_model": cpu_model(),
            "build": "-O3 -march=native -fopenmp" if native else "-O2 -fopenmp (native rebuild failed)",
            "sample": f"{'x'.join(map(str, shape))} f32, 1 plane of the workload"
                      f"{' (whole image)' if list(shape) == list(cfg['shape']) else ' (crop)'}, best of 2 runs over "
                      f"{' / '.join(map(str, counts))} threads, tile 32 for the tiled schedule"}


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N ranks with torch.distributed.run (one per GPU, rendezvous
    on 127.0.0.1) and return its exit code."""
    import socket
    import subprocess
    import torch
    have = torch.cuda.device_count()                 # (a count; the ranks are child processes either way, see main)
    if args.device < 0 and have < args.gpus:
        print(f"bench.py: --gpus {args.gpus} needs {args.gpus} visible devices, found {have}", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: what RCCL needs on this driver
    return subprocess.call(cmd, env=env)


def metric_name(workload_name, shape, planes):
    """BASELINE.json's metric string for the configuration it is quoted on (cfg3 at full size); other workloads and
    debug sizes say what they ran."""
    if workload_name == "cfg3" and tuple(shape) == (16384, 16384) and planes == 1:
        return "Mpixels/s + achieved HBM GB/s, 16384^2 order-2 x/y Gaussian IIR"
    what = {"cfg2": "order-1 summed-area table", "cfg3": "order-2 x/y Gaussian IIR", "cfg4a": "bicubic B-spline prefilter",
            "cfg4b": "order-3 x/y Gaussian IIR", "cfg5": "order-2 x/y/z filter (test_generic_xyz)",
            "order12": "order-12 x/y filter, direct form (not a BASELINE config: the matrix path's reference point)"}[workload_name]
    return f"Mpixels/s + achieved HBM GB/s, {'x'.join(map(str, shape))} x{planes} {what}"


def strict_rel_err(got, want):
    """SURVEY 8d's pointwise metric on device tensors: max |got - want| / max(|want|, 1e-6)."""
    import torch
    worst = 0.0
    rows = max(1, (1 << 26) // max(1, got[0].numel()))             # in pieces of 256 MiB: no slab-sized temporaries pile up
    for g, w in zip(got.split(rows), want.split(rows)):
        worst = max(worst, float(((g - w).abs() / w.abs().clamp_min(1e-6)).max().item()))
    return worst


def run_workload(args, dist, rank, world, name, size, strong, primary):
    """One bench line (a dict; meaningful on rank 0) for one workload: cold region, per-kernel pass, timed region, and -- for a
    sharded run -- the parity of the sharded result against the unsharded plan on the gathered image."""
    import numpy as np
    import torch
    import recfilter_amd as rfa
    from recfilter_amd.dist import ShardedFilter

    cfg = workload(name, size, getattr(args, "rows", 0) if primary else 0)
    shape, planes = cfg["shape"], cfg["planes"]
    global_shape = tuple(shape)
    if strong and world > 1:
        if shape[0] % (world * 64) != 0:
            raise SystemExit(f"--strong: outermost extent {shape[0]} is not a multiple of {world} slabs of whole tiles")
        shape = (shape[0] // world,) + tuple(shape[1:])
    else:
        global_shape = (shape[0] * world,) + tuple(shape[1:])       # weak scaling: every rank owns a full-size slab
    dtype = torch.float32
    gen = torch.Generator(device="cuda").manual_seed(1234 + rank)
    inputs = [torch.rand(shape, generator=gen, device="cuda", dtype=dtype) for _ in range(planes)]
    inflight = args.inflight if args.inflight > 0 else (1 if world == 1 else 2)
    # steps in flight at the same time write distinct output planes
    output_sets = [[torch.empty_like(t) for t in inputs] for _ in range(inflight)]
    outputs = output_sets[0]
    samples_local = int(np.prod(shape)) * planes

    stepping = world > 1 or (args.force_stepping and dist is not None)
    if args.force_stepping and dist is None:
        raise SystemExit("--force-stepping needs a process group: launch with torch.distributed.run --nproc-per-node=1")
    filt = ShardedFilter(shape, cfg["scans"], clamped=cfg["clamped"], planes=planes, rank=rank, world=world,
                         path=args.path, dtype=np.float32, group=None, inflight=inflight, force_exchange=stepping and world == 1)

    def barrier():
        filt.drain()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    def timed_region():
        """W untimed steps, then exactly K steps between barriers; max over ranks.  A step = one execute of the whole filter
        on the resident image; with --inflight D the steps are submitted round robin to D streams (recfilter_amd/dist.py)."""
        for i in range(args.warmup):
            filt.submit(inputs, output_sets[i % inflight])
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            filt.submit(inputs, output_sets[i % inflight])
        barrier()
        dt = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([dt], device="cuda", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    # --- cold: the driver's W + K steps on a GPU that has run nothing but the generation of the inputs (reported as
    # ms_per_step_cold / value_cold: the strict reading of the protocol; with the driver's 5 + 20 steps -- 15 ms of work -- the
    # GPU has not reached its steady clocks yet: 0.62 against 0.59 ms on cfg3) ------------------------------------------------
    elapsed_cold = timed_region()

    # --- per-kernel timing with HIP events on the launch stream (single-device plan) -----------------
    # Before the timed region proper, on every rank (rank 0 reports).  A GPU that starts from idle needs some 25 ms of load
    # before its kernels run at their steady durations (tools/probes/region_probe.py), so the first two thirds of this pass
    # are not recorded.  A rank whose own plan is sharded (or forced into the stepping protocol) times an unsharded plan of the
    # same slab, which has the same kernels.
    def per_kernel_pass():
        own_plan = world == 1 and not stepping
        # (an unsharded plan on the path the sharded one resolved to: the same kernels)
        plan = filt.plan if own_plan else rfa.Plan(shape, cfg["scans"], clamped=cfg["clamped"], planes=planes, path=filt.plan.path)
        reps = max(5, min(args.steps, 20))
        preheat = 3 * reps
        acc = {}
        order = []
        for i in range(3 * reps):
            _, times = plan.execute_timed(inputs, outputs)
            if i < 2 * reps:          # not recorded: the GPU is still on its way up (see above)
                continue
            for kname, ms in times:
                if kname not in acc:
                    acc[kname] = []
                    order.append(kname)
                acc[kname].append(ms)
        kernels = {n: float(np.mean(acc[n])) for n in order}        # ms per step, all planes
        # the dominant kernel among those that move image bytes (at toy sizes a launch-bound carry kernel can take longer)
        passes = [n for n in order if algorithmic_bytes_per_launch(n, 1, 4) > 0] or order
        dom = max(passes, key=lambda n: kernels[n])
        # the Tuple planes of a 2-D filter ride in one launch per step; otherwise one launch per plane
        batched = len(shape) == 2 and 1 < planes <= 16
        launches = 1 if batched else planes
        alg = algorithmic_bytes_per_launch(dom, samples_local // launches, 4)
        avg_ms = kernels[dom] / launches
        achieved = alg / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        traffic, traffic_info = pmc_traffic(dom, name, shape)
        roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4),
                    "traffic": traffic, **traffic_info,
                    "avg_launch_ms": round(avg_ms, 4), "algorithmic_bytes_per_launch": alg}
        # The measured ceiling of a read-once / write-once pass on this box (SURVEY 8d asks for one): the library's own
        # rf_stream_copy -- the final pass's access shape (256 x 128 tiles, 16-byte non-temporal loads and stores, eight loads
        # per thread in flight) with the arithmetic and the LDS taken out -- over one plane of this workload, HIP events on the
        # stream it runs on.  (Rounds 1-5 timed torch's copy_ here: 4.8 TB/s, BELOW what the final pass itself reaches.)
        # two_pass_ceiling_frac: an EXACT filter reads every sample twice and writes it once -- every output depends on every
        # input, and the image (1 GiB) does not fit the 256 MiB Infinity Cache -- so under the 8-bytes-per-sample accounting
        # no two-pass scheme can exceed 8/12 of the rate a copy reaches (a volume, whose z stage is a tiled pass of its own:
        # 8/20, min_bytes_per_sample); filter_frac_of_two_pass_ceiling (the line's top level) is the whole filter against that.
        plane_rows = (samples_local // planes) // shape[-1]
        if shape[-1] % 256 == 0 and plane_rows % 128 == 0:
            copy_ms = rfa.stream_copy_ms(inputs[0], outputs[0], reps=5)
            roofline["copy_ceiling_gbps"] = round(2 * 4 * (samples_local // planes) / (copy_ms * 1e-3) / 1e9, 1)
            roofline["copy_kernel"] = "rf_stream_copy: 256 x 128 tiles, 16-byte non-temporal loads and stores (this library)"
            # (a volume on this design: the one-read first pass 4 B + two tiled final passes of 8 B each = 20 B per sample;
            #  an image or a signal: 4 + 8 = 12)
            min_bytes = 20.0 if len(shape) == 3 else 12.0
            roofline["min_bytes_per_sample"] = min_bytes
            roofline["two_pass_ceiling_frac"] = round((8.0 / min_bytes) * roofline["copy_ceiling_gbps"] / HBM_PEAK_GBPS, 4)
        else:
            roofline["copy_ceiling_gbps"] = None
            roofline["two_pass_ceiling_frac"] = None
        if not own_plan:
            plan.close()
        return roofline, kernels, preheat

    roofline, kernels, preheat = per_kernel_pass()

    # --- steady: the same W + K steps once the GPU runs at its steady clocks (value, ms_per_step: the protocol of rounds 2
    # and 3, kept so that the rounds' headline numbers stay comparable; the cold figure is reported beside it) ---------------
    elapsed = timed_region()
    joined = dist.get_world_size() if dist is not None else 1      # ranks that actually took part
    # `value` / `ms_per_step` are the STRICT reading of the protocol since round 5: the first W + K steps this process ran
    # (VERDICT r4 weak 7); the figure at steady clocks -- what rounds 2-4 printed as `value` -- rides beside it
    # as value_steady / ms_per_step_steady.
    ms_per_step_steady = elapsed * 1000.0 / args.steps
    ms_per_step = elapsed_cold * 1000.0 / args.steps
    total_px = samples_local * joined
    value = total_px / (ms_per_step * 1e-3) / 1e6

    # --- sharded parity (outside the timed region): the result of the path that was just timed, against the UNSHARDED plan
    # on the whole image.  Every rank gathers all slabs of the input (rank-major along the outermost dimension = the global
    # image), filters it with an unsharded plan on the same path, and compares its own slab of that with what the sharded
    # protocol -- exit carries, all-gather, entering carries -- leaves in its output planes, after one more execute into
    # cleared planes.  MAX over ranks; above 1e-4 the bench fails (a wrong exchange must not print as a speed-up).
    parity = None
    parity_sat = None
    if stepping and not args.no_sharded_parity:
        if args.corrupt_exchange:
            # (test hook) what a broken exchange looks like: every rank receives carries that are not the ones that were sent
            def corrupted(gathered, send):
                dist.all_gather_into_tensor(gathered, send)
                gathered.view(torch.float32).mul_(1.25)
                return None
            filt.collective = corrupted
        for o in outputs:
            o.zero_()
        filt.execute(inputs, outputs)
        torch.cuda.synchronize()
        worst = 0.0
        lo = rank * shape[0]
        whole_in, whole_out = [], []
        for p in range(planes):
            w = torch.empty(global_shape, device="cuda", dtype=dtype)
            if joined > 1:
                dist.all_gather_into_tensor(w, inputs[p])
            else:
                w.copy_(inputs[p])
            whole_in.append(w)
            whole_out.append(torch.empty_like(w))
        with rfa.Plan(global_shape, cfg["scans"], clamped=cfg["clamped"], planes=planes, path=filt.plan.path) as whole_plan:
            whole_plan.execute(whole_in, whole_out)
            torch.cuda.synchronize()
        for p in range(planes):
            worst = max(worst, strict_rel_err(outputs[p], whole_out[p][lo:lo + shape[0]]))
        # The same check with a filter whose carries do NOT decay: one causal order-1 scan {1, 1} per dimension, the
        # summed-area table (apps/summed_table).  With the workload's own poles (0.79 .. 0.81) a carry from beyond the
        # neighbouring slab is below one ulp after 16384 rows (2048 / N planes), so a wrong multi-hop table -- A^M, the
        # cross-scan transfers X[q][s] over two or more slabs -- could not show in `sharded_parity` at bench size; an
        # integrator hands the full sum of EVERY slab before it to every slab, the last one's entering carry is the sum
        # over N - 1 slabs.  Same planes, same slabs, same exchange path; plane 0 only.
        sat_scans = [(d, True, [1.0, 1.0]) for d in range(len(shape))]
        sat = ShardedFilter(shape, sat_scans, clamped=False, planes=1, rank=rank, world=world, path=args.path, dtype=np.float32,
                            group=None, inflight=1, force_exchange=stepping and world == 1)
        if args.corrupt_exchange:
            sat.collective = filt.collective
        outputs[0].zero_()
        sat.execute([inputs[0]], [outputs[0]])
        torch.cuda.synchronize()
        with rfa.Plan(global_shape, sat_scans, clamped=False, planes=1, path=sat.plan.path) as whole_plan:
            whole_plan.execute([whole_in[0]], [whole_out[0]])
            torch.cuda.synchronize()
        worst_sat = strict_rel_err(outputs[0], whole_out[0][lo:lo + shape[0]])
        for pl in sat.plans:
            pl.close()
        del whole_in, whole_out, sat
        t = torch.tensor([worst, worst_sat], device="cuda", dtype=torch.float64)
        if dist is not None:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        parity, parity_sat = float(t[0].item()), float(t[1].item())
        for what, val in (("", parity), (" (summed-area table: non-decaying carries)", parity_sat)):
            if not (val <= SHARDED_PARITY_BAR):           # (also catches NaN)
                raise SystemExit(f"bench.py: sharded result differs from the unsharded plan on the gathered image{what}: "
                                 f"max rel err {val:.3e} > {SHARDED_PARITY_BAR:g} ({name}, {joined} rank(s))")

    # --- phases of one sharded step (outside the timed region): HIP events on the step's stream at the boundaries of the
    # stepping protocol, so that the first run on real xGMI says where a step's time goes -- begin (pass 1 + slab-local
    # carries), interior (the work enqueued beside the all-gather), exchange_wait (the stream blocked on the collective beyond
    # that), apply, finish.  Mean of 5 steps after 2 unrecorded ones; MAX over ranks per phase.
    phases = None
    if stepping:
        acc = {}
        for i in range(7):
            ph = filt.profile_step(inputs, outputs)
            if i >= 2:
                for k_, v_ in ph.items():
                    acc[k_] = acc.get(k_, 0.0) + v_ / 5.0
        keys = ["begin_ms", "interior_ms", "exchange_wait_ms", "apply_ms", "finish_ms"]
        t = torch.tensor([acc[k_] for k_ in keys], device="cuda", dtype=torch.float64)
        if dist is not None:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        phases = {k_: round(float(v_), 4) for k_, v_ in zip(keys, t.tolist())}
        phases["allgather_bytes"] = int(round(acc["allgather_bytes"]))
        phases["exchanges_per_step"] = int(round(acc["exchanges"]))

    whole = 8.0 * total_px / (ms_per_step * 1e-3) / 1e9      # 8 B per f32 sample per filter
    line = {
        "metric": metric_name(name, cfg["shape"], planes),
        "value": round(value, 1), "unit": "Mpixels/s", "n_gpus": joined, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
        # value / ms_per_step: the FIRST W + K steps this process ran (nothing but the generation of the inputs before them) --
        # the driver's protocol read strictly; value_steady / ms_per_step_steady: the same W + K steps once the GPU runs at its
        # steady clocks, behind that region and the per-kernel pass (preheat_executions executions, six device copies) -- the
        # figure rounds 2-4 printed as `value`.  (ms_per_step_cold / value_cold: kept as aliases of the strict figure.)
        "ms_per_step_steady": round(ms_per_step_steady, 4),
        "value_steady": round(total_px / (ms_per_step_steady * 1e-3) / 1e6, 1),
        "ms_per_step_cold": round(ms_per_step, 4), "value_cold": round(value, 1),
        "preheat_executions": preheat + args.warmup + args.steps, "preheat_copies": 6,
        "higher_is_better": True,
        "scaling": "strong" if (strong and world > 1) else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{name}: {'x'.join(map(str, shape))} f32 x{planes} plane(s) per GPU, "
                               f"{len(cfg['scans'])} scans, {'clamped' if cfg['clamped'] else 'zero'} border",
                   "global_shape": list(global_shape),
                   "path": filt.plan.path_name, "tiles": list(filt.plan.tiles),
                   "sharding": "rows (outermost dim), one all-gather per step" if world > 1 else
                               ("none (one rank driven through the sharded protocol: exit carries, all-gather, entering carries)"
                                if stepping else "none"),
                   "exchange": ("stepping" if stepping else "none"),
                   "collectives_per_step": (filt.plan.num_exchanges if stepping else 0),
                   "interior_beside_collective": bool(stepping and filt.plan.has_interior),
                   "steps_in_flight": inflight,
                   "backend": (args.backend if dist is not None else "none"),
                   "rccl_ranks": (joined if (dist is not None and args.backend == "nccl") else 0)},
        # max over ranks and samples of |sharded - unsharded| / max(|unsharded|, 1e-6), the unsharded plan run on the gathered
        # image; null when nothing was sharded (one rank, plain execute)
        "sharded_parity": parity, "sharded_parity_bar": SHARDED_PARITY_BAR,
        # the same comparison for a summed-area table over the same slabs (carries that do not decay: every slab's entering
        # carry is the sum over ALL slabs before it, so multi-hop exchange tables are numerically alive at bench size)
        "sharded_parity_sat": parity_sat,
        # one sharded step taken apart (HIP events on its stream, max over ranks; null when nothing was sharded): where the
        # time of a step goes -- a run on real xGMI whose exchange_wait_ms is not ~0 has a collective the interior work does
        # not cover
        "step_phases": phases,
        "filter_gbps": round(whole, 1), "filter_roofline_frac": round(whole / HBM_PEAK_GBPS / joined, 4),
        # the whole filter against what an exact two-pass scheme can reach at this box's measured copy rate (see roofline)
        "filter_frac_of_two_pass_ceiling": (round(whole / HBM_PEAK_GBPS / joined / roofline["two_pass_ceiling_frac"], 4)
                                            if roofline.get("two_pass_ceiling_frac") else None),
        "mibipixels_per_s": round(total_px * 1000.0 / (ms_per_step * 2 ** 20), 1),   # lib/timing.cpp:3-5
        "roofline": roofline,
        "kernels_ms": {k: round(v, 4) for k, v in kernels.items()},
    }
    if primary and rank == 0 and not args.no_cpu_baseline and world == 1:
        line["cpu_baseline"] = cpu_baseline(cfg)
    # leave the GPU as it was found: the next workload of this process needs the memory
    for pl in filt.plans:
        pl.close()
    del filt, inputs, output_sets, outputs
    torch.cuda.empty_cache()
    return line


SHARDED_PARITY_BAR = 1e-4       # north_star's float tolerance


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default=None, help="cfg2 | cfg3 | cfg4a | cfg4b | cfg5 | order12 (default: cfg3, the configuration the "
                    "metric is quoted on, followed by BASELINE config 5 -- 2048^3 sharded along z, strong scaling -- whose line "
                    "rides in the same JSON object under \"configs\")")
    ap.add_argument("--size", type=int, default=0, help="override every extent (debug)")
    ap.add_argument("--rows", type=int, default=0, help="override the outermost extent alone: `--workload cfg3 --rows 2048 --force-stepping` "
                    "under a one-rank launcher is ONE rank's share of `--gpus 8 --workload cfg3 --strong`, driven through the "
                    "sharded protocol on one GPU (tools/rehearse_n8.py)")
    ap.add_argument("--path", type=int, default=0, help="rf_path override (debug)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-configs", action="store_true", help="default run: only the headline workload, not config 5 behind it")
    ap.add_argument("--extra-size", type=int, default=0, help="override every extent of the config-5 line of a default run (debug)")
    ap.add_argument("--no-sharded-parity", action="store_true", help="skip the check of the sharded result against the unsharded "
                    "plan on the gathered image (it runs outside the timed region)")
    ap.add_argument("--corrupt-exchange", action="store_true", help="(test hook) scale the gathered carries of the parity execute: "
                    "the bench must then fail")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to exercise the "
                    "multi-rank path on a box with one GPU)")
    ap.add_argument("--device", type=int, default=-1, help="device ordinal for every rank (debug; default LOCAL_RANK)")
    ap.add_argument("--strong", action="store_true", help="strong scaling: the workload's shape is the GLOBAL image, every rank "
                    "owns 1/N of its outermost dimension (BASELINE config 5: 2048^3 z-sharded over 8 GPUs); default is weak "
                    "scaling, every rank owns a full-size slab")
    ap.add_argument("--force-stepping", action="store_true", help="one rank: drive the sharded protocol anyway (begin, exit carries, "
                    "all-gather over the process group, interior, entering carries, finish) on a plan built with "
                    "RF_PLAN_FORCE_EXCHANGE -- what every rank of an N-GPU run executes, on a box with one GPU")
    ap.add_argument("--inflight", type=int, default=0, help="steps in flight per GPU, each on its own HIP stream with its own "
                    "plan and output planes (1 = strictly one after the other on one stream; 0 = auto: 1 on one GPU, so "
                    "that per-kernel durations under rocprofv3 are those of kernels running alone, 2 on several GPUs, "
                    "where the second step hides the all-gather)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Not launched by torch.distributed.run: become the launcher.  This process only counts the devices, starts the
        # ranks as CHILD processes (nothing is exec'ed over it), waits and passes their exit code on.
        sys.exit(spawn_ranks(args))

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.device < 0 and torch.cuda.device_count() < world:
        raise SystemExit(f"--gpus {world} needs {world} visible devices, found {torch.cuda.device_count()}")
    device = local_rank if args.device < 0 else args.device
    torch.cuda.set_device(device)
    dist = None
    if world > 1 or "WORLD_SIZE" in os.environ:      # (a launcher with one rank still gets its process group: the
        import torch.distributed as dist            #  barrier and the reduction of the step time then run over RCCL)
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(args.backend)

    # The default run -- what a driver that only varies --gpus gets -- is the headline workload (cfg3; weak scaling on N
    # GPUs) AND north_star's config 5: 2048^3, six scans of order 2, sharded along z over the N GPUs (strong scaling, the
    # early exchange); its line, same keys, is the one entry of "configs".  Any explicit --workload / --size / --path /
    # --force-stepping runs exactly what it names.
    default_run = args.workload is None and not args.size and not args.rows and not args.path and not args.force_stepping and not args.strong
    name = args.workload or "cfg3"
    line = run_workload(args, dist, rank, world, name, args.size, args.strong, primary=True)
    if default_run and not args.no_extra_configs:
        if (args.extra_size or 2048) % (world * 64) == 0:
            extra = run_workload(args, dist, rank, world, "cfg5", args.extra_size, True, primary=False)
            extra["scaling"] = "strong"                  # (one GPU holds the whole volume: the N = 1 point of the strong curve)
            line["configs"] = [extra]
        else:
            line["configs_skipped"] = f"cfg5 --strong: 2048 planes do not split into {world} slabs of whole 64-plane tiles"
        # north_star's ">= 6x throughput at 8 GPUs" on the 16384^2 Gaussian reads as STRONG scaling: the headline image itself
        # split into N row slabs of whole tiles, beside the weak line above (every rank a full 16384-row slab).  On one GPU the
        # two are the same run.
        if world > 1 and 16384 % (world * 128) == 0:
            strong_line = run_workload(args, dist, rank, world, "cfg3", 0, True, primary=False)
            line.setdefault("configs", []).append(strong_line)
    if rank == 0:
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()                  # rank 0 was still timing single kernels: leave the group together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
