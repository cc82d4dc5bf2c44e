#!/usr/bin/env python3
"""Rehearsal of `bench.py --gpus 8`'s sharded-parity leg on ONE GPU (the pool has no multi-GPU box).

    python tools/rehearse_n8.py [--world 8] [--slab 16384] [--cfg5 2048]      (cfg3 weak, cfg3 strong, cfg5 strong)

cfg3, weak scaling: `world` slabs of slab x 16384 rows go through the stepping protocol (emulated ranks: one plan per rank
on this device, the all-gather a rank-major device buffer), and the concatenated result is compared with the UNSHARDED plan
on the whole (world * slab) x 16384 image -- the plan bench.py builds on every rank of an 8-GPU run (8 GiB in, 8 GiB out,
1024 tiles per column), which no other test builds at that size.  Then cfg5 --strong: 2048^3 in `world` z slabs against the
unsharded volume.  Prints one line per workload with bench.py's metric (max |a - b| / max(|b|, 1e-6))."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def emulate(name, local_shape, world, cfg):
    import torch
    import recfilter_amd as rfa
    from bench import strict_rel_err
    planes = cfg.get("planes", 1)
    assert planes == 1
    gen = torch.Generator(device="cuda").manual_seed(99)
    global_shape = (local_shape[0] * world,) + tuple(local_shape[1:])
    whole_in = torch.rand(global_shape, generator=gen, device="cuda", dtype=torch.float32)
    sharded_out = torch.empty_like(whole_in)
    ins = list(whole_in.split(local_shape[0]))
    outs = list(sharded_out.split(local_shape[0]))
    t0 = time.perf_counter()
    plans = [rfa.Plan(local_shape, cfg["scans"], clamped=cfg["clamped"], shard_rank=r, shard_world=world) for r in range(world)]
    nex = plans[0].num_exchanges
    gathered = [torch.empty(world * plans[0].exchange_bytes(e), dtype=torch.uint8, device="cuda") for e in range(nex)]

    def all_ranks_one_step():
        for r in range(world):
            plans[r].begin([ins[r]], [outs[r]])
        for e in range(nex):
            nbytes = plans[0].exchange_bytes(e)
            for r in range(world):
                plans[r].exchange_local(e, gathered[e].data_ptr() + r * nbytes)
            if e == nex - 1:
                for r in range(world):
                    plans[r].interior()
            for r in range(world):
                plans[r].exchange_apply(e, gathered[e].data_ptr())
        for r in range(world):
            plans[r].finish()

    all_ranks_one_step()
    torch.cuda.synchronize()
    # the device time of ONE rank's step: all `world` ranks' steps back to back on this GPU (the all-gather a device buffer that
    # the ranks' exchange_local launches fill), divided by `world` -- what a rank of the real run spends on its own kernels
    s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(2):
        all_ranks_one_step()
    s0.record()
    for _ in range(5):
        all_ranks_one_step()
    s1.record()
    torch.cuda.synchronize()
    rank_ms = s0.elapsed_time(s1) / 5 / world
    path, tiles = plans[0].path_name, list(plans[0].tiles)
    for p in plans:
        p.close()
    whole_out = torch.empty_like(whole_in)
    with rfa.Plan(global_shape, cfg["scans"], clamped=cfg["clamped"]) as plan:
        plan.execute([whole_in], [whole_out])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        plan.execute([whole_in], [whole_out])
        e1.record()
        torch.cuda.synchronize()
        whole_ms = e0.elapsed_time(e1)
        whole_path, whole_tiles = plan.path_name, list(plan.tiles)
    err = max(strict_rel_err(outs[r], whole_out.split(local_shape[0])[r]) for r in range(world))
    px = 1.0
    for n in global_shape:
        px *= n
    print(f"{name}: world {world}, slab {'x'.join(map(str, local_shape))} ({path}, tiles {tiles}), {nex} exchange(s); unsharded "
          f"{'x'.join(map(str, global_shape))} ({whole_path}, tiles {whole_tiles}) {whole_ms:.3f} ms; "
          f"one rank's step {rank_ms:.3f} ms of kernels (x {world} ranks in parallel = {px / rank_ms / 1e3:.0f} Mpixels/s before the "
          f"all-gather's latency; the unsharded plan: {px / whole_ms / 1e3:.0f}); "
          f"sharded_parity {err:.3e}  [{time.perf_counter() - t0:.1f} s]", flush=True)
    assert err < 1e-4
    del whole_in, whole_out, sharded_out
    torch.cuda.empty_cache()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--slab", type=int, default=16384)
    ap.add_argument("--cfg5", type=int, default=2048)
    args = ap.parse_args()
    import ref_cases as rc
    cfg3 = dict(rc.BASELINE_CONFIGS["cfg3_gaussian2_xy"])
    emulate("cfg3 weak", (args.slab, 16384), args.world, cfg3)
    # The same slabs with carries that do not decay (bench.py's `sharded_parity_sat`): with cfg3's poles (0.79) every carry
    # from beyond the neighbouring slab is below one ulp after 16384 rows, so the line above cannot see a wrong multi-hop
    # table (A^M, X[q][s] over two or more slabs); a summed-area table hands every slab the sum over ALL slabs before it.
    sat2 = {"scans": [(0, True, [1.0, 1.0]), (1, True, [1.0, 1.0])], "clamped": False}
    emulate("summed-area table weak", (args.slab, 16384), args.world, sat2)
    # north_star's ">= 6x at 8 GPUs" read as STRONG scaling: the 16384^2 image itself in `world` row slabs of whole 128-row tiles
    # (bench.py's configs[1] at N > 1): 2048-row slabs at world 8
    if 16384 % (args.world * 128) == 0:
        emulate("cfg3 strong", (16384 // args.world, 16384), args.world, cfg3)
        emulate("summed-area table strong", (16384 // args.world, 16384), args.world, sat2)
    if args.cfg5:
        cfg5 = dict(rc.BASELINE_CONFIGS["cfg5_generic_xyz"])
        emulate("cfg5 strong", (args.cfg5 // args.world, args.cfg5, args.cfg5), args.world, cfg5)
        sat3 = {"scans": [(d, True, [1.0, 1.0]) for d in range(3)], "clamped": False}
        emulate("summed-volume table strong", (args.cfg5 // args.world, args.cfg5, args.cfg5), args.world, sat3)


if __name__ == "__main__":
    main()
