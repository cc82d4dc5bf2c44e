"""Two real processes drive the product's GPU plans through recfilter_amd.dist.ShardedFilter.

The pool's GPU boxes have one device, so both ranks share cuda:0 and the process group is gloo (RCCL refuses two
ranks on one device); everything else -- the HIP kernels, the stepping API, the exchange buffers, the all-gather
of device tensors -- is the code path `bench.py --gpus N` runs over RCCL."""
import os
import socket
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, shape, scans, clamped, planes, result_dir, extents=None, flags=0, expect_path="tiled_fused"):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    import torch
    import torch.distributed as dist
    import oracle
    import ref_cases as rc
    from recfilter_amd.dist import ShardedFilter

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        full = [rc.random_image(shape, np.float32, 91 + p) for p in range(planes)]
        ext = list(extents) if extents else [shape[0] // world] * world        # slabs may differ (shard_extents)
        lo, n = sum(ext[:rank]), ext[rank]
        local = (n,) + tuple(shape[1:])
        inputs = [torch.from_numpy(np.ascontiguousarray(f[lo:lo + n])).cuda() for f in full]
        outputs = [torch.empty_like(t) for t in inputs]
        filt = ShardedFilter(local, scans, clamped=clamped, planes=planes, rank=rank, world=world, slab_extents=extents, flags=flags)
        assert filt.plan.path_name == expect_path
        if flags:      # the slab's begin step is the one-read pass 1: its table of z responses exists
            assert filt.plan.table("H_z").size > 0 and filt.plan.has_interior
        for _ in range(2):                     # the second execute reuses the exchange buffers
            filt.execute(inputs, outputs)
        torch.cuda.synchronize()
        wants = [oracle.apply_filter(full[p].astype(np.float64), scans, clamped)[lo:lo + n] for p in range(planes)]
        for p in range(planes):
            err = rc.rel_err(outputs[p].cpu().numpy(), wants[p])
            assert err < 1e-4, f"rank {rank} plane {p}: rel err {err}"
        # steps in flight: two slots (own stream, plan, exchange buffers, output planes), five submits, one drain
        piped = ShardedFilter(local, scans, clamped=clamped, planes=planes, rank=rank, world=world, inflight=2, slab_extents=extents, flags=flags)
        sets = [[torch.zeros_like(t) for t in inputs] for _ in range(2)]
        for i in range(5):
            piped.submit(inputs, sets[i % 2])
        piped.drain()
        torch.cuda.synchronize()
        for k in range(2):
            for p in range(planes):
                err = rc.rel_err(sets[k][p].cpu().numpy(), wants[p])
                assert err < 1e-4, f"rank {rank} slot {k} plane {p}: rel err {err}"
        open(os.path.join(result_dir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case", ["rows_2d", "z_slabs_3d", "rows_2d_unequal", "z_slabs_3d_one_read_pass1", "z_slabs_3d_one_read_unequal",
                                  "rows_2d_order_12_matrix_path"])
def test_two_processes_one_gpu(case, tmp_path):
    import torch.multiprocessing as mp
    sys.path.insert(0, HERE)
    import ref_cases as rc
    extents, flags, expect_path = None, 0, "tiled_fused"
    if case == "z_slabs_3d_one_read_pass1":
        # whole tiles (256 x 32 x 64 planes per slab): the slabs' begin step forms the x, y and z tails in one read
        from recfilter_amd import capi
        shape, scans, clamped, planes, flags = (128, 64, 256), rc.BASELINE_CONFIGS["cfg5_generic_xyz"]["scans"], True, 1, capi.RF_PLAN_WALK_PASS1
    elif case == "z_slabs_3d_one_read_unequal":
        from recfilter_amd import capi
        shape, scans, clamped, planes, flags, extents = (96, 64, 256), rc.BASELINE_CONFIGS["cfg5_generic_xyz"]["scans"], False, 1, capi.RF_PLAN_WALK_PASS1, [64, 32]
    elif case == "rows_2d_order_12_matrix_path":
        # orders above 8 shard on the matrix path: one exchange per y scan, the ranks' exits chained with A^M on the matrix cores
        rng = np.random.default_rng(12)
        a = rng.standard_normal(12) * np.exp(-0.15 * np.arange(12))
        a *= 0.9 / np.abs(a).sum()
        c12 = [0.4] + [float(np.float32(v)) for v in a]
        shape, scans, clamped, planes, expect_path = (256, 512), [(0, True, c12), (0, False, c12), (1, True, c12), (1, False, c12)], True, 2, "tiled_matrix"
    elif case == "rows_2d":
        shape, scans, clamped, planes = (256, 768), rc.xy_pm(rc.GAUSS2), True, 2
    elif case == "rows_2d_unequal":
        shape, scans, clamped, planes, extents = (320, 768), rc.xy_pm(rc.GAUSS2), True, 2, [192, 128]
    else:
        shape, scans, clamped, planes = (32, 96, 256), rc.BASELINE_CONFIGS["cfg5_generic_xyz"]["scans"], False, 1
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, shape, scans, clamped, planes, str(tmp_path), extents, flags, expect_path), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / f"ok{r}") for r in range(world))


@pytest.mark.parametrize("launcher", ["self", "torchrun"])
def test_bench_two_ranks_over_gloo(launcher):
    """bench.py's multi-rank flow (rendezvous, sharded steps with one all-gather each, barriers, max-over-ranks
    timing, one JSON line from rank 0) with two ranks on this box's one GPU.  "self": `python bench.py --gpus 2` with
    no launcher around it -- bench.py starts its own ranks as child processes before anything touches the GPU;
    "torchrun": the way the driver launches it.  The driver runs the same script over RCCL."""
    import json
    import subprocess
    tail = [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--backend", "gloo", "--device", "0",
            "--size", "1024"]
    if launcher == "self":
        cmd = [sys.executable] + tail
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port())] + tail
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0 and d["config"]["path"] == "tiled_fused"
    assert d["config"]["backend"] == "gloo" and d["config"]["rccl_ranks"] == 0
    assert "all-gather" in d["config"]["sharding"] and d["roofline"]["kernel"] in ("fused_pass2", "fused_tails")     # (the dominant one; a toss-up at this size)
    # the sharded result was checked against the unsharded plan on the gathered 2048 x 1024 image
    assert d["config"]["global_shape"] == [2048, 1024] and 0 <= d["sharded_parity"] < 1e-4 and "configs" not in d
    # the step taken apart with HIP events on its stream: every phase present, the exit carries of both y scans counted
    assert 0 <= d["sharded_parity_sat"] < 1e-4          # ... and for a summed-area table over the same slabs (non-decaying carries)
    ph = d["step_phases"]
    assert all(ph[k] >= 0 for k in ("begin_ms", "interior_ms", "exchange_wait_ms", "apply_ms", "finish_ms"))
    assert ph["begin_ms"] > 0 and ph["finish_ms"] > 0 and ph["exchanges_per_step"] == 1
    assert ph["allgather_bytes"] == 2 * (2 * 2 * 1024 * 4)        # two ranks x (two scans x order 2 x 1024 columns x 4 bytes)


def test_bench_fails_on_a_corrupted_exchange():
    """A wrong exchange must not print as a speed-up: with the gathered carries of the parity execute scaled by 1.25 the bench
    exits non-zero and prints no JSON line, for row shards (the correction inside pass 2) and for z slabs (early exchange)."""
    import subprocess
    for extra in (["--size", "1024"], ["--workload", "cfg5", "--size", "256", "--strong"]):
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--backend", "gloo", "--device", "0",
               "--corrupt-exchange"] + extra
        env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
        assert out.returncode != 0, out.stdout[-2000:]
        assert "differs from the unsharded plan" in out.stderr
        assert not [l for l in out.stdout.splitlines() if l.startswith("{")]


def test_bench_default_run_carries_config5_strong():
    """What a driver that only varies --gpus gets: the headline line (cfg3, weak) with north_star's config 5 -- the volume
    sharded along z, strong scaling, early exchange -- as configs[0] and the headline image in strong scaling (16384^2 split
    into N row slabs) as configs[1], all checked against the unsharded plan.  Two ranks on
    this box's one GPU over gloo; config 5 shrunk to 256^3 (two ranks of the full volume do not fit one device)."""
    import json
    import subprocess
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--backend", "gloo", "--device", "0",
           "--extra-size", "256"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["global_shape"] == [32768, 16384]
    assert d["metric"].startswith("Mpixels/s + achieved HBM GB/s, 16384^2 order-2") and 0 <= d["sharded_parity"] < 1e-4
    e, s = d["configs"]
    assert e["scaling"] == "strong" and e["n_gpus"] == 2 and e["config"]["global_shape"] == [256, 256, 256]
    assert e["config"]["interior_beside_collective"] is True and 0 <= e["sharded_parity"] < 1e-4 and e["value"] > 0
    # ... and the headline image itself split into row slabs (north_star's ">= 6x at 8 GPUs" read as strong scaling): same keys
    assert s["scaling"] == "strong" and s["n_gpus"] == 2 and s["config"]["global_shape"] == [16384, 16384]
    assert "8192x16384" in s["config"]["workload"] and s["metric"] == d["metric"] and s["value"] > 0
    assert 0 <= s["sharded_parity"] < 1e-4 and 0 <= s["sharded_parity_sat"] < 1e-4
    assert s["step_phases"]["exchanges_per_step"] == 1 and s["step_phases"]["allgather_bytes"] == 2 * (2 * 2 * 16384 * 4)
    assert d["roofline"]["copy_kernel"].startswith("rf_stream_copy") and 0.05 < d["roofline"]["two_pass_ceiling_frac"] < 0.67      # (two ranks share this box's one GPU: each copy sees half of it)


def test_bench_strong_scaling_volume_two_ranks_over_gloo():
    """`bench.py --gpus 2 --workload cfg5 --strong`: the z-sharded volume with the early exchange (x/y stage beside the
    all-gather) in the bench's own flow, two ranks on this box's one GPU over gloo."""
    import json
    import subprocess
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--backend", "gloo", "--device", "0",
           "--workload", "cfg5", "--size", "256", "--strong"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    c = d["config"]
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0 and c["path"] == "tiled_fused"
    assert c["exchange"] == "stepping" and c["collectives_per_step"] == 1 and c["interior_beside_collective"] is True
    assert d["step_phases"]["interior_ms"] > 0 and d["step_phases"]["allgather_bytes"] == 2 * (2 * 2 * 256 * 256 * 4)
    assert "128x256x256" in c["workload"] and c["global_shape"] == [256, 256, 256] and 0 <= d["sharded_parity"] < 1e-4


def test_bench_refuses_more_ranks_than_devices():
    """`--gpus N` on a box with fewer than N devices must fail loudly, not benchmark one GPU and print n_gpus = 1."""
    import subprocess
    import torch
    n = torch.cuda.device_count() + 1
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert out.returncode != 0 and "visible devices" in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]


_RCCL_CALLS = r"""
import os, sys, torch, torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[1], RANK="0", WORLD_SIZE="1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
assert dist.get_backend() == "nccl"
# the exchange of ShardedFilter._run: byte buffers, all_gather_into_tensor issued under a side stream, twice in flight
streams = [torch.cuda.Stream() for _ in range(2)]
torch.manual_seed(212)
send = [torch.randint(0, 256, (1 << 20,), dtype=torch.uint8, device="cuda") for _ in range(2)]
got = [torch.zeros(1 << 20, dtype=torch.uint8, device="cuda") for _ in range(2)]
for it in range(4):
    st = streams[it % 2]
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        send[it % 2].add_(1)                                   # work queued ahead of the collective on the same stream
        dist.all_gather_into_tensor(got[it % 2], send[it % 2])
        got[it % 2].add_(0)                                    # ... and behind it
for st in streams:
    torch.cuda.current_stream().wait_stream(st)
torch.cuda.synchronize()
assert all(torch.equal(g, s) for g, s in zip(got, send))
# bench.py's timing reduction and barrier
t = torch.tensor([1.5], device="cuda", dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
torch.cuda.synchronize()
assert float(t.item()) == 1.5 and dist.get_world_size() == 1
dist.destroy_process_group()
print("rccl-ok")
"""


def test_rccl_calls_single_rank():
    """The torch.distributed calls of the sharded driver and of bench.py, as they make them, on the REAL backend
    (nccl = RCCL) with the one rank this box allows: process group bound to the device, all-gather of uint8 exchange
    buffers issued under side streams with work queued on either side, the float64 MAX all-reduce of the step time,
    barrier.  Two ranks over RCCL need two devices; their arithmetic is what the gloo tests above cover."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    out = subprocess.run([sys.executable, "-c", _RCCL_CALLS, str(_free_port())], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0 and "rccl-ok" in out.stdout, out.stderr[-3000:]


_RCCL_SHARDED = r"""
import os, sys
import numpy as np
import torch, torch.distributed as dist
ROOT, HERE = sys.argv[2], sys.argv[3]
sys.path.insert(0, ROOT); sys.path.insert(0, HERE)
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[1], RANK="0", WORLD_SIZE="1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
import oracle
import ref_cases as rc
from recfilter_amd import capi
from recfilter_amd.dist import ShardedFilter
issued = [0]
real = dist.all_gather_into_tensor
def counting(*a, **k):
    issued[0] += 1
    return real(*a, **k)
dist.all_gather_into_tensor = counting
cases = [("rows_2d", (512, 1024), rc.xy_pm(rc.GAUSS2), True, 2, False),
         ("z_slab_3d", (64, 96, 512), rc.BASELINE_CONFIGS["cfg5_generic_xyz"]["scans"], False, 1, True)]
for name, shape, scans, clamped, planes, interior in cases:
    imgs = [rc.random_image(shape, np.float32, 400 + p) for p in range(planes)]
    dev = [torch.from_numpy(i).cuda() for i in imgs]
    want = [oracle.apply_filter(i.astype(np.float64), scans, clamped) for i in imgs]
    for inflight in (1, 2):
        filt = ShardedFilter(shape, scans, clamped=clamped, planes=planes, rank=0, world=1, force_exchange=True,
                             inflight=inflight, flags=capi.RF_PLAN_TILED_ONLY)
        assert filt.plan.path_name == "tiled_fused" and filt.plan.num_exchanges == 1 and filt.plan.has_interior == interior, name
        before = issued[0]
        sets = [[torch.zeros_like(t) for t in dev] for _ in range(inflight)]
        for i in range(2 * inflight + 1):
            filt.submit(dev, sets[i % inflight])
        filt.drain()
        torch.cuda.synchronize()
        assert issued[0] - before == 2 * inflight + 1, "one RCCL all-gather per execute"
        for outs in sets:
            for o, w in zip(outs, want):
                err = rc.rel_err(o.cpu().numpy(), w)
                assert err < 1e-4, (name, inflight, err)
dist.barrier()
dist.destroy_process_group()
print("rccl-sharded-ok", issued[0])
"""


def test_sharded_protocol_over_rccl_one_rank():
    """recfilter_amd.dist.ShardedFilter._run -- begin, exit carries, the all-gather ISSUED ON RCCL (backend nccl, bound
    to the device, asynchronous), the exchange-independent work beside it, entering carries, final pass -- on plans built
    with RF_PLAN_FORCE_EXCHANGE, with the one rank this box allows.  A row-sharded 2-D image (merged exchange, the
    correction inside pass 2) and a z-sharded volume (early exchange: x/y stage beside the collective), one and two
    executes in flight; against the oracle.  A fresh child process: it must own the GPU from its first HIP call."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    out = subprocess.run([sys.executable, "-c", _RCCL_SHARDED, str(_free_port()), ROOT, HERE], capture_output=True, text=True,
                         timeout=900, env=env)
    assert out.returncode == 0 and "rccl-sharded-ok" in out.stdout, (out.stdout[-2000:], out.stderr[-3000:])


def test_bench_forced_stepping_one_rank_over_rccl():
    """`python -m torch.distributed.run --nproc-per-node=1 bench.py --gpus 1 --force-stepping`: the bench's timed steps go
    through the sharded protocol with its all-gather on RCCL (config.exchange == "stepping", one collective per step)."""
    import json
    import subprocess
    for extra in (["--size", "2048"], ["--workload", "cfg5", "--size", "256"]):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
               "--no-cpu-baseline", "--force-stepping"] + extra
        env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
        assert out.returncode == 0, out.stderr[-3000:]
        lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1
        d = json.loads(lines[0])
        c = d["config"]
        assert d["n_gpus"] == 1 and c["backend"] == "nccl" and c["rccl_ranks"] == 1 and d["value"] > 0
        assert c["exchange"] == "stepping" and c["collectives_per_step"] == 1 and c["path"] == "tiled_fused"
        assert c["interior_beside_collective"] == ("cfg5" in extra)
        assert 0 <= d["sharded_parity"] < 1e-4          # one rank through the protocol against the plain plan
        assert d["ms_per_step_cold"] > 0 and d["value_cold"] > 0 and d["preheat_executions"] >= 3 * 5


def test_bench_one_rank_over_rccl():
    """bench.py launched the driver's way (`python -m torch.distributed.run … bench.py --gpus 1`) with the default backend:
    process group on nccl (= RCCL) bound to the device, barriers and the max-over-ranks reduction through it."""
    import json
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
           "--size", "2048", "--no-cpu-baseline"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["config"]["backend"] == "nccl" and d["config"]["rccl_ranks"] == 1 and d["value"] > 0
