"""Row-sharded execution over the GPUs of one node: one process per GPU, RCCL over xGMI.

The image is partitioned along its outermost dimension (y for 2-D, z for 3-D) into `world`
contiguous slabs of whole tiles.  Scans along inner dimensions are slab-local.  For every scan
along the sharded dimension the only cross-GPU dependency is the k-row carry at the slab
boundary; because the carry recurrence is linear with known k x k matrices per slab, each rank
publishes the exit carries of its slab computed with zero incoming carries and ONE all-gather
(for all scans of the dimension together; one per scan for orders > 3 or more than 4 scans) lets
every rank form its true incoming carries locally.  Slabs may have
different extents (`slab_extents`, the same list on every rank; `split_extent` makes one of whole
tiles): the bytes a rank exchanges do not depend on them.  The reference has no multi-device path;
this is the MI355X-native addition BASELINE.json asks for.

The driver is backend-agnostic: `engine` is anything with the stepping API of
recfilter_amd.Plan (begin / num_exchanges / exchange_bytes / exchange_local / exchange_apply /
finish) and `group` any torch.distributed process group (nccl == RCCL on ROCm; gloo in the CPU
tests, where the engine is a numpy stand-in living in tests/).
"""
from __future__ import annotations

from typing import Optional, Sequence

import numpy as np

from . import capi
from .plan import Plan


def split_extent(extent: int, world: int, granule: int = 64) -> list:
    """Slab extents for `world` ranks along a sharded dimension of `extent` samples: whole granules (tiles), as even
    as they come -- the first extent/granule % world ranks get one granule more.  `extent` must be a multiple of
    `granule` and hold at least one granule per rank."""
    if extent % granule or extent // granule < world:
        raise ValueError(f"cannot split {extent} into {world} slabs of whole {granule}-sample tiles")
    g, extra = divmod(extent // granule, world)
    return [(g + (1 if r < extra else 0)) * granule for r in range(world)]


class ShardedFilter:
    """`inflight` > 1 keeps that many executions in flight, each on its own HIP stream with its own plan (workspace)
    and exchange buffers: the latency-bound carry kernels and the all-gather of one execution then run beside the
    HBM-bound passes of another ("steps in flight").  Use `submit` + `drain` for that; `execute` stays
    the one-at-a-time call on the current stream."""

    def __init__(self, local_shape: Sequence[int], scans, clamped: bool = False, planes: int = 1,
                 rank: int = 0, world: int = 1, path: int = capi.RF_PATH_AUTO, dtype=np.float32,
                 tile=None, group=None, engine=None, inflight: int = 1, slab_extents: Optional[Sequence[int]] = None,
                 force_exchange: bool = False, flags: int = 0, collective=None):
        """force_exchange: with world == 1, still build the plan with the exchange structure (RF_PLAN_FORCE_EXCHANGE) and
        drive begin / exchange_local / all-gather / interior / exchange_apply / finish -- what every rank of an N-GPU run
        does, on a box with one GPU (the all-gather of one rank over RCCL is the identity).
        collective(gathered, send) -> work-or-None replaces torch.distributed.all_gather_into_tensor (timing probes that
        stand a delay kernel in for the collective); default: the process group's all-gather, asynchronous."""
        self.rank, self.world, self.group = int(rank), int(world), group
        self.force_exchange = bool(force_exchange) and int(world) == 1
        self.collective = collective
        flags = int(flags) | (capi.RF_PLAN_FORCE_EXCHANGE if self.force_exchange else 0)
        self.inflight = max(1, int(inflight))
        if engine is not None and self.inflight > 1:
            raise ValueError("inflight > 1 builds its own plans: pass no engine")

        def make():
            from . import plan as _plan
            return Plan(local_shape, scans, dtype=dtype, clamped=clamped, planes=planes, tile=tile, path=path,
                        shard_rank=rank, shard_world=world, shard_extents=slab_extents, flags=_plan.DEFAULT_FLAGS | flags)
        self.plans = [engine if engine is not None else make()]
        self.plans += [make() for _ in range(self.inflight - 1)]
        self.plan = self.plans[0]
        self._send = {}
        self._gathered = {}
        self._streams = None          # one per slot, created on the first submit
        self._next = 0

    def _buffers(self, key, plan, like):
        import torch
        if key not in self._send:
            nbytes = plan.exchange_bytes(key[1])
            self._send[key] = torch.empty(nbytes, dtype=torch.uint8, device=like.device)
            self._gathered[key] = torch.empty(nbytes * self.world, dtype=torch.uint8, device=like.device)
        return self._send[key], self._gathered[key]

    def _run(self, slot, inputs, outputs, stream=None, marks=None):
        """marks: a list that receives (phase, event) pairs recorded on the step's stream at the phase boundaries
        (profile_step below); None in normal operation."""
        plan = self.plans[slot]
        kw = {} if stream is None else {"stream": stream}       # (the numpy stand-in of the CPU tests has no streams)
        if self.world == 1 and not self.force_exchange:
            return plan.execute(inputs, outputs, **kw)

        def mark(phase):
            if marks is not None:
                import torch
                ev = torch.cuda.Event(enable_timing=True)
                ev.record(torch.cuda.current_stream(inputs[0].device))
                marks.append((phase, ev))
        mark("start")
        plan.begin(inputs, outputs, **kw)
        try:
            n_ex = plan.num_exchanges
            for i in range(n_ex):
                send, gathered = self._buffers((slot, i), plan, inputs[0])
                plan.exchange_local(i, send.data_ptr())
                mark("begin")                # pass 1, the slab-local carry stages, the exit carries
                # issued after what the current stream holds (the exit carries); asynchronous to what follows on it
                if self.collective is not None:
                    work = self.collective(gathered, send)
                else:
                    import torch.distributed as dist
                    work = dist.all_gather_into_tensor(gathered, send, group=self.group, async_op=True)
                if i == n_ex - 1 and getattr(plan, "has_interior", False):
                    plan.interior()          # exchange-independent work (a z-sharded volume's x/y stage) beside the collective
                mark("interior")
                if work is not None:
                    work.wait()              # the current stream waits for the gathered carries
                mark("exchange_wait")        # what the stream spent blocked on the collective BEYOND its own interior work
                plan.exchange_apply(i, gathered.data_ptr())
                mark("apply")
            plan.finish()
            mark("finish")
        except BaseException:
            # a collective that raised, a stepping call that failed: hand the execution instance back (rf_plan_abort), so
            # that the next execute of this thread starts afresh instead of finding the plan "begun"
            abort = getattr(plan, "abort", None)
            if abort is not None:
                abort()
            raise
        return outputs

    def profile_step(self, inputs, outputs):
        """One sharded execute on the current stream with HIP events at the phase boundaries: milliseconds the STREAM spent in
        begin (pass 1 + slab-local carries + exit carries), interior (the exchange-independent work enqueued beside the
        collective), exchange_wait (blocked on the all-gather beyond that), apply (entering carries), finish (final pass);
        allgather_bytes = bytes every rank receives per step.  Synchronises the device.  With several exchanges per step the
        phases of all of them add up.  A plain (unsharded, unforced) filter has no phases: {}."""
        import torch
        if self.world == 1 and not self.force_exchange:
            return {}
        marks = []
        self._run(0, inputs, outputs, marks=marks)
        torch.cuda.synchronize()
        out = {"begin_ms": 0.0, "interior_ms": 0.0, "exchange_wait_ms": 0.0, "apply_ms": 0.0, "finish_ms": 0.0}
        for (_, e0), (phase, e1) in zip(marks, marks[1:]):
            out[phase + "_ms"] += e0.elapsed_time(e1)
        plan = self.plans[0]
        out["allgather_bytes"] = int(sum(plan.exchange_bytes(i) for i in range(plan.num_exchanges)) * self.world)
        out["exchanges"] = int(plan.num_exchanges)
        return out

    def execute(self, inputs, outputs):
        """One filter execution on this rank's slab.  Asynchronous on the current stream for a
        GPU engine; the all-gathers run on the same stream (torch.distributed orders them)."""
        return self._run(0, inputs, outputs)

    def submit(self, inputs, outputs):
        """Like execute, on the next slot's stream (round robin over `inflight` slots): it starts after what the current
        stream holds now (the inputs are ready) and after the slot's previous execution, and runs beside the other
        slots.  Executions in flight at the same time need distinct output planes.  Call drain() before reading."""
        import torch
        if self.inflight == 1:
            return self._run(0, inputs, outputs)
        if self._streams is None:
            self._streams = [torch.cuda.Stream(device=inputs[0].device) for _ in range(self.inflight)]
        slot, self._next = self._next, (self._next + 1) % self.inflight
        st = self._streams[slot]
        st.wait_stream(torch.cuda.current_stream(inputs[0].device))
        with torch.cuda.stream(st):
            self._run(slot, inputs, outputs, stream=st)
        return outputs

    def drain(self):
        """The current stream waits for every execution submitted so far."""
        import torch
        if self._streams is not None:
            for st in self._streams:
                torch.cuda.current_stream(st.device).wait_stream(st)
