"""Row-sharded execution over the GPUs of one node: one process per GPU, RCCL over xGMI.

The image is partitioned along its outermost dimension (y for 2-D, z for 3-D) into `world`
contiguous slabs of whole tiles.  Scans along inner dimensions are slab-local.  For every scan
along the sharded dimension the only cross-GPU dependency is the k-row carry at the slab
boundary; because the carry recurrence is linear with known k x k matrices per slab, each rank
publishes the exit carries of its slab computed with zero incoming carries and ONE all-gather
(for all scans of the dimension together; one per scan for orders > 3 or more than 4 scans) lets
every rank form its true incoming carries locally.  Slabs must have
equal extents.  The reference has no multi-device path; this is the MI355X-native addition
BASELINE.json asks for.

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


class ShardedFilter:
    def __init__(self, local_shape: Sequence[int], scans, clamped: bool = False, planes: int = 1,
                 rank: int = 0, world: int = 1, path: int = capi.RF_PATH_AUTO, dtype=np.float32,
                 tile=None, group=None, engine=None):
        self.rank, self.world, self.group = int(rank), int(world), group
        self.plan = engine if engine is not None else Plan(
            local_shape, scans, dtype=dtype, clamped=clamped, planes=planes, tile=tile, path=path,
            shard_rank=rank, shard_world=world)
        self._send = {}
        self._gathered = {}

    def _buffers(self, i: int, like):
        import torch
        if i not in self._send:
            nbytes = self.plan.exchange_bytes(i)
            self._send[i] = torch.empty(nbytes, dtype=torch.uint8, device=like.device)
            self._gathered[i] = torch.empty(nbytes * self.world, dtype=torch.uint8, device=like.device)
        return self._send[i], self._gathered[i]

    def execute(self, inputs, outputs):
        """One filter execution on this rank's slab.  Asynchronous on the current stream for a
        GPU engine; the all-gathers run on the same stream (torch.distributed orders them)."""
        if self.world == 1:
            return self.plan.execute(inputs, outputs)
        import torch.distributed as dist
        self.plan.begin(inputs, outputs)
        for i in range(self.plan.num_exchanges):
            send, gathered = self._buffers(i, inputs[0])
            self.plan.exchange_local(i, send.data_ptr())
            dist.all_gather_into_tensor(gathered, send, group=self.group)
            self.plan.exchange_apply(i, gathered.data_ptr())
        self.plan.finish()
        return outputs
