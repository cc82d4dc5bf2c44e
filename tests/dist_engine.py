"""A numpy stand-in for recfilter_amd.Plan's stepping API, for the CPU (gloo) tests of the sharded driver.

It implements the SAME protocol the C ABI exposes (rf_plan_begin / exchange_local / exchange_apply /
finish, include/recfilter_amd.h) and the SAME algebra kernels_generic.hip runs for the sharded
(outermost) dimension -- pass 1 with zero incoming carry, per-scan chaining + slab-local recurrence,
the slab's exit tail as the exchange payload, incoming = sum over preceding slabs of
(A^M)^(distance-1) * exit, propagation A^(t+1) * incoming, pass 2 -- with the W and A tables taken
from a host-only product plan.  Test infrastructure only; it lets `ShardedFilter` (recfilter_amd/dist.py)
run end to end over gloo without a GPU.
"""
from __future__ import annotations

import ctypes

import numpy as np

import recfilter_amd as rfa
from recfilter_amd import capi
from tiled_emulator import emulate_dimension, scan_tile


def _view(ptr: int, n: int) -> np.ndarray:
    buf = (ctypes.c_float * n).from_address(ptr)
    return np.frombuffer(buf, dtype=np.float32)


class NumpySlabEngine:
    def __init__(self, local_shape, scans, clamped, planes, rank, world, tile=None, slab_extents=None, early=False,
                 force_exchange=False):
        """early=True: the "early exchange" of a sharded outermost dimension (recfilter_amd/csrc/plan_strided.h): the
        carries of the RAW slab are exchanged, the inner dimensions are filtered beside the all-gather (`interior`), and
        the completed carry planes are filtered along the inner dimensions afterwards -- the operators of the outermost
        dimension commute with the filter of the inner ones."""
        self.early = bool(early)
        self.shape, self.scans, self.clamped = tuple(local_shape), list(scans), clamped
        self.planes, self.rank, self.world = planes, rank, world
        nd = len(self.shape)
        if tile is None:
            tile = [0] * nd
        self.plan = rfa.Plan(self.shape, scans, dtype=np.float64, clamped=clamped, planes=planes, tile=tile,
                             path=capi.RF_PATH_TILED_GENERIC, device=capi.RF_DEVICE_HOST_ONLY,
                             shard_rank=rank, shard_world=world, shard_extents=slab_extents,
                             flags=capi.RF_PLAN_FORCE_EXCHANGE if (force_exchange and world == 1) else 0)
        self.slab_extents = list(slab_extents) if slab_extents is not None else [self.shape[0]] * world
        self.tiles = self.plan.tiles
        self.outer = nd - 1
        self.outer_scans = [(bool(c), [float(np.float32(v)) for v in co]) for d, c, co in scans if d == self.outer]
        self.n = len(self.outer_scans)
        self.k = max([len(co) - 1 for _, co in self.outer_scans], default=0)
        if self.n:
            name = "xyz"[self.outer]
            self.W = self.plan.table("W_" + name).reshape(4, self.n, self.n, self.k, self.k)
            self.A = self.plan.table("A_" + name).reshape(self.n, self.k, self.k)
        self.N = self.shape[0]                      # numpy axis 0 is the outermost dimension
        self.T = self.tiles[self.outer] if self.n else self.N
        self.M = self.N // self.T if self.n else 1
        self.lines = int(np.prod(self.shape[1:])) if nd > 1 else 1
        # merged exchange (one all-gather for all scans): the plan publishes the cross-scan transfers Y / X
        self.merged = False
        if self.n and (world > 1 or force_exchange):
            try:
                name = "xyz"[self.outer]
                self.Y = self.plan.table("Y_" + name).reshape(self.n, self.n, self.M, self.k, self.k)
                self.X = self.plan.table("X_" + name).reshape(world, self.n, self.n, self.k, self.k)      # one per slab
                self.merged = True
            except Exception:
                self.merged = False

    # ---- protocol ---------------------------------------------------------------------------
    @property
    def num_exchanges(self):
        return 1 if self.merged else self.n

    def exchange_bytes(self, i):
        return self.planes * (self.n if self.merged else 1) * self.k * self.lines * 4

    def _first(self, s, t):
        return t == 0 if self.outer_scans[s][0] else t == self.M - 1

    def _border(self, s, t):
        c = self.outer_scans[s][0]
        return (t == 0 and self.rank == 0) if c else (t == self.M - 1 and self.rank == self.world - 1)

    def _variant(self, t):
        return (1 if (t == 0 and self.rank == 0) else 0) | (2 if (t == self.M - 1 and self.rank == self.world - 1) else 0)

    def _coef(self, s):
        co = self.outer_scans[s][1]
        return co[0], list(co[1:]) + [0.0] * (self.k - len(co) + 1)

    def _carry_into(self, pl, s, t):
        if self._first(s, t):
            return [self.incoming[pl][s][j] for j in range(self.k)]
        tp = t - 1 if self.outer_scans[s][0] else t + 1
        return [self.tails[pl][s, tp, j] for j in range(self.k)]

    def _inner(self, img):
        """the slab-local filter of the inner dimensions on an array whose LAST axes are the slab's inner axes (same
        tables as the product plan)"""
        nd = len(self.shape)
        for d in range(nd - 1):
            dim_scans = [(c, co) for (dd, c, co) in self.scans if dd == d]
            if not dim_scans:
                continue
            axis = img.ndim - 1 - d
            moved = np.moveaxis(img, axis, -1)
            flat = np.ascontiguousarray(moved).reshape(-1, moved.shape[-1])
            name = "xyz"[d]
            res = emulate_dimension(flat, dim_scans, self.tiles[d], self.clamped,
                                    self.plan.table("W_" + name), self.plan.table("A_" + name))
            img = np.moveaxis(res.reshape(moved.shape), -1, axis)
        return img

    @property
    def has_interior(self):
        return self.early and self.n > 0

    def interior(self):
        """exchange-independent work: the inner dimensions of the slab (early exchange only)"""
        if self._interior_done:
            return
        self._interior_done = True
        self.data = [np.ascontiguousarray(self._inner(d.T.reshape(self.shape))).reshape(self.N, self.lines).T.copy()
                     for d in self.data]

    def begin(self, inputs, outputs, stream=None):
        self.outputs = outputs
        self.data = []
        self._interior_done = not self.has_interior
        for pl in range(self.planes):
            img = inputs[pl].numpy().astype(np.float64)
            if not self.has_interior:
                img = self._inner(img)          # inner dimensions first; the exchange then carries filtered data
            self.data.append(np.ascontiguousarray(img).reshape(self.N, self.lines).T.copy())   # [lines, N]
        if not self.n:
            return
        k, M, T = self.k, self.M, self.T
        self.tails = [np.zeros((self.n, M, k, self.lines)) for _ in range(self.planes)]
        self.incoming = [[[np.zeros(self.lines) for _ in range(k)] for _ in range(self.n)] for _ in range(self.planes)]
        for pl in range(self.planes):
            for t in range(M):
                v = self.data[pl][:, t * T:(t + 1) * T].copy()
                for s in range(self.n):
                    b, a = self._coef(s)
                    c = self.outer_scans[s][0]
                    scan_tile(v, c, b, a, k, self.clamped and self._border(s, t))
                    for r in range(k):
                        p = T - 1 - r
                        self.tails[pl][s, t, r] = v[:, p if c else T - 1 - p]

    def exchange_local(self, s, send_ptr):
        if self.merged:
            return self._merged_local(send_ptr)
        k, M = self.k, self.M
        send = _view(send_ptr, self.planes * k * self.lines).reshape(self.planes, k, self.lines)
        self._local_scan(s, send)

    def _merged_local(self, send_ptr):
        k = self.k
        send = _view(send_ptr, self.planes * self.n * k * self.lines).reshape(self.planes, self.n, k, self.lines)
        for pl in range(self.planes):
            for s in range(self.n):
                for j in range(k):
                    self.incoming[pl][s][j] = np.zeros(self.lines)
        for s in range(self.n):
            self._local_scan(s, send[:, s])

    def _local_scan(self, s, send):
        k, M = self.k, self.M
        causal = self.outer_scans[s][0]
        for pl in range(self.planes):
            prev = None
            for i in range(M):
                t = i if causal else M - 1 - i
                cur = [self.tails[pl][s, t, r].copy() for r in range(k)]
                for q in range(s):
                    c = self._carry_into(pl, q, t)
                    for r in range(k):
                        for o in range(k):
                            cur[r] = cur[r] + self.W[self._variant(t), q, s, r, o] * c[o]
                if i > 0:
                    for r in range(k):
                        for j in range(k):
                            cur[r] = cur[r] + self.A[s, r, j] * prev[j]
                for r in range(k):
                    self.tails[pl][s, t, r] = cur[r]
                prev = cur
            for r in range(k):
                send[pl, r] = prev[r]

    def _merged_apply(self, gathered_ptr):
        k, M, n, W = self.k, self.M, self.n, self.world
        gathered = _view(gathered_ptr, W * self.planes * n * k * self.lines).reshape(
            W, self.planes, n, k, self.lines).astype(np.float64)
        for pl in range(self.planes):
            ins = np.zeros((n, W, k, self.lines))              # carry entering every slab, every scan
            for s in range(n):
                causal = self.outer_scans[s][0]
                prev = np.zeros((k, self.lines))
                for i in range(W):
                    h = i if causal else W - 1 - i
                    ins[s, h] = prev
                    if i == W - 1:
                        break
                    e = gathered[h, pl, s].copy()
                    for q in range(s + 1):
                        e += self.X[h, q, s] @ ins[q, h]
                    prev = e
                for j in range(k):
                    self.incoming[pl][s][j] = ins[s, self.rank, j].copy()
            for s in range(n):
                for t in range(M):
                    for q in range(s + 1):
                        self.tails[pl][s, t] += self.Y[q, s, t] @ ins[q, self.rank]

    def _filter_carry_planes(self):
        """early exchange: every completed tail / entering carry is a plane of the inner dimensions; filtered along them
        it is the carry of the FILTERED slab"""
        inner_shape = self.shape[1:]
        for pl in range(self.planes):
            t = self.tails[pl]
            self.tails[pl] = self._inner(t.reshape(t.shape[:3] + inner_shape)).reshape(t.shape)
            for s in range(self.n):
                for j in range(self.k):
                    self.incoming[pl][s][j] = self._inner(self.incoming[pl][s][j].reshape(inner_shape)).reshape(-1)

    def exchange_apply(self, s, gathered_ptr):
        if self.has_interior:
            self.interior()                      # (a driver that never asked for it)
            assert self.merged, "early exchange needs the merged exchange"
            self._merged_apply(gathered_ptr)
            return self._filter_carry_planes()
        if self.merged:
            return self._merged_apply(gathered_ptr)
        k, M = self.k, self.M
        gathered = _view(gathered_ptr, self.world * self.planes * k * self.lines).reshape(
            self.world, self.planes, k, self.lines).astype(np.float64)
        causal = self.outer_scans[s][0]
        AM = [np.linalg.matrix_power(self.A[s], e // self.T) for e in self.slab_extents]      # per slab
        for pl in range(self.planes):
            x = np.zeros((k, self.lines))
            count = self.rank if causal else self.world - 1 - self.rank
            for i in range(count):
                h = i if causal else self.world - 1 - i
                x = gathered[h, pl] + AM[h] @ x
            for j in range(k):
                self.incoming[pl][s][j] = x[j].copy()
            for i in range(M):
                t = i if causal else M - 1 - i
                x = self.A[s] @ x
                self.tails[pl][s, t] += x

    def finish(self):
        k, M, T = self.k, self.M, self.T
        import torch
        self.interior()
        for pl in range(self.planes):
            out = self.data[pl]
            if self.n:
                res = np.empty_like(out)
                for t in range(M):
                    v = out[:, t * T:(t + 1) * T].copy()
                    for s in range(self.n):
                        b, a = self._coef(s)
                        scan_tile(v, self.outer_scans[s][0], b, a, k, self.clamped and self._border(s, t),
                                  self._carry_into(pl, s, t))
                    res[:, t * T:(t + 1) * T] = v
                out = res
            self.outputs[pl].copy_(torch.from_numpy(out.T.reshape(self.shape).astype(np.float32)))

    def execute(self, inputs, outputs, stream=None):      # world == 1 path of ShardedFilter
        self.begin(inputs, outputs)
        import torch
        for i in range(self.num_exchanges):
            send = torch.empty(self.exchange_bytes(i), dtype=torch.uint8)
            self.exchange_local(i, send.data_ptr())
        self.finish()
        return outputs
