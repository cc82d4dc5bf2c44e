"""numpy emulation of the tiled algorithm, driven by the PRODUCT's plan tables.

The carry algebra (which tables exist, how they are indexed, which border variant applies)
is decided on the host inside librecfilter_amd.so.  This emulator replays the same stages the
kernels run -- pass 1 (intra-tile scans + tail extraction), carry (same-dimension chaining +
recurrence), pass 2 (final correction) -- in float64 numpy, taking W and A from
rf_plan_table(), so a host-only plan can be checked against the oracle without a GPU.
It mirrors kernels_generic.hip line by line and is test infrastructure only.
"""
from __future__ import annotations

import numpy as np


def scan_tile(v, causal, b, a, k, clamp_first, carry=None):
    """In-place scan of tiles v[lines, T] in memory order (mirrors rf::scan_tile / scan_step)."""
    lines, T = v.shape
    hist = [np.zeros(lines) if carry is None else carry[j].copy() for j in range(k)]
    y0 = np.zeros(lines)
    for p in range(T):
        m = p if causal else T - 1 - p
        x = v[:, m].copy()
        acc = b * x
        for j in range(k):
            g = hist[j]
            if clamp_first and p <= j:
                g = x if p == 0 else y0
            acc = acc + a[j] * g
        hist = [acc] + hist[:-1]
        if p == 0:
            y0 = acc
        v[:, m] = acc


def emulate_dimension(data, scans, T, clamped, W, A):
    """data: [lines, N] float64; scans: list of (causal, [b, a1..]) of ONE dimension, in order.
    W: [4, n, n, k, k], A: [n, k, k] from the plan.  Returns the filtered [lines, N]."""
    lines, N = data.shape
    n = len(scans)
    k = max(len(c) - 1 for _, c in scans)
    M = N // T
    # coefficients cross the C ABI as floats (rf_scan_desc), exactly like the reference's vector<float>
    coef = [(float(np.float32(c[0])), [float(np.float32(v)) for v in c[1:]] + [0.0] * (k - len(c) + 1)) for _, c in scans]
    caus = [bool(c) for c, _ in scans]
    W = np.asarray(W).reshape(4, n, n, k, k)
    A = np.asarray(A).reshape(n, k, k)

    def first(s, t):
        return t == 0 if caus[s] else t == M - 1

    def variant(t):
        return (1 if t == 0 else 0) | (2 if t == M - 1 else 0)

    tails = np.zeros((n, M, k, lines))
    # pass 1
    for t in range(M):
        v = data[:, t * T:(t + 1) * T].copy()
        for s in range(n):
            scan_tile(v, caus[s], coef[s][0], coef[s][1], k, clamped and first(s, t))
            for r in range(k):
                p = T - 1 - r
                tails[s, t, r] = v[:, p if caus[s] else T - 1 - p]

    def carry_into(s, t):
        if first(s, t):
            return [np.zeros(lines) for _ in range(k)]
        tp = t - 1 if caus[s] else t + 1
        return [tails[s, tp, j] for j in range(k)]

    # carry
    for s in range(n):
        prev = None
        for i in range(M):
            t = i if caus[s] else M - 1 - i
            cur = [tails[s, t, r].copy() for r in range(k)]
            for q in range(s):
                c = carry_into(q, t)
                for r in range(k):
                    for o in range(k):
                        cur[r] = cur[r] + W[variant(t), q, s, r, o] * c[o]
            if i > 0:
                for r in range(k):
                    for j in range(k):
                        cur[r] = cur[r] + A[s, r, j] * prev[j]
            for r in range(k):
                tails[s, t, r] = cur[r]
            prev = cur
    # pass 2
    out = np.empty_like(data)
    for t in range(M):
        v = data[:, t * T:(t + 1) * T].copy()
        for s in range(n):
            scan_tile(v, caus[s], coef[s][0], coef[s][1], k, clamped and first(s, t), carry_into(s, t))
        out[:, t * T:(t + 1) * T] = v
    return out


def emulate_filter(image, scans, tiles, clamped, plan):
    """Dimension-cascaded tiled filter on image[..., z, y, x] with the plan's tables."""
    out = np.array(image, dtype=np.float64)
    nd = out.ndim
    for d in range(nd):
        dim_scans = [(c, co) for (dd, c, co) in scans if dd == d]
        if not dim_scans:
            continue
        axis = nd - 1 - d
        moved = np.moveaxis(out, axis, -1)
        flat = np.ascontiguousarray(moved).reshape(-1, moved.shape[-1])
        name = "xyz"[d]
        res = emulate_dimension(flat, dim_scans, tiles[d], clamped, plan.table("W_" + name), plan.table("A_" + name))
        out = np.moveaxis(res.reshape(moved.shape), -1, axis)
    return out
