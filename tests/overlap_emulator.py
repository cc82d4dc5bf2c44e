"""numpy emulation of the fully overlapped N-D tiling (kernels_overlap.hip), driven by the PRODUCT's plan tables.

Replays pass 1 / { residual_d, carry_d } / pass 2 of RF_PATH_TILED_OVERLAPPED in float64 with W, A and G taken from
rf_plan_table(), so the cross-dimension residual algebra (lib/split.cpp:1215-1633, every pair of dimensions) is
checked against the oracle without a GPU.  Test infrastructure only."""
from __future__ import annotations

import itertools

import numpy as np

from tiled_emulator import scan_tile


def _scan_block(block, e, causal, b, a, k, clamp_first, carry=None):
    """One scan along dimension e (0 = x) of a 3-D block [Tz, Ty, Tx], in place; returns the k tails as arrays over the
    two other dimensions (ascending dimension order = (b, a) numpy order)."""
    ax = 2 - e
    moved = np.moveaxis(block, ax, -1)
    shp = moved.shape
    flat = np.ascontiguousarray(moved).reshape(-1, shp[-1])
    c = None if carry is None else [np.ascontiguousarray(cj).reshape(-1) for cj in carry]
    scan_tile(flat, causal, b, a, k, clamp_first, c)
    block[...] = np.moveaxis(flat.reshape(shp), -1, ax)
    T = shp[-1]
    return [flat[:, T - 1 - r if causal else r].reshape(shp[:-1]).copy() for r in range(k)]


def emulate_overlapped(image, scans, tiles, clamped, plan):
    img = np.array(image, dtype=np.float64)
    nd = img.ndim
    vol = img.reshape((1,) * (3 - nd) + img.shape)              # (Nz, Ny, Nx)
    N = [vol.shape[2], vol.shape[1], vol.shape[0]]
    dims = []
    for d in range(3):
        sc = [(bool(c), co) for (dd, c, co) in scans if dd == d]
        if not sc:
            dims.append(dict(n=0, T=1, M=N[d], k=0))
            continue
        k = max(len(co) - 1 for _, co in sc)
        coef = [(float(np.float32(co[0])), [float(np.float32(v)) for v in co[1:]] + [0.0] * (k - len(co) + 1)) for _, co in sc]
        T = tiles[d]
        n = len(sc)
        name = "xyz"[d]
        dims.append(dict(n=n, T=T, M=N[d] // T, k=k, causal=[c for c, _ in sc], coef=coef,
                         W=np.asarray(plan.table("W_" + name)).reshape(4, n, n, k, k),
                         A=np.asarray(plan.table("A_" + name)).reshape(n, k, k),
                         G=np.asarray(plan.table("G_" + name)).reshape(4, n, T, k)))
    # tails[d][s, t, r] is an array over the two other dimensions in numpy order (higher dimension first)
    def others(e):
        return [x for x in (2, 1, 0) if x != e]                  # dims, higher first -> numpy order of the remaining axes
    tails = {}
    for d in range(3):
        if dims[d]["n"]:
            o = others(d)
            tails[d] = np.zeros((dims[d]["n"], dims[d]["M"], dims[d]["k"], N[o[0]], N[o[1]]))

    def first(d, s, t):
        return t == 0 if dims[d]["causal"][s] else t == dims[d]["M"] - 1

    def variant(d, t):
        return (1 if t == 0 else 0) | (2 if t == dims[d]["M"] - 1 else 0)

    def rng(d, t):
        return slice(t * dims[d]["T"], (t + 1) * dims[d]["T"])

    def tile_view(arr, t):
        return arr[rng(2, t[2]), rng(1, t[1]), rng(0, t[0])]

    def tail_slice(d, t):
        o = others(d)
        return (rng(o[0], t[o[0]]), rng(o[1], t[o[1]]))

    all_tiles = list(itertools.product(*[range(dims[d]["M"]) for d in range(3)]))      # (tx, ty, tz)

    # ---- pass 1 ----
    for t in all_tiles:
        block = tile_view(vol, t).copy()
        for e in range(3):
            D = dims[e]
            for s in range(D["n"]):
                tl = _scan_block(block, e, D["causal"][s], D["coef"][s][0], D["coef"][s][1], D["k"], clamped and first(e, s, t[e]))
                for r in range(D["k"]):
                    tails[e][(s, t[e], r) + tail_slice(e, t)] = tl[r]

    def carry_stage(d):
        D = dims[d]
        n, k, M = D["n"], D["k"], D["M"]
        tl = tails[d]

        def carry_into(s, t):
            if first(d, s, t):
                return [np.zeros(tl.shape[3:]) for _ in range(k)]
            tp = t - 1 if D["causal"][s] else t + 1
            return [tl[s, tp, j] for j in range(k)]
        for s in range(n):
            prev = None
            for i in range(M):
                t = i if D["causal"][s] else M - 1 - i
                cur = [tl[s, t, r].copy() for r in range(k)]
                for q in range(s):
                    c = carry_into(q, t)
                    for r in range(k):
                        for o in range(k):
                            cur[r] = cur[r] + D["W"][variant(d, t), q, s, r, o] * c[o]
                if i > 0:
                    for r in range(k):
                        for j in range(k):
                            cur[r] = cur[r] + D["A"][s, r, j] * prev[j]
                for r in range(k):
                    tl[s, t, r] = cur[r]
                prev = cur

    def residual_stage(d):
        for t in all_tiles:
            r_field = np.zeros((dims[2]["T"], dims[1]["T"], dims[0]["T"]))
            nonzero = False
            for e in range(d):
                E = dims[e]
                if not E["n"]:
                    continue
                if nonzero:
                    for s in range(E["n"]):
                        _scan_block(r_field, e, E["causal"][s], E["coef"][s][0], E["coef"][s][1], E["k"],
                                    clamped and first(e, s, t[e]))
                ax = 2 - e
                for q in range(E["n"]):
                    if first(e, q, t[e]):
                        continue
                    tp = t[e] - 1 if E["causal"][q] else t[e] + 1
                    for o in range(E["k"]):
                        c = tails[e][(q, tp, o) + tail_slice(e, t)]               # over the two other dims
                        g = E["G"][variant(e, t[e]), q, :, o]                       # over positions along e
                        r_field += np.expand_dims(c, ax) * g.reshape([-1 if a == ax else 1 for a in range(3)])
                nonzero = True
            if not nonzero:
                continue
            D = dims[d]
            for s in range(D["n"]):
                tl = _scan_block(r_field, d, D["causal"][s], D["coef"][s][0], D["coef"][s][1], D["k"], clamped and first(d, s, t[d]))
                for r in range(D["k"]):
                    tails[d][(s, t[d], r) + tail_slice(d, t)] += tl[r]

    earlier = False
    for d in range(3):
        if not dims[d]["n"]:
            continue
        if earlier:
            residual_stage(d)
        carry_stage(d)
        earlier = True

    # ---- pass 2 ----
    out = np.empty_like(vol)
    for t in all_tiles:
        block = tile_view(vol, t).copy()
        for e in range(3):
            D = dims[e]
            for s in range(D["n"]):
                f = first(e, s, t[e])
                carry = None
                if not f:
                    tp = t[e] - 1 if D["causal"][s] else t[e] + 1
                    carry = [tails[e][(s, tp, j) + tail_slice(e, t)] for j in range(D["k"])]
                _scan_block(block, e, D["causal"][s], D["coef"][s][0], D["coef"][s][1], D["k"], clamped and f, carry)
        tile_view(out, t)[...] = block
    return out.reshape(img.shape)
