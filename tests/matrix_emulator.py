"""numpy replay of the matrix path (recfilter_amd/csrc/kernels_matrix.hip, plan_matrix.cpp) from the PRODUCT'S tables.

Test infrastructure: it restates in float32 numpy what the three kernels of a stage do -- pass 1 (tails = H . tile, plus
the rank-one clamped-border term), the carry chain with its levels (chunks of 16, transfer matrices A^(16^l), propagation
with the tabulated powers) and pass 2 (y_b = G x_b + R y_(b-1) over 32-sample sub-blocks) -- reading G, R, dG, H, dH, A
through rf_plan_table of a host-only plan.  So the tiling algebra of orders up to 32 is checked on the CPU against the
untiled oracle; the GPU tests then check the kernels (the MFMA operand layouts, the LDS staging, the index arithmetic)."""
from __future__ import annotations

import numpy as np

CHUNK, TOPMAX = 16, 24       # kMxChunk, kMxTopMax (kernels_matrix.h; the default when the plan's levels are not given)


def _f32(a):
    return np.asarray(a, dtype=np.float32)


def chain_levels(local, A, causal, nlevels=None):
    """local: [lines, M, k] tile-local tails (memory order of tiles) -> completed tails, by the blocked scheme of the
    kernels: chunk-local chains, the chunk exits as the next level's sequence with transfer matrix B^16, then the
    propagation down with B^(j+1)."""
    A64 = np.asarray(A, dtype=np.float64)
    seq = local if causal else local[:, ::-1]          # scan order
    seqs, mats = [np.array(seq, dtype=np.float32)], [A64]
    # (the plan says how many levels it runs -- mx_levels_<scan>: a long line is chained in one go where the lines alone fill the chip)
    while (len(seqs) < nlevels) if nlevels is not None else (seqs[-1].shape[1] > TOPMAX):
        cur, B = seqs[-1], _f32(mats[-1])
        M = cur.shape[1]
        nch = (M + CHUNK - 1) // CHUNK
        exits = np.zeros((cur.shape[0], nch, cur.shape[2]), dtype=np.float32)
        for c in range(nch):
            x = np.zeros((cur.shape[0], cur.shape[2]), dtype=np.float32)
            for g in range(c * CHUNK, min(M, (c + 1) * CHUNK)):
                x = cur[:, g] + x @ B.T
                cur[:, g] = x
            exits[:, c] = x
        seqs.append(exits)
        mats.append(np.linalg.matrix_power(mats[-1], CHUNK))
    top, B = seqs[-1], _f32(mats[-1])
    x = np.zeros((top.shape[0], top.shape[2]), dtype=np.float32)
    for g in range(top.shape[1]):
        x = top[:, g] + x @ B.T
        top[:, g] = x
    for l in range(len(seqs) - 2, -1, -1):
        cur, up = seqs[l], seqs[l + 1]
        M = cur.shape[1]
        for c in range(1, up.shape[1]):
            for j, g in enumerate(range(c * CHUNK, min(M, (c + 1) * CHUNK))):
                P = _f32(np.linalg.matrix_power(mats[l], j + 1))
                cur[:, g] = cur[:, g] + up[:, c - 1] @ P.T
    done = seqs[0]
    return done if causal else done[:, ::-1]


def _pair_stage(plan, out, rows, i, p, clamped):
    """A PAIR stage (MxPassArgs::pair): the causal scan i and the anticausal scan p of one dimension in one final pass -- pass 1 forms
    both scans' tile-local tails (H x and H21 x), the causal carries are chained, W21 . (the carry entering a tile) and the
    anticausal scan's clamped-border term go into the anticausal tails, those are chained, and the final pass walks the tile
    forward (w) and backward (y)."""
    from recfilter_amd import capi
    dim, k1, k2 = int(rows[i][0]), int(rows[i][2]), int(rows[p][2])
    assert bool(rows[i][1]) and not bool(rows[p][1]) and int(rows[p][0]) == dim
    T = int(plan.tiles[dim])
    G1, R1 = _f32(plan.table(f"mx_G_{i}").reshape(32, 32)), _f32(plan.table(f"mx_R_{i}").reshape(32, 32))
    G2, R2 = _f32(plan.table(f"mx_G_{p}").reshape(32, 32)), _f32(plan.table(f"mx_R_{p}").reshape(32, 32))
    dG1, dH1 = _f32(plan.table(f"mx_dG_{i}")), _f32(plan.table(f"mx_dH_{i}"))
    dG2, dH2 = _f32(plan.table(f"mx_dG_{p}")), _f32(plan.table(f"mx_dH_{p}"))
    H1 = _f32(plan.table(f"mx_H_{i}").reshape(32, T))
    H21, v21 = _f32(plan.table(f"mx_H21_{i}").reshape(32, T)), _f32(plan.table(f"mx_v21_{i}"))
    W21 = _f32(plan.table(f"mx_W21_{i}").reshape(k2, k1))
    A1, A2 = plan.table(f"mx_A_{i}").reshape(k1, k1), plan.table(f"mx_A_{p}").reshape(k2, k2)
    nd = out.ndim
    axis = nd - 1 - dim
    v = np.moveaxis(out, axis, -1)
    shp = v.shape
    N = shp[-1]
    Tg, M, off = (int(g) for g in plan.table(f"mx_geom_{i}"))
    assert Tg == T and M * T == N and off == 0
    x = np.ascontiguousarray(v).reshape(-1, M, T).astype(np.float32)
    NB = T // 32
    l1 = np.einsum("lmt,rt->lmr", x, H1[:k1]).astype(np.float32)
    l2 = np.einsum("lmt,rt->lmr", x, H21[:k2]).astype(np.float32)
    if clamped:
        l1[:, 0] += x[:, 0, 0][:, None] * dH1[:k1][None, :]
        l2[:, 0] += x[:, 0, 0][:, None] * v21[:k2][None, :]
    c1 = chain_levels(l1, A1, True, len(plan.table(f"mx_levels_{i}")) // 2)
    for t in range(1, M):
        l2[:, t] = l2[:, t] + c1[:, t - 1] @ W21.T
    if clamped:
        l2[:, M - 1] = l2[:, M - 1] + c1[:, M - 1, 0][:, None] * dH2[:k2][None, :]
    c2 = chain_levels(l2.astype(np.float32), A2, False, len(plan.table(f"mx_levels_{p}")) // 2)
    y = np.empty_like(x)
    for t in range(M):
        prev = np.zeros((x.shape[0], 32), dtype=np.float32)
        if t > 0:
            for r in range(k1):
                prev[:, 31 - r] = c1[:, t - 1, r]
        w = []
        for b in range(NB):
            c = x[:, t, 32 * b:32 * b + 32] @ G1.T + prev @ R1.T
            if b == 0 and clamped and t == 0:
                c = c + x[:, 0, 0][:, None] * dG1[None, :]
            c = c.astype(np.float32)
            w.append(c)
            prev = c
        back = np.zeros((x.shape[0], 32), dtype=np.float32)
        if t < M - 1:
            for r in range(k2):
                back[:, r] = c2[:, t + 1, r]
        for b in range(NB - 1, -1, -1):
            c = w[b] @ G2.T + back @ R2.T
            if b == NB - 1 and clamped and t == M - 1:
                c = c + w[NB - 1][:, 31][:, None] * dG2[None, :]
            c = c.astype(np.float32)
            y[:, t, 32 * b:32 * b + 32] = c
            back = c
    return np.ascontiguousarray(np.moveaxis(y.reshape(shp[:-1] + (N,)), -1, axis))


def run(plan, img, clamped):
    """The filter of a host-only RF_PATH_TILED_MATRIX plan applied to img (numpy, (z,) y, x order), stage by stage."""
    from recfilter_amd import capi
    Kmax = capi.RF_MAX_ORDER
    rows = plan.table("scans").reshape(-1, 5 + 2 * Kmax)
    tiles = plan.tiles
    out = np.array(img, dtype=np.float32)
    nd = out.ndim
    skip = set()
    for i, row in enumerate(rows):
        if i in skip:
            continue
        try:
            partner = int(plan.table(f"mx_pair_{i}")[0])
        except Exception:
            partner = None
        if partner is not None:
            out = _pair_stage(plan, out, rows, i, partner, clamped)
            skip.add(partner)
            continue
        dim, causal, k = int(row[0]), bool(row[1]), int(row[2])
        T = int(tiles[dim])
        G, R = _f32(plan.table(f"mx_G_{i}").reshape(32, 32)), _f32(plan.table(f"mx_R_{i}").reshape(32, 32))
        dG, dH = _f32(plan.table(f"mx_dG_{i}")), _f32(plan.table(f"mx_dH_{i}"))
        H = _f32(plan.table(f"mx_H_{i}").reshape(32, T))
        A = plan.table(f"mx_A_{i}").reshape(k, k)
        axis = nd - 1 - dim
        v = np.moveaxis(out, axis, -1)
        shp = v.shape
        N = shp[-1]
        Tg, M, off = (int(g) for g in plan.table(f"mx_geom_{i}"))
        assert Tg == T and (M - 1) * T < N + off <= M * T and (off == 0 or not causal)
        # tiles that do not divide the extent: zeros where the scan leaves the image (behind the end for a causal scan, in front
        # of the start for an anticausal one), never stored
        padded = np.zeros(v.shape[:-1] + (M * T,), dtype=np.float32)
        padded[..., off:off + N] = v
        x = padded.reshape(-1, M, T)
        first_tile, m0 = (0, 0) if causal else (M - 1, T - 1)
        # pass 1
        local = np.einsum("lmt,rt->lmr", x, H[:k]).astype(np.float32)
        if clamped:
            local[:, first_tile] += x[:, first_tile, m0][:, None] * dH[:k][None, :]
        tails = chain_levels(local, A, causal, len(plan.table(f"mx_levels_{i}")) // 2)
        # pass 2
        y = np.empty_like(x)
        NB = T // 32
        for t in range(M):
            enters = t == first_tile
            prev = np.zeros((x.shape[0], 32), dtype=np.float32)
            if not enters:
                carry = tails[:, t - 1 if causal else t + 1]            # [lines, k]: carry[r] = tail r
                for r in range(k):
                    prev[:, 31 - r if causal else r] = carry[:, r]
            for bi in range(NB):
                sb = bi if causal else NB - 1 - bi
                xb = x[:, t, 32 * sb:32 * sb + 32]
                c = xb @ G.T + prev @ R.T
                if bi == 0 and clamped and enters:
                    c = c + x[:, t, m0][:, None] * dG[None, :]
                c = c.astype(np.float32)
                y[:, t, 32 * sb:32 * sb + 32] = c
                prev = c
        out = np.moveaxis(y.reshape(shp[:-1] + (M * T,))[..., off:off + N], -1, axis)
    return np.ascontiguousarray(out)
