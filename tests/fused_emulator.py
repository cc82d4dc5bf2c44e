"""numpy emulation of the fused x/y path, driven by the product's plan tables.

Mirrors, stage by stage, what the GPU does for a 2-D image:
  fused_tails   x tails = H_x contraction of every row; y tails' combined rows = H_y contraction of the tile
  carry_x       blocked carry scan == serial recurrence with W_x / A_x
  xscan_rows    tile-local x scans of the combined rows (segment level, Kogge-Stone over 16 lanes) + the
                cross-dimension residual sum G_x * tau, tau = H_y contraction of the completed x-carry strips
  carry_y       serial recurrence with W_y / A_y
  fused_pass2   x phase / y phase with the completed carries
All constants come from rf_plan_table() of a host-only plan, so the tiling algebra of the product is
checked against the oracle on the CPU.  Test infrastructure only.
"""
from __future__ import annotations

import numpy as np

from recfilter_amd import capi

from tiled_emulator import scan_tile

TX, SEG = 256, 16


def _f32(c):
    return float(np.float32(c))


def plan_scans(plan):
    """The scans as the plan runs them (rf_plan_table("scans"): after the rewrite of high orders into sections), as
    (dim, causal, [b, a...]) plus, per scan, its border modification (mod_n, [g...]) -- mod_n < 0: native clamped prologue."""
    K = capi.RF_MAX_ORDER
    t = plan.table("scans").reshape(-1, 5 + 2 * K)
    scans = [(int(r[0]), bool(r[1]), [float(r[3])] + [float(v) for v in r[4:4 + int(r[2])]]) for r in t]
    mods = [(int(r[4 + K]), [float(v) for v in r[5 + K:5 + 2 * K]]) for r in t]
    return scans, mods


class FusedEmu:
    def __init__(self, plan, scans, clamped, mods=None):
        """mods: per scan of `scans` (mod_n, [g...]) for plans in mod form (plan_scans); None: native clamped prologues."""
        self.plan, self.clamped = plan, clamped
        mods = mods if mods is not None else [(-1, [])] * len(scans)
        self.xs = [(bool(c), [_f32(v) for v in co]) for d, c, co in scans if d == 0]
        self.ys = [(bool(c), [_f32(v) for v in co]) for d, c, co in scans if d == 1]
        self.xmod = [m for (d, _, _), m in zip(scans, mods) if d == 0]
        self.ymod = [m for (d, _, _), m in zip(scans, mods) if d == 1]
        self.K = max([len(co) - 1 for _, co in self.xs + self.ys])
        K, nx, ny = self.K, len(self.xs), len(self.ys)
        self.TY = plan.tiles[1]
        pad = lambda co: (co[0], list(co[1:]) + [0.0] * (K - len(co) + 1))
        self.xc = [pad(co) for _, co in self.xs]
        self.yc = [pad(co) for _, co in self.ys]
        if nx:
            self.segR = plan.table("seg_R_x").reshape(nx, SEG, K)
            self.segP = plan.table("seg_P_x").reshape(nx, 4, K, K)
            self.Wx = plan.table("W_x").reshape(4, nx, nx, K, K)
            self.Ax = plan.table("A_x").reshape(nx, K, K)
            self.G = plan.table("G_x").reshape(4, nx, K, TX)
            self.Hx = plan.table("H_x").reshape(4, nx, K, TX)
        if ny:
            self.Wy = plan.table("W_y").reshape(4, ny, ny, K, K)
            self.Ay = plan.table("A_y").reshape(ny, K, K)
            self.Hy = plan.table("H_y").reshape(4, ny, K, self.TY)

    # -- x phase of one scan on rows [R, 256], exactly the kernel's decomposition ------------------
    def xphase(self, rows, s, carry, clamp_first, last_lane=SEG - 1, entry_valid=SEG):
        """last_lane < 15 or entry_valid < 16: the tile is a row's partial last tile; its segments beyond last_lane and the
        samples beyond entry_valid of segment last_lane do not exist.  An anticausal scan enters at the last existing
        sample, the dead samples are cleared and the dead segments' exit states are dropped."""
        causal = self.xs[s][0]
        b, a = self.xc[s]
        K = self.K
        R = rows.shape[0]
        d = rows if causal else rows[:, ::-1]
        seg = d.reshape(R, SEG, SEG).copy()          # [row, lane, sample] in direction coordinates
        S = np.zeros((R, SEG, K))
        first = 0 if causal else SEG - 1 - last_lane  # direction-lane where the scan enters the row
        for l in range(SEG):
            v = seg[:, l, :].copy()
            c = None
            if l == first and carry is not None:
                c = [carry[j] for j in range(K)]
            if not causal and l == first and entry_valid < SEG:
                off = SEG - entry_valid                 # direction positions before the image
                v[:, :off] = 0.0
                w = np.ascontiguousarray(v[:, off:])
                scan_tile(w, True, b, a, K, clamp_first, c)
                v[:, off:] = w
            else:
                cf = clamp_first and l == first
                mod_n, g = self.xmod[s]
                if cf and mod_n >= 0:                     # zero-border form behind a border modification (plan.cpp)
                    x0 = v[:, 0].copy()
                    for r in range(mod_n):
                        v[:, r] += _f32(g[r]) * x0
                    cf = False
                scan_tile(v, True, b, a, K, cf, c)
            seg[:, l, :] = v
            if not causal and l < first:
                continue                              # dead lane: exit state stays zero
            for r in range(K):
                S[:, l, r] = v[:, SEG - 1 - r]
        for step, dist in enumerate((1, 2, 4, 8)):
            Sh = np.zeros_like(S)
            Sh[:, dist:, :] = S[:, :-dist, :]
            S = S + np.einsum("rj,nlj->nlr", self.segP[s, step], Sh)
        C = np.zeros_like(S)
        C[:, 1:, :] = S[:, :-1, :]
        seg = seg + np.einsum("pj,nlj->nlp", self.segR[s], C)
        out = seg.reshape(R, TX)
        return out if causal else out[:, ::-1]

    def run(self, img):
        img = np.asarray(img, dtype=np.float64)
        NY, NX_real = img.shape
        K, TY = self.K, self.TY
        MX = (NX_real + TX - 1) // TX                 # the last tile of a row may be partial: zero padded here,
        NX = MX * TX                                  # never read back
        TV = NX_real - (MX - 1) * TX                  # columns of the last tile
        last_lane_of = lambda tx: (TV - 1) // SEG if tx == MX - 1 else SEG - 1
        entry_of = lambda tx: TV - SEG * ((TV - 1) // SEG) if tx == MX - 1 else SEG
        if NX != NX_real:
            img = np.concatenate([img, np.zeros((NY, NX - NX_real))], axis=1)
        NY_real = NY
        MY = (NY_real + TY - 1) // TY                 # likewise the last tile row
        NY = MY * TY
        rows_of = lambda ty: NY_real - (MY - 1) * TY if ty == MY - 1 else TY
        if NY != NY_real:
            img = np.concatenate([img, np.zeros((NY - NY_real, NX))], axis=0)
        nx, ny = len(self.xs), len(self.ys)
        xt = np.zeros((nx, MX, K, NY))
        yt = np.zeros((ny, MY, K, NX))
        clamped = self.clamped

        def xfirst(s, tx):
            return tx == 0 if self.xs[s][0] else tx == MX - 1

        def yfirst(j, ty):
            return ty == 0 if self.ys[j][0] else ty == MY - 1

        def xcarry(s, tx):        # [K, NY]
            if xfirst(s, tx):
                return np.zeros((K, NY))
            return xt[s, tx - 1 if self.xs[s][0] else tx + 1]

        def ycarry(j, ty):
            if yfirst(j, ty):
                return np.zeros((K, NX))
            return yt[j, ty - 1 if self.ys[j][0] else ty + 1]

        def yscan(tile, j, carry, ty):
            # tile [TY, W]: scan along axis 0 == scan_tile over transposed rows; only the existing rows of a partial
            # last tile row are scanned (an anticausal scan enters at the last of them)
            rows = rows_of(ty)
            v = np.ascontiguousarray(tile[:rows].T)
            b, a = self.yc[j]
            cf = clamped and yfirst(j, ty)
            mod_n, g = self.ymod[j]
            if cf and mod_n >= 0:                         # zero-border form behind a border modification (plan.cpp)
                idx = (lambda r: r) if self.ys[j][0] else (lambda r: rows - 1 - r)
                x0 = v[:, idx(0)].copy()
                for r in range(mod_n):
                    v[:, idx(r)] += _f32(g[r]) * x0
                cf = False
            scan_tile(v, self.ys[j][0], b, a, K, cf,
                      None if carry is None else [carry[r] for r in range(K)])
            out = tile.copy()
            out[:rows] = v.T
            return out

        def ytail(tile, j):
            return np.stack([tile[TY - 1 - r] if self.ys[j][0] else tile[r] for r in range(K)])

        def vxof(tx):
            return (1 if tx == 0 else 0) | (2 if tx == MX - 1 else 0)

        def vyof(ty):
            return (1 if ty == 0 else 0) | (2 if ty == MY - 1 else 0)

        # ---- pass 1 (fused_tails): contractions, no recurrences ----
        for ty in range(MY):
            for tx in range(MX):
                t = img[ty * TY:(ty + 1) * TY, tx * TX:(tx + 1) * TX]
                for s in range(nx):
                    for r in range(K):
                        xt[s, tx, r, ty * TY:(ty + 1) * TY] = t @ self.Hx[vxof(tx), s, r]
                for j in range(ny):
                    for r in range(K):
                        yt[j, ty, r, tx * TX:(tx + 1) * TX] = self.Hy[vyof(ty), j, r] @ t      # combined rows

        # ---- x carry stage (generic_carry_scan_kernel on the x tails) ----
        for s in range(nx):
            prev = None
            for i in range(MX):
                tx = i if self.xs[s][0] else MX - 1 - i
                v = (1 if tx == 0 else 0) | (2 if tx == MX - 1 else 0)
                cur = xt[s, tx].copy()
                for q in range(s):
                    cur = cur + self.Wx[v, q, s] @ xcarry(q, tx)
                if i > 0:
                    cur = cur + self.Ax[s] @ prev
                xt[s, tx] = cur
                prev = cur

        # ---- xscan_rows: tile-local x scans of the combined rows + cross-dimension residual ----
        if nx and ny:
            xi = np.arange(TX)
            for ty in range(MY):
                for tx in range(MX):
                    rows = np.stack([yt[j, ty, r, tx * TX:(tx + 1) * TX] for j in range(ny) for r in range(K)])
                    for s in range(nx):
                        rows = self.xphase(rows, s, None, clamped and xfirst(s, tx), last_lane_of(tx), entry_of(tx))
                    for j in range(ny):
                        for r in range(K):
                            acc = rows[j * K + r].copy()
                            for q in range(nx):
                                if xfirst(q, tx):
                                    continue
                                strips = xcarry(q, tx)[:, ty * TY:(ty + 1) * TY]                  # [o, TY]
                                tau = strips @ self.Hy[vyof(ty), j, r]                              # [o]
                                acc = acc + tau @ self.G[vxof(tx), q]
                            yt[j, ty, r, tx * TX:(tx + 1) * TX] = acc

        # ---- y carry stage ----
        for j in range(ny):
            prev = None
            for i in range(MY):
                ty = i if self.ys[j][0] else MY - 1 - i
                cur = yt[j, ty].copy()
                for q in range(j):
                    cur = cur + self.Wy[vyof(ty), q, j] @ ycarry(q, ty)
                if i > 0:
                    cur = cur + self.Ay[j] @ prev
                yt[j, ty] = cur
                prev = cur

        # ---- pass 2 ----
        out = np.empty_like(img)
        for ty in range(MY):
            for tx in range(MX):
                t = img[ty * TY:(ty + 1) * TY, tx * TX:(tx + 1) * TX].copy()
                for s in range(nx):
                    c = None if xfirst(s, tx) else xcarry(s, tx)[:, ty * TY:(ty + 1) * TY]
                    t = self.xphase(t, s, c, clamped and xfirst(s, tx), last_lane_of(tx), entry_of(tx))
                for j in range(ny):
                    c = None if yfirst(j, ty) else ycarry(j, ty)[:, tx * TX:(tx + 1) * TX]
                    t = yscan(t, j, c, ty)
                out[ty * TY:(ty + 1) * TY, tx * TX:(tx + 1) * TX] = t
        return out[:NY_real, :NX_real]
