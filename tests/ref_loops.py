"""Independent restatement of the loop references inside the reference's tests/apps.

The reference's tests hold no stored vectors: each test file computes its expected
result with hand-written raster loops and prints the difference
(/root/reference/tests/test_*.cpp, apps/summed_table/summed_table.cpp:66-82,
apps/bspline/bicubic_filter.cpp:124-156).  Those loops are the known-answer
definition the oracle is pinned against, so they are restated here in plain
Python/numpy scalar arithmetic (float32 like the tests), written in the *tests'*
form -- `ref += tap1 + tap2 + ...` walking the image in raster order -- not in
the oracle's form.  Small images only (pure-Python loops).
"""
from __future__ import annotations

import numpy as np

f32 = np.float32


def _lines(img: np.ndarray, dim: int):
    """Yield 1-D views along scan dim `dim` (0 = x = last numpy axis)."""
    axis = img.ndim - 1 - dim
    moved = np.moveaxis(img, axis, -1)
    for idx in np.ndindex(*moved.shape[:-1]):
        yield moved[idx]


def zero_border_loops(image: np.ndarray, scans) -> np.ndarray:
    """The `ref(x,y) += (x>0 ? W0*ref(x-1,y) : 0) + (x>1 ? W1*ref(x-2,y) : 0) ...` loops of
    tests/test_generic_xy.cpp:62-110 and friends (feedforward is 1 in every such test)."""
    ref = np.array(image, copy=True)
    is_int = np.issubdtype(ref.dtype, np.integer)
    for dim, causal, coeff in scans:
        assert float(coeff[0]) == 1.0, "the tests' loops hard-code feedforward 1"
        fb = coeff[1:]
        for line in _lines(ref, dim):
            n = line.shape[0]
            for r in range(n):
                i = r if causal else n - 1 - r
                if is_int:
                    # tests/test_type_invariance.cpp:48-60: int16_t(W)*ref(...)
                    acc = 0
                    for j, w in enumerate(fb):
                        if r > j:
                            acc += int(w) * int(line[i - (j + 1) if causal else i + (j + 1)])
                    line[i] = np.array(int(line[i]) + acc).astype(ref.dtype)
                else:
                    acc = f32(0.0)
                    for j, w in enumerate(fb):
                        if r > j:
                            acc = f32(acc + f32(f32(w) * line[i - (j + 1) if causal else i + (j + 1)]))
                    line[i] = f32(line[i] + acc)
    return ref


def summed_table_loops(image: np.ndarray) -> np.ndarray:
    """apps/summed_table/summed_table.cpp:66-82."""
    ref = np.array(image, copy=True)
    h, w = ref.shape
    for y in range(h):
        for x in range(1, w):
            ref[y, x] = ref[y, x] + ref[y, x - 1]
    for y in range(1, h):
        for x in range(w):
            ref[y, x] = ref[y, x] + ref[y - 1, x]
    return ref


def clamped_xy_loops(image: np.ndarray, coeff) -> np.ndarray:
    """apps/bspline/bicubic_filter.cpp:124-156: b0*ref + a1*ref(max(x-1,0)) + a2*ref(max(x-2,0)),
    applied +x, +y, -x, -y in place.  (The app reads filter_coeff[2] of a 2-vector, an
    out-of-bounds read; the restatement uses a2 = 0 for order-1 filters.)"""
    ref = np.array(image, dtype=np.float32, copy=True)
    h, w = ref.shape
    b0 = f32(coeff[0])
    a1 = f32(coeff[1]) if len(coeff) > 1 else f32(0)
    a2 = f32(coeff[2]) if len(coeff) > 2 else f32(0)
    for y in range(h):
        for x in range(w):
            ref[y, x] = b0 * ref[y, x] + a1 * ref[y, max(x - 1, 0)] + a2 * ref[y, max(x - 2, 0)]
    for y in range(h):
        for x in range(w):
            ref[y, x] = b0 * ref[y, x] + a1 * ref[max(y - 1, 0), x] + a2 * ref[max(y - 2, 0), x]
    for y in range(h):
        for x in range(w):
            ref[y, w - 1 - x] = (b0 * ref[y, w - 1 - x] + a1 * ref[y, w - 1 - max(x - 1, 0)]
                                 + a2 * ref[y, w - 1 - max(x - 2, 0)])
    for y in range(h):
        for x in range(w):
            ref[h - 1 - y, x] = (b0 * ref[h - 1 - y, x] + a1 * ref[h - 1 - max(y - 1, 0), x]
                                 + a2 * ref[h - 1 - max(y - 2, 0), x])
    return ref


def box_difference(table, radius, order):
    """apps/box/box_filter.h:36-39,128-139 restated: along every dimension d (order given in x, y, z order, numpy axes
    reversed), order[d] times, out(i) = (s(min(i+B, N-1)) - s(max(i-B-1, 0))) / (2B+1)."""
    out = np.asarray(table, dtype=np.float64)
    nd = out.ndim
    for d, o in enumerate(order):
        axis = nd - 1 - d
        n = out.shape[axis]
        i = np.arange(n)
        hi, lo = np.minimum(i + radius, n - 1), np.maximum(i - radius - 1, 0)
        for _ in range(o):
            out = (np.take(out, hi, axis=axis) - np.take(out, lo, axis=axis)) / (2 * radius + 1)
    return out


def tap_filter(planes, taps):
    """rf_tap_filter restated: out(p) = sum_t w_t * planes[plane_t](clamp(p + off_t)), offsets in (x, y, z) order."""
    planes = [np.asarray(p, dtype=np.float64) for p in planes]
    shape = planes[0].shape
    nd = len(shape)
    out = np.zeros(shape)
    grids = np.meshgrid(*[np.arange(n) for n in shape], indexing="ij")
    for plane, off, w in taps:
        off = list(off) + [0] * (nd - len(off))
        idx = tuple(np.clip(grids[ax] + off[nd - 1 - ax], 0, shape[ax] - 1) for ax in range(nd))
        out += w * planes[plane][idx]
    return out


def dog_taps(B1, B2):
    """The difference operators of apps/DoG/diff_gauss.cpp as tap lists (plane, (dx, dy), weight):
    diff_op_xy (:187-193) for two radii on one table, diff_op_x / diff_op_y (:176-184) per plane, and the final
    difference of the two planes (:103)."""
    def xy(B):
        s = 1.0 / (2 * B + 1) ** 2
        return [(0, (B, B), s), (0, (B, -B - 1), -s), (0, (-B - 1, -B - 1), s), (0, (-B - 1, B), -s)]

    def x2(B, plane=0, sign=1.0):
        s = sign / (2 * B + 1)
        return [(plane, (B, 0), s), (plane, (-1, 0), -2.0 * s), (plane, (-2 * B - 2, 0), s)]

    def y2(B, plane=0, sign=1.0):
        s = sign / (2 * B + 1)
        return [(plane, (0, B), s), (plane, (0, -1), -2.0 * s), (plane, (0, -2 * B - 2), s)]
    return dict(box1=[xy(B1), xy(B2)], box2x=[x2(B1), x2(B2)], dog=y2(B1, 0, 1.0) + y2(B2, 1, -1.0))


def dog_pipeline(image, B1, B2, scan):
    """apps/DoG/diff_gauss.cpp:66-103 with `scan(img, scans)` as the recursive-filter step (zero border)."""
    taps = dog_taps(B1, B2)
    sat = scan(image, [(0, True, [1.0, 1.0]), (1, True, [1.0, 1.0])])
    box1 = [tap_filter([sat], t) for t in taps["box1"]]
    sat2x = [scan(b, [(0, True, [1.0, 2.0, -1.0])]) for b in box1]
    box2x = [tap_filter([s2], t) for s2, t in zip(sat2x, taps["box2x"])]
    sat2y = [scan(b, [(1, True, [1.0, 2.0, -1.0])]) for b in box2x]
    return tap_filter(sat2y, taps["dog"])
