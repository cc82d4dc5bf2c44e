"""Plan: thin Python handle over rf_plan_* (include/recfilter_amd.h).

Images are torch CUDA(HIP) tensors used purely as device memory: a tensor of shape
(..., z, y, x), C-contiguous, is the reference's dense x-fastest buffer
(/root/reference lib/recfilter.cpp:969-981).  All arithmetic happens in the HIP kernels.
"""
from __future__ import annotations

import ctypes
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import capi

Scan = Tuple[int, bool, Sequence[float]]   # (dim, causal, [feedfwd, fb1..fbk]); dim 0 = x

# rf_filter_desc.flags of plans created without an explicit `flags`.  0 as shipped.  tests/conftest.py sets
# capi.RF_PLAN_TILED_ONLY: most parity tests use small shapes on purpose and mean the TILED kernels when they ask for the
# automatic path (which sends images up to 1024^2 to the line kernels); tests/test_shipped_defaults.py resets it to 0.
DEFAULT_FLAGS = 0

_NP_DTYPES = {np.dtype(np.float32): capi.RF_F32, np.dtype(np.float64): capi.RF_F64,
              np.dtype(np.int32): capi.RF_I32, np.dtype(np.int16): capi.RF_I16}


def _dtype_code(dtype) -> int:
    try:
        import torch
        if isinstance(dtype, torch.dtype):
            dtype = {torch.float32: np.float32, torch.float64: np.float64,
                     torch.int32: np.int32, torch.int16: np.int16}[dtype]
    except ImportError:  # pragma: no cover
        pass
    except KeyError:
        raise TypeError(f"unsupported pixel type {dtype}")
    dt = np.dtype(dtype)
    if dt not in _NP_DTYPES:
        raise TypeError(f"unsupported pixel type {dt}")
    return _NP_DTYPES[dt]


class Plan:
    """One tiled (or untiled) recursive filter on one device: rf_plan_create .. rf_plan_destroy."""

    def __init__(self, shape: Sequence[int], scans: Sequence[Scan], dtype=np.float32, clamped: bool = False,
                 planes: int = 1, tile: Optional[Sequence[int]] = None, path: int = capi.RF_PATH_AUTO,
                 device: int = -1, shard_rank: int = 0, shard_world: int = 1, shard_extents: Optional[Sequence[int]] = None,
                 prologue: Optional[Tuple[float, float]] = None,
                 epilogue: Optional[Tuple[float, float, float]] = None, input_dtype=None, flags: Optional[int] = None):
        """prologue = (scale, bias): x' = scale*in + bias before the first scan;
        epilogue = (w_filtered, w_input, bias): out = w_filtered*F(x') + w_input*x' + bias
        (rf_pointwise_desc; fused into pass 1 / pass 2 on the fused path).  input_dtype=np.uint8 (with dtype float32):
        the input planes are unsigned bytes converted on load (rf_pointwise_desc.in_dtype = RF_IN_U8).
        flags: rf_filter_desc.flags (capi.RF_PLAN_*); None = recfilter_amd.plan.DEFAULT_FLAGS (0 as shipped)."""
        L = capi.lib()
        shape = tuple(int(s) for s in shape)
        if not 1 <= len(shape) <= capi.RF_MAX_DIMS:
            raise ValueError(f"1..{capi.RF_MAX_DIMS} dimensions supported, got shape {shape}")
        scans = list(scans)
        self._scan_arr = (capi.ScanDesc * max(len(scans), 1))()
        for i, (dim, causal, coeff) in enumerate(scans):
            coeff = [float(c) for c in coeff]
            if len(coeff) < 2:
                # lib/recfilter.cpp:274-278
                raise ValueError("cannot add a scan without feed forward and feedback coefficients")
            if len(coeff) - 1 > capi.RF_MAX_ORDER:
                raise ValueError(f"filter order {len(coeff) - 1} exceeds RF_MAX_ORDER={capi.RF_MAX_ORDER}")
            s = self._scan_arr[i]
            s.dim, s.causal, s.order, s.feedfwd = int(dim), int(bool(causal)), len(coeff) - 1, coeff[0]
            for j, c in enumerate(coeff[1:]):
                s.feedback[j] = c
        d = capi.FilterDesc()
        d.abi = capi.RF_ABI
        d.ndim = len(shape)
        for i, e in enumerate(reversed(shape)):      # numpy (z,y,x) -> extent[0] = x
            d.extent[i] = e
        d.dtype = _dtype_code(dtype)
        d.n_planes = int(planes)
        d.border = capi.RF_BORDER_CLAMP if clamped else capi.RF_BORDER_ZERO
        d.n_scans = len(scans)
        d.scans = ctypes.cast(self._scan_arr, ctypes.POINTER(capi.ScanDesc))
        if tile is not None:
            tile = list(tile)
            if len(tile) != len(shape):
                raise ValueError("tile needs one entry per dimension, in (x, y, z) order")
            for i, t in enumerate(tile):
                d.tile[i] = int(t)
        d.path = int(path)
        d.device = int(device)
        d.shard_rank, d.shard_world = int(shard_rank), int(shard_world)
        if shard_extents is not None:
            # rf_filter_desc.shard_extents: the extent of every rank's slab along the outermost dimension
            if len(shard_extents) != int(shard_world):
                raise ValueError("shard_extents needs one extent per rank")
            self._shard_extents = (ctypes.c_int64 * int(shard_world))(*[int(e) for e in shard_extents])
            d.shard_extents = ctypes.cast(self._shard_extents, ctypes.POINTER(ctypes.c_int64))
        if prologue is not None:
            d.pointwise.flags |= capi.RF_POINTWISE_PRE
            d.pointwise.pre_scale, d.pointwise.pre_bias = float(prologue[0]), float(prologue[1])
        if epilogue is not None:
            d.pointwise.flags |= capi.RF_POINTWISE_POST
            d.pointwise.post_filtered, d.pointwise.post_input, d.pointwise.post_bias = (float(v) for v in epilogue)
        self.input_np_dtype = None
        if input_dtype is not None and np.dtype(input_dtype if not hasattr(input_dtype, "is_floating_point") else
                                                 str(input_dtype).replace("torch.", "")) == np.dtype(np.uint8):
            d.pointwise.in_dtype = capi.RF_IN_U8
            self.input_np_dtype = np.dtype(np.uint8)
        elif input_dtype is not None and _dtype_code(input_dtype) != d.dtype:
            raise TypeError(f"unsupported input type {input_dtype} for pixel type {dtype}")
        d.flags = int(DEFAULT_FLAGS if flags is None else flags)
        self._desc = d
        self.shape = shape
        self.planes = int(planes)
        self.np_dtype = np.dtype({capi.RF_F32: np.float32, capi.RF_F64: np.float64,
                                  capi.RF_I32: np.int32, capi.RF_I16: np.int16}[d.dtype])
        self.shard_world = int(shard_world)
        self._h = ctypes.c_void_p()
        capi.check(L.rf_plan_create(ctypes.byref(d), ctypes.byref(self._h)))

    # -- lifetime ---------------------------------------------------------------------------
    def close(self) -> None:
        if getattr(self, "_h", None) and self._h.value:
            capi.lib().rf_plan_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- queries ----------------------------------------------------------------------------
    @property
    def path(self) -> int:
        return capi.lib().rf_plan_path(self._h)

    @property
    def path_name(self) -> str:
        return capi.PATH_NAMES.get(self.path, "?")

    @property
    def tiles(self) -> Tuple[int, ...]:
        t = (ctypes.c_int32 * capi.RF_MAX_DIMS)()
        capi.check(capi.lib().rf_plan_tiles(self._h, t))
        return tuple(t[: len(self.shape)])

    @property
    def workspace_bytes(self) -> int:
        return int(capi.lib().rf_plan_workspace_bytes(self._h))

    @property
    def num_instances(self) -> int:
        """execution instances the plan holds: 1 + the replicas concurrent executes on other streams made it build"""
        return int(capi.lib().rf_plan_num_instances(self._h))

    @property
    def num_kernels(self) -> int:
        return int(capi.lib().rf_plan_num_kernels(self._h))

    @property
    def num_exchanges(self) -> int:
        return int(capi.lib().rf_plan_num_exchanges(self._h))

    def table(self, name: str) -> np.ndarray:
        n = ctypes.c_size_t()
        capi.check(capi.lib().rf_plan_table(self._h, name.encode(), None, 0, ctypes.byref(n)))
        out = np.empty(n.value, dtype=np.float64)
        capi.check(capi.lib().rf_plan_table(self._h, name.encode(),
                                            out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), n.value, None))
        return out

    # -- execution --------------------------------------------------------------------------
    def _pointers(self, tensors, inputs: bool = False) -> ctypes.Array:
        if len(tensors) != self.planes:
            raise ValueError(f"expected {self.planes} planes, got {len(tensors)}")
        arr = (ctypes.c_void_p * self.planes)()
        for i, t in enumerate(tensors):
            if tuple(t.shape) != self.shape:
                raise ValueError(f"plane {i}: shape {tuple(t.shape)} != plan shape {self.shape}")
            if not t.is_cuda or not t.is_contiguous():
                raise ValueError("planes must be contiguous device tensors")
            if inputs and self.input_np_dtype is not None:
                import torch
                if t.dtype != torch.uint8:
                    raise TypeError(f"plane {i}: the plan expects unsigned-byte input planes, got {t.dtype}")
            elif _dtype_code(t.dtype) != self._desc.dtype:
                raise TypeError(f"plane {i}: dtype {t.dtype} does not match the plan")
            arr[i] = t.data_ptr()
        return arr

    def _new_outputs(self, inputs):
        import torch
        tdt = {np.dtype(np.float32): torch.float32, np.dtype(np.float64): torch.float64, np.dtype(np.int32): torch.int32,
               np.dtype(np.int16): torch.int16}[self.np_dtype]
        return [torch.empty(t.shape, dtype=tdt, device=t.device) for t in inputs]

    @staticmethod
    def _stream(stream) -> ctypes.c_void_p:
        import torch
        s = stream if stream is not None else torch.cuda.current_stream()
        return ctypes.c_void_p(s.cuda_stream)

    def execute(self, inputs, outputs=None, stream=None):
        """rf_plan_execute: asynchronous on `stream` (default: torch's current stream).  Executes on distinct streams may
        overlap: the plan runs them on replicas of itself with their own workspaces (include/recfilter_amd.h)."""
        import torch
        if outputs is None:
            outputs = self._new_outputs(inputs)
        pin, pout = self._pointers(inputs, True), self._pointers(outputs)
        capi.check(capi.lib().rf_plan_execute(self._h, pin, pout, self._stream(stream)))
        return outputs

    def execute_timed(self, inputs, outputs=None, stream=None):
        """rf_plan_execute_timed: returns (outputs, [(kernel name, ms), ...]) measured with HIP events."""
        import torch
        if outputs is None:
            outputs = self._new_outputs(inputs)
        pin, pout = self._pointers(inputs, True), self._pointers(outputs)
        n = self.num_kernels
        ms = (ctypes.c_float * max(n, 1))()
        names = (ctypes.c_char_p * max(n, 1))()
        capi.check(capi.lib().rf_plan_execute_timed(self._h, pin, pout, self._stream(stream), ms, names, n))
        return outputs, [(names[i].decode(), float(ms[i])) for i in range(n)]

    # stepping API (sharded execution) ---------------------------------------------------------
    def begin(self, inputs, outputs, stream=None):
        pin, pout = self._pointers(inputs, True), self._pointers(outputs)
        capi.check(capi.lib().rf_plan_begin(self._h, pin, pout, self._stream(stream)))

    def exchange_bytes(self, i: int) -> int:
        return int(capi.lib().rf_plan_exchange_bytes(self._h, i))

    def exchange_local(self, i: int, send_ptr: int = 0) -> None:
        """Slab-local recurrence of exchange i; writes the slab's exit carry to the device buffer `send_ptr`."""
        capi.check(capi.lib().rf_plan_exchange_local(self._h, i, ctypes.c_void_p(send_ptr or None)))

    def exchange_apply(self, i: int, gathered_ptr: int) -> None:
        capi.check(capi.lib().rf_plan_exchange_apply(self._h, i, ctypes.c_void_p(gathered_ptr)))

    @property
    def has_interior(self) -> bool:
        return bool(capi.lib().rf_plan_has_interior(self._h))

    def interior(self) -> None:
        """rf_plan_interior: the work of this execute that does not depend on the exchange (the x/y stage of a z-sharded
        volume) -- call it between issuing the last all-gather and waiting for it."""
        capi.check(capi.lib().rf_plan_interior(self._h))

    def finish(self):
        capi.check(capi.lib().rf_plan_finish(self._h))

    def abort(self) -> None:
        """rf_plan_abort: abandon the execute this thread began and did not finish (a collective that raised)."""
        if self._h:
            capi.lib().rf_plan_abort(self._h)


# ---- coefficient design (lib/iir_coeff.cpp) ---------------------------------------------------
def gaussian_weights(sigma: float, order: int) -> List[float]:
    out = (ctypes.c_float * (order + 1))()
    capi.check(capi.lib().rf_gaussian_weights(float(sigma), int(order), out))
    return [float(v) for v in out]


def integral_image_coeff(n: int) -> List[float]:
    out = (ctypes.c_float * (n + 1))()
    capi.check(capi.lib().rf_integral_image_coeff(int(n), out))
    return [float(v) for v in out]


def overlap_feedback_coeff(a: Sequence[float], b: Sequence[float]) -> List[float]:
    fa, fb = (ctypes.c_float * len(a))(*a), (ctypes.c_float * len(b))(*b)
    out = (ctypes.c_float * (len(a) + len(b)))()
    capi.check(capi.lib().rf_overlap_feedback_coeff(fa, len(a), fb, len(b), out))
    return [float(v) for v in out]


def gaussian_box_filter(k: int, sigma: float) -> int:
    w = ctypes.c_int()
    capi.check(capi.lib().rf_gaussian_box_filter(int(k), float(sigma), ctypes.byref(w)))
    return int(w.value)


def box_difference(table, radius: int, order: Sequence[int], out=None, stream=None):
    """rf_box_difference: finite differences that turn a (higher-order) summed-area table into an iterated box filter
    (apps/box/box_filter.h).  `table` is a device tensor (..., y, x); `order` is per dimension in (x, y, z) order."""
    import torch
    if out is None:
        out = torch.empty_like(table)
    shape = tuple(table.shape)
    ext = (ctypes.c_int64 * len(shape))(*reversed(shape))
    order = list(order) + [0] * (len(shape) - len(order))
    ords = (ctypes.c_int32 * len(shape))(*order[:len(shape)])
    if not table.is_cuda or not table.is_contiguous() or not out.is_contiguous():
        raise ValueError("box_difference needs contiguous device tensors")
    capi.check(capi.lib().rf_box_difference(ctypes.c_void_p(table.data_ptr()), ctypes.c_void_p(out.data_ptr()), len(shape), ext,
                                            _dtype_code(table.dtype), int(radius), ords, Plan._stream(stream)))
    return out


def tap_filter(inputs, taps, out=None, stream=None):
    """rf_tap_filter: out(p) = sum_t weight_t * inputs[plane_t](clamp(p + offset_t)) -- the clamped difference operators
    the reference's apps put behind a summed-area table (apps/DoG/diff_gauss.cpp:176-197).  `inputs` are device tensors
    of one shape and floating-point type; `taps` is a list of (plane, offset, weight) with offset in (x, y, z) order."""
    import torch
    inputs = list(inputs)
    if out is None:
        out = torch.empty_like(inputs[0])
    shape = tuple(inputs[0].shape)
    for t in inputs + [out]:
        if not t.is_cuda or not t.is_contiguous() or tuple(t.shape) != shape or t.dtype != inputs[0].dtype:
            raise ValueError("tap_filter needs contiguous device tensors of one shape and type")
    nd = len(shape)
    ext = (ctypes.c_int64 * nd)(*reversed(shape))
    arr = (capi.Tap * len(taps))()
    for i, (plane, offset, weight) in enumerate(taps):
        arr[i].plane, arr[i].weight = int(plane), float(weight)
        offset = list(offset) + [0] * (capi.RF_MAX_DIMS - len(offset))
        for d in range(capi.RF_MAX_DIMS):
            arr[i].offset[d] = int(offset[d])
    ptrs = (ctypes.c_void_p * len(inputs))(*[t.data_ptr() for t in inputs])
    capi.check(capi.lib().rf_tap_filter(ptrs, len(inputs), ctypes.c_void_p(out.data_ptr()), nd, ext, _dtype_code(inputs[0].dtype),
                                        arr, len(taps), Plan._stream(stream)))
    return out


def stream_copy_ms(src, dst, reps: int = 5, stream=None) -> float:
    """rf_stream_copy timed with HIP events on `stream` (the current stream by default): milliseconds per copy of the f32
    image `src` (rows x width, whole 256 x 128 tiles) into `dst`, mean of `reps` after one untimed copy.  The measured
    ceiling of a read-once / write-once pass in the final pass's access shape (bench.py: roofline.copy_ceiling_gbps)."""
    import torch
    if src.dtype != torch.float32 or dst.dtype != torch.float32 or src.dim() < 2 or src.shape != dst.shape or not (src.is_contiguous() and dst.is_contiguous()):
        raise ValueError("stream_copy_ms needs two contiguous f32 device tensors of one shape")
    width, rows = int(src.shape[-1]), int(src.numel() // src.shape[-1])
    st = torch.cuda.current_stream() if stream is None else stream
    h = ctypes.c_void_p(st.cuda_stream)
    call = lambda: capi.check(capi.lib().rf_stream_copy(ctypes.c_void_p(src.data_ptr()), ctypes.c_void_p(dst.data_ptr()), width, rows, h))
    call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(st):
        e0.record(st)
        for _ in range(reps):
            call()
        e1.record(st)
    e1.synchronize()
    return e0.elapsed_time(e1) / reps


def second_order_sections(coeff: Sequence[float]) -> List[List[float]]:
    """Factors one scan {b, a1..ak} (y[i] = b x[i] + sum_j a_j y[i-j-1]) into first/second-order scans with the same
    transfer function: the poles (roots of z^k - a1 z^(k-1) - ... - ak) are paired into conjugate pairs / pairs of real
    poles, b goes to the first section.  Applied one after the other in the same direction with a ZERO border the
    sections reproduce the original scan exactly (up to rounding); this is how orders above 3 reach the fused kernels
    (the reference's audio apps sweep orders up to 29, apps/audio/audio_filter_high_order.cpp).  The inverse of
    overlap_feedback_coeff (lib/iir_coeff.cpp:236-263).  Numerically this is a cascade form: well conditioned for
    the usual low-pass designs (order-3 Gaussian: 3e-7), but for polynomials whose poles sit evenly on a circle (that
    app's dummy coefficients) f32 cascades lose accuracy quickly above order 15 -- use f64 pixels or the direct form
    there."""
    coeff = [float(c) for c in coeff]
    k = len(coeff) - 1
    if k <= 2:
        return [coeff]
    poles = np.roots([1.0] + [-c for c in coeff[1:]])
    cplx = sorted([p for p in poles if p.imag > 1e-12 * max(1.0, abs(p))], key=lambda p: -abs(p))
    real = sorted([p.real for p in poles if abs(p.imag) <= 1e-12 * max(1.0, abs(p))], key=lambda p: -abs(p))
    sections = [[1.0, float(2.0 * p.real), float(-(abs(p) ** 2))] for p in cplx]
    while len(real) >= 2:
        p, q = real.pop(0), real.pop(0)
        sections.append([1.0, float(p + q), float(-p * q)])
    if real:
        sections.append([1.0, float(real[0])])
    sections[0][0] = coeff[0]
    return sections
