"""ctypes loader for the CPU oracle (oracle/recfilter_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never from recfilter_amd/ (the product).

The functions restate /root/reference lib/recfilter.cpp:302-343 (scan
operator), lib/iir_coeff.cpp (coefficients) and lib/coefficients.cpp (tile
matrices); see recfilter_oracle.h for the per-function citations.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Iterable, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "librecfilter_oracle.so")

F32, F64, I32, I16 = 0, 1, 2, 3
BORDER_ZERO, BORDER_CLAMP = 0, 1

_NP2ORC = {np.dtype(np.float32): F32, np.dtype(np.float64): F64,
           np.dtype(np.int32): I32, np.dtype(np.int16): I16}


class _Scan(ctypes.Structure):
    _fields_ = [("dim", ctypes.c_int), ("causal", ctypes.c_int),
                ("order", ctypes.c_int), ("coeff", ctypes.c_float * 33)]


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (recipe: oracle/Makefile)."""
    src = os.path.join(_HERE, "recfilter_oracle.c")
    src2 = os.path.join(_HERE, "recfilter_cpu_tiled.c")
    stale = (not os.path.exists(_LIB_PATH)
             or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(src), os.path.getmtime(src2)))
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "librecfilter_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def use_native_build(cache_dir: str = None) -> bool:
    """For TIMING the CPU baseline: rebuild the two C files with `-O3 -march=native` for the host this runs on (into a
    scratch directory, the in-tree checker build stays as it is) and use that library from now on.  The arithmetic is
    the same (`-ffp-contract=off` kept); returns False and keeps the portable build when the compile fails."""
    global _lib
    import tempfile
    cache_dir = cache_dir or os.path.join(tempfile.gettempdir(), f"recfilter_oracle_native_{os.getuid()}")
    os.makedirs(cache_dir, exist_ok=True)
    out = os.path.join(cache_dir, "librecfilter_oracle_native.so")
    cmd = ["gcc", "-O3", "-march=native", "-fPIC", "-std=c11", "-ffp-contract=off", "-fopenmp", "-shared", "-o", out,
           os.path.join(_HERE, "recfilter_oracle.c"), os.path.join(_HERE, "recfilter_cpu_tiled.c"), "-lm"]
    try:
        subprocess.check_call(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    except Exception:
        return False
    _lib = None
    lib(out)
    return True


def lib(path: str = None) -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if path is None:
            build()
        L = ctypes.CDLL(path or _LIB_PATH)
        i64p = ctypes.POINTER(ctypes.c_int64)
        fp = ctypes.POINTER(ctypes.c_float)
        L.orc_apply_filter.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, i64p,
                                       ctypes.POINTER(_Scan), ctypes.c_int, ctypes.c_int, ctypes.c_int]
        L.orc_apply_filter.restype = ctypes.c_int
        L.orc_apply_filter_tiled_f32.argtypes = [ctypes.c_void_p, ctypes.c_int, i64p, ctypes.POINTER(_Scan), ctypes.c_int,
                                                 ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.c_int]
        L.orc_apply_filter_tiled_f32.restype = ctypes.c_int
        L.orc_gaussian_weights.argtypes = [ctypes.c_float, ctypes.c_int, fp]
        L.orc_integral_image_coeff.argtypes = [ctypes.c_int, fp]
        L.orc_overlap_feedback_coeff.argtypes = [fp, ctypes.c_int, fp, ctypes.c_int, fp]
        L.orc_gaussian_box_filter.argtypes = [ctypes.c_int, ctypes.c_float]
        L.orc_gaussian_box_filter.restype = ctypes.c_int
        L.orc_matrix_B.argtypes = [ctypes.c_float, fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, fp]
        L.orc_matrix_R.argtypes = [fp, ctypes.c_int, ctypes.c_int, fp]
        L.orc_check_result_f32.argtypes = [fp, fp, ctypes.c_size_t, ctypes.POINTER(ctypes.c_double)]
        L.orc_check_result_f32.restype = ctypes.c_double
        L.orc_max_threads.restype = ctypes.c_int
        _lib = L
    return _lib


Scan = Tuple[int, bool, Sequence[float]]  # (dim, causal, [feedfwd, fb1..fbk])


def _scan_array(scans: Iterable[Scan]):
    scans = list(scans)
    arr = (_Scan * max(len(scans), 1))()
    for i, (dim, causal, coeff) in enumerate(scans):
        coeff = [float(c) for c in coeff]
        if len(coeff) < 2 or len(coeff) > 33:
            raise ValueError("a scan needs a feedforward and 1..32 feedback coefficients")
        arr[i].dim, arr[i].causal, arr[i].order = int(dim), int(bool(causal)), len(coeff) - 1
        for j, c in enumerate(coeff):
            arr[i].coeff[j] = c
    return arr, len(scans)


def apply_filter(image: np.ndarray, scans: Iterable[Scan], clamped: bool = False,
                 threads: int = 1, inplace: bool = False) -> np.ndarray:
    """Untiled reference result.  `image` is indexed [..., z, y, x] (numpy C order, x fastest),
    scan dim 0 = x = last numpy axis.  Returns a new array of the same dtype (inplace=True: filters a C-contiguous
    `image` itself, for volumes too large to copy)."""
    if image.dtype not in _NP2ORC:
        raise TypeError(f"unsupported dtype {image.dtype}")
    out = image if (inplace and image.flags["C_CONTIGUOUS"]) else np.ascontiguousarray(image).copy()
    ext = (ctypes.c_int64 * out.ndim)(*reversed(out.shape))
    arr, n = _scan_array(scans)
    rc = lib().orc_apply_filter(out.ctypes.data_as(ctypes.c_void_p), _NP2ORC[out.dtype], out.ndim,
                                ext, arr, n, BORDER_CLAMP if clamped else BORDER_ZERO, int(threads))
    if rc:
        raise RuntimeError(f"orc_apply_filter failed with {rc}")
    return out


def apply_filter_tiled(image: np.ndarray, scans: Iterable[Scan], clamped: bool = False, tile=32, threads: int = 1,
                       inplace: bool = False) -> np.ndarray:
    """The tiled CPU schedule's counterpart (recfilter_cpu_tiled.c): pass 1 / carry / pass 2 per dimension, tiles in
    parallel.  f32 only; `tile` = one width for every filtered dimension, or a list per dimension in (x, y, z) order
    (0 = untiled).  Same result as apply_filter up to f32 rounding."""
    if image.dtype != np.float32:
        raise TypeError("the tiled CPU counterpart is f32 only")
    out = image if (inplace and image.flags["C_CONTIGUOUS"]) else np.ascontiguousarray(image).copy()
    ext = (ctypes.c_int64 * out.ndim)(*reversed(out.shape))
    arr, n = _scan_array(scans)
    tiles = [int(tile)] * out.ndim if np.isscalar(tile) else [int(t) for t in tile]
    tl = (ctypes.c_int * out.ndim)(*tiles)
    rc = lib().orc_apply_filter_tiled_f32(out.ctypes.data_as(ctypes.c_void_p), out.ndim, ext, arr, n,
                                          BORDER_CLAMP if clamped else BORDER_ZERO, tl, int(threads))
    if rc:
        raise RuntimeError(f"orc_apply_filter_tiled_f32 failed with {rc}")
    return out


def gaussian_weights(sigma: float, order: int) -> np.ndarray:
    n = 3 if order not in (1, 2) else order
    out = (ctypes.c_float * 4)()
    lib().orc_gaussian_weights(float(sigma), int(order), out)
    return np.array(out[: n + 1], dtype=np.float32)


def integral_image_coeff(n: int) -> np.ndarray:
    out = (ctypes.c_float * (n + 1))()
    lib().orc_integral_image_coeff(int(n), out)
    return np.array(out[:], dtype=np.float32)


def overlap_feedback_coeff(a: Sequence[float], b: Sequence[float]) -> np.ndarray:
    fa = (ctypes.c_float * len(a))(*a)
    fb = (ctypes.c_float * len(b))(*b)
    out = (ctypes.c_float * (len(a) + len(b)))()
    lib().orc_overlap_feedback_coeff(fa, len(a), fb, len(b), out)
    return np.array(out[:], dtype=np.float32)


def gaussian_box_filter(k: int, sigma: float) -> int:
    return int(lib().orc_gaussian_box_filter(int(k), float(sigma)))


def matrix_B(feedfwd: float, feedback: Sequence[float], tile: int, clamp_border: bool = False) -> np.ndarray:
    fb = (ctypes.c_float * len(feedback))(*feedback)
    out = (ctypes.c_float * (tile * tile))()
    lib().orc_matrix_B(float(feedfwd), fb, len(feedback), int(tile), int(clamp_border), out)
    return np.array(out[:], dtype=np.float32).reshape(tile, tile)


def matrix_R(feedback: Sequence[float], tile: int) -> np.ndarray:
    fb = (ctypes.c_float * len(feedback))(*feedback)
    out = (ctypes.c_float * (tile * len(feedback)))()
    lib().orc_matrix_R(fb, len(feedback), int(tile), out)
    return np.array(out[:], dtype=np.float32).reshape(tile, len(feedback))


def check_result(ref: np.ndarray, out: np.ndarray) -> Tuple[float, float]:
    """(max %, mean %) relative error as the reference's CheckResult prints them."""
    r = np.ascontiguousarray(ref, dtype=np.float32).ravel()
    o = np.ascontiguousarray(out, dtype=np.float32).ravel()
    mean = ctypes.c_double()
    fp = ctypes.POINTER(ctypes.c_float)
    mx = lib().orc_check_result_f32(r.ctypes.data_as(fp), o.ctypes.data_as(fp), r.size, ctypes.byref(mean))
    return float(mx), float(mean.value)


def max_threads() -> int:
    return int(lib().orc_max_threads())
