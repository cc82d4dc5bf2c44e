"""ctypes binding of the C ABI declared in include/recfilter_amd.h.

This is plumbing: every call goes straight into librecfilter_amd.so (hand-written gfx950
kernels).  There is no Python or CPU implementation of the filter behind it -- if the shared
library is missing, importing this module raises.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
# RECFILTER_AMD_LIB: load another build of the same library (A/B timing of kernel changes)
LIB_PATH = os.environ.get("RECFILTER_AMD_LIB") or os.path.join(_PKG, "librecfilter_amd.so")
CSRC = os.path.join(_PKG, "csrc")

RF_MAX_DIMS = 3
RF_MAX_ORDER = 32
RF_ABI = 3          # revision of include/recfilter_amd.h this module mirrors (rf_filter_desc.abi)
RF_MAX_SCANS = 32
RF_MAX_PLANES = 16
RF_DEVICE_HOST_ONLY = -2

RF_OK, RF_ERR_INVALID_ARG, RF_ERR_UNSUPPORTED, RF_ERR_HIP, RF_ERR_NOMEM, RF_ERR_STATE = range(6)
RF_F32, RF_F64, RF_I32, RF_I16 = range(4)
RF_BORDER_ZERO, RF_BORDER_CLAMP = 0, 1
RF_POINTWISE_PRE, RF_POINTWISE_POST = 1, 2
RF_IN_PIXEL, RF_IN_U8 = 0, 1
RF_PATH_AUTO, RF_PATH_UNTILED, RF_PATH_TILED_GENERIC, RF_PATH_TILED_FUSED, RF_PATH_TILED_OVERLAPPED, RF_PATH_TILED_MATRIX = range(6)
# rf_filter_desc.flags (plan options; the library reads no environment variable)
RF_PLAN_FORCE_EXCHANGE, RF_PLAN_TILED_ONLY, RF_PLAN_NO_CASCADE, RF_PLAN_NO_SECTIONS = 0x01, 0x02, 0x04, 0x08
RF_PLAN_NO_PLANE_BATCH, RF_PLAN_STREAM_PASS1, RF_PLAN_STAGED_PASS1, RF_PLAN_LATE_EXCHANGE = 0x10, 0x20, 0x40, 0x80
RF_PLAN_SERIAL_UNTILED = 0x01000000
RF_PLAN_MFMA_PASS1 = 0x02000000
RF_PLAN_WALK_PASS1 = 0x04000000
RF_PLAN_NO_OVERLAP = 0x08000000
RF_PLAN_INPLACE_Z = 0x10000000


def RF_PLAN_TILE_ROWS(n: int) -> int:
    return (int(n) & 0xff) << 8


def RF_PLAN_TILE_PLANES(n: int) -> int:
    return (int(n) & 0xff) << 16


PATH_NAMES = {RF_PATH_AUTO: "auto", RF_PATH_UNTILED: "untiled",
              RF_PATH_TILED_GENERIC: "tiled_generic", RF_PATH_TILED_FUSED: "tiled_fused",
              RF_PATH_TILED_OVERLAPPED: "tiled_overlapped", RF_PATH_TILED_MATRIX: "tiled_matrix"}

# every symbol include/recfilter_amd.h declares
EXPORTED_SYMBOLS = [
    "rf_plan_create", "rf_plan_destroy", "rf_plan_workspace_bytes", "rf_plan_num_instances", "rf_plan_path", "rf_plan_tiles",
    "rf_plan_num_kernels", "rf_plan_execute", "rf_plan_execute_timed", "rf_plan_num_exchanges",
    "rf_plan_exchange_bytes",
    "rf_plan_begin", "rf_plan_exchange_local", "rf_plan_exchange_apply", "rf_plan_has_interior", "rf_plan_interior", "rf_plan_finish", "rf_plan_abort",
    "rf_plan_table", "rf_plan_debug_buffer", "rf_gaussian_weights", "rf_integral_image_coeff", "rf_overlap_feedback_coeff",
    "rf_gaussian_box_filter", "rf_box_difference", "rf_tap_filter", "rf_stream_copy", "rf_last_error_string", "rf_version", "rf_device_count",
]


class ScanDesc(ctypes.Structure):
    _fields_ = [("dim", ctypes.c_int32), ("causal", ctypes.c_int32), ("order", ctypes.c_int32),
                ("feedfwd", ctypes.c_float), ("feedback", ctypes.c_float * RF_MAX_ORDER)]


class PointwiseDesc(ctypes.Structure):
    _fields_ = [("flags", ctypes.c_int32), ("pre_scale", ctypes.c_float), ("pre_bias", ctypes.c_float),
                ("post_filtered", ctypes.c_float), ("post_input", ctypes.c_float), ("post_bias", ctypes.c_float),
                ("in_dtype", ctypes.c_int32)]


class FilterDesc(ctypes.Structure):
    _fields_ = [("ndim", ctypes.c_int32), ("abi", ctypes.c_uint32), ("extent", ctypes.c_int64 * RF_MAX_DIMS),
                ("dtype", ctypes.c_int32), ("n_planes", ctypes.c_int32), ("border", ctypes.c_int32),
                ("n_scans", ctypes.c_int32), ("scans", ctypes.POINTER(ScanDesc)),
                ("tile", ctypes.c_int32 * RF_MAX_DIMS), ("path", ctypes.c_int32),
                ("device", ctypes.c_int32), ("shard_rank", ctypes.c_int32), ("shard_world", ctypes.c_int32),
                ("pointwise", PointwiseDesc), ("shard_extents", ctypes.POINTER(ctypes.c_int64)),
                ("flags", ctypes.c_uint32)]


class Tap(ctypes.Structure):
    _fields_ = [("plane", ctypes.c_int32), ("offset", ctypes.c_int32 * RF_MAX_DIMS), ("weight", ctypes.c_float)]


class RecFilterError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"recfilter_amd status {status}: {message}")
        self.status = status


def build_library(force: bool = False) -> str:
    """Compile every HIP source for gfx950 into recfilter_amd/librecfilter_amd.so (in-tree)."""
    cmd = ["make", "-C", CSRC, "-j4"]
    if force:
        cmd.append("-B")
    subprocess.check_call(cmd, stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def lib() -> ctypes.CDLL:
    """Load librecfilter_amd.so.  torch is imported first so that both share ONE HIP runtime
    (torch bundles its own libamdhip64 with the same SONAME as /opt/rocm's)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `make -C {CSRC}` (or __graft_entry__.build()); "
            "recfilter_amd has no fallback implementation")
    try:
        import torch  # noqa: F401  (loads torch's libamdhip64 before ours resolves the SONAME)
    except Exception:  # pragma: no cover - torch is plumbing, the library works without it
        pass
    L = ctypes.CDLL(LIB_PATH)
    vp, vpp = ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p)
    fp = ctypes.POINTER(ctypes.c_float)
    L.rf_plan_create.argtypes = [ctypes.POINTER(FilterDesc), vpp]
    L.rf_plan_destroy.argtypes = [vp]
    L.rf_plan_workspace_bytes.argtypes = [vp]
    L.rf_plan_workspace_bytes.restype = ctypes.c_size_t
    L.rf_plan_num_instances.argtypes = [vp]
    L.rf_plan_path.argtypes = [vp]
    L.rf_plan_tiles.argtypes = [vp, ctypes.POINTER(ctypes.c_int32)]
    L.rf_plan_num_kernels.argtypes = [vp]
    L.rf_plan_execute.argtypes = [vp, vpp, vpp, vp]
    L.rf_plan_execute_timed.argtypes = [vp, vpp, vpp, vp, fp, ctypes.POINTER(ctypes.c_char_p), ctypes.c_int]
    L.rf_plan_num_exchanges.argtypes = [vp]
    L.rf_plan_begin.argtypes = [vp, vpp, vpp, vp]
    L.rf_plan_exchange_bytes.argtypes = [vp, ctypes.c_int]
    L.rf_plan_exchange_bytes.restype = ctypes.c_size_t
    L.rf_plan_exchange_local.argtypes = [vp, ctypes.c_int, vp]
    L.rf_plan_exchange_apply.argtypes = [vp, ctypes.c_int, vp]
    L.rf_plan_finish.argtypes = [vp]
    L.rf_plan_abort.argtypes = [vp]
    L.rf_plan_has_interior.argtypes = [vp]
    L.rf_plan_interior.argtypes = [vp]
    L.rf_plan_table.argtypes = [vp, ctypes.c_char_p, ctypes.POINTER(ctypes.c_double), ctypes.c_size_t,
                                ctypes.POINTER(ctypes.c_size_t)]
    L.rf_plan_debug_buffer.argtypes = [vp, ctypes.c_int, vpp, ctypes.POINTER(ctypes.c_size_t)]
    L.rf_gaussian_weights.argtypes = [ctypes.c_float, ctypes.c_int, fp]
    L.rf_integral_image_coeff.argtypes = [ctypes.c_int, fp]
    L.rf_overlap_feedback_coeff.argtypes = [fp, ctypes.c_int, fp, ctypes.c_int, fp]
    L.rf_gaussian_box_filter.argtypes = [ctypes.c_int, ctypes.c_float, ctypes.POINTER(ctypes.c_int)]
    L.rf_box_difference.argtypes = [vp, vp, ctypes.c_int, ctypes.POINTER(ctypes.c_int64), ctypes.c_int, ctypes.c_int,
                                    ctypes.POINTER(ctypes.c_int32), vp]
    L.rf_stream_copy.argtypes = [vp, vp, ctypes.c_int64, ctypes.c_int64, vp]
    L.rf_tap_filter.argtypes = [vpp, ctypes.c_int, vp, ctypes.c_int, ctypes.POINTER(ctypes.c_int64), ctypes.c_int,
                                ctypes.POINTER(Tap), ctypes.c_int, vp]
    L.rf_last_error_string.restype = ctypes.c_char_p
    L.rf_version.restype = ctypes.c_char_p
    _lib = L
    return L


def check(status: int) -> None:
    if status != RF_OK:
        raise RecFilterError(status, lib().rf_last_error_string().decode())
