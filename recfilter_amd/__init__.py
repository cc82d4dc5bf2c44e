"""recfilter_amd -- MI355X-native tiled recursive (IIR) filters.

The product is recfilter_amd/librecfilter_amd.so (hand-written gfx950 HIP kernels behind the
C ABI of include/recfilter_amd.h).  This package is the host-side mirror of the reference's
RecFilter front-end plus the ctypes plumbing to reach the library; it contains no fallback
implementation of the filter.
"""
from . import capi
from .capi import RecFilterError, build_library
from .filter import (Pointwise, RecFilter, RecFilterDim, RecFilterDimAndCausality, RecFilterSchedule,
                     RecFilterUsageError)
from .plan import (Plan, box_difference, tap_filter, stream_copy_ms, gaussian_box_filter, second_order_sections, gaussian_weights, integral_image_coeff,
                   overlap_feedback_coeff)

__all__ = [
    "capi", "RecFilterError", "build_library", "Pointwise", "RecFilter", "RecFilterDim", "RecFilterDimAndCausality",
    "RecFilterSchedule", "RecFilterUsageError", "Plan", "box_difference", "tap_filter", "second_order_sections", "gaussian_box_filter", "gaussian_weights",
    "integral_image_coeff", "overlap_feedback_coeff",
]
