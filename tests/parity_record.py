"""Record mode of the parity tests (TEST INFRASTRUCTURE).

With RF_RECORD_PARITY=<file> in the environment, every evaluation of a parity metric (ref_cases.rel_err*) is appended to
<file> as one JSON line, together with the oracle call its reference result came from (the filter, the border mode, the shape
and the value range of the input) and the pointwise epilogue, if any.  tests/metric_margin.py replays those lines on the CPU:
the SAME filter through the f32 oracle -- the reference operator itself, serial and untiled, in the pixel type
(/root/reference/lib/recfilter.cpp:302-343) -- on many seeds, judged by the SAME metric against the f64 oracle.  An assertion
the f32 reference does not pass with a wide margin is a defect of the test, not evidence about a kernel (one such
assertion turned the round's GPU run red).

Without the variable nothing is recorded and the hooks cost one dictionary lookup.
"""
from __future__ import annotations

import json
import os

import numpy as np

_PATH = os.environ.get("RF_RECORD_PARITY")
_last_call = None      # the most recent oracle.apply_filter call of this process
_epilogue = None       # set by ref_cases.pointwise_want for the metric evaluation that follows
_entry = None          # "auto" while ref_cases.rel_err chooses between strict and floored


def enabled() -> bool:
    return bool(_PATH)


def install() -> None:
    """Wrap oracle.apply_filter (tests call it through the module attribute) so that the filter behind a reference result
    is known when the metric is evaluated."""
    if not enabled():
        return
    import oracle
    if getattr(oracle.apply_filter, "_recorded", False):
        return
    inner = oracle.apply_filter

    def apply_filter(image, scans, clamped=False, *args, **kwargs):
        global _last_call
        scans = list(scans)
        img = np.asarray(image)
        flat = img.reshape(-1)[:: max(1, img.size // (1 << 20))]          # a sample is enough for the range
        _last_call = dict(shape=list(img.shape), in_dtype=str(img.dtype), clamped=bool(clamped),
                          scans=[[int(d), bool(c), [float(v) for v in w]] for d, c, w in scans],
                          in_lo=float(flat.min()) if flat.size else 0.0, in_hi=float(flat.max()) if flat.size else 0.0)
        return inner(image, scans, clamped, *args, **kwargs)

    apply_filter._recorded = True
    oracle.apply_filter = apply_filter


def set_entry(name):
    global _entry
    _entry = name


def note_epilogue(epilogue, x) -> None:
    global _epilogue
    if enabled():
        _epilogue = [float(v) for v in epilogue]


def note_metric(metric, value, out, ref, out_dtype=None) -> None:
    global _epilogue
    if not enabled():
        return
    rec = dict(test=os.environ.get("PYTEST_CURRENT_TEST", "").split(" ")[0], metric=metric, entry=_entry or metric,
               value=float(value), ref_shape=list(np.shape(ref)), out_dtype=str(out_dtype) if out_dtype is not None else None,
               epilogue=_epilogue, oracle_call=_last_call)
    _epilogue = None
    with open(_PATH, "a") as f:
        f.write(json.dumps(rec) + "\n")
