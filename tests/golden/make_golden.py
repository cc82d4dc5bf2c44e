#!/usr/bin/env python3
"""Generates tests/golden/vectors.npz: seeded inputs and expected outputs of the scan operator for a few small filters.

The reference cannot run in this image (every translation unit needs Halide), so the expected outputs
come from the CPU oracle (oracle/, a restatement of lib/recfilter.cpp:302-343), which is itself pinned against the
reference's tests' loop references and known answers (tests/test_oracle_golden.py).  The 64x64 cfg3 case carries the
values SURVEY.md 8(c) quotes from an untiled f32 run of the reference's operator (first / last / centre / sum) as an
independent anchor; make_golden.py refuses to write the file if the oracle disagrees with them.

    python tests/golden/make_golden.py        # rewrites tests/golden/vectors.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(HERE))

import oracle
import ref_cases as rc

CASES = {
    # name: (shape, dtype, scans, clamped, seed)
    "cfg3_gauss2_xy_64": ((64, 64), np.float32, rc.xy_pm(rc.GAUSS2), True, 1234),
    "generic_xy_48x80": ((48, 80), np.float32, rc.REFERENCE_TESTS["test_generic_xy"]["scans"], False, 11),
    "generic_xyz_8x24x32": ((8, 24, 32), np.float32, rc.REFERENCE_TESTS["test_generic_xyz"]["scans"], False, 12),
    "bicubic_clamped_40x272": ((40, 272), np.float32, rc.xy_pm(rc.BICUBIC_COEFF), True, 13),
    "gauss3_clamped_70x176": ((70, 176), np.float32, rc.xy_pm(rc.GAUSS3), True, 14),
    "sat_int32_64x256": ((64, 256), np.int32, [(0, True, [1.0, 1.0]), (1, True, [1.0, 1.0])], False, 15),
    "signal_1d_8192": ((8192,), np.float32, [(0, True, rc.GAUSS2), (0, False, rc.GAUSS2)], False, 16),
}


def build():
    out = {}
    for name, (shape, dtype, scans, clamped, seed) in CASES.items():
        img = rc.random_image(shape, dtype, seed)
        if np.issubdtype(dtype, np.integer):
            want = oracle.apply_filter(img, scans, clamped)
        else:
            want = oracle.apply_filter(img.astype(np.float64), scans, clamped)
        out[name + "/input"] = img
        out[name + "/expected"] = want if np.issubdtype(dtype, np.integer) else want.astype(np.float32)
    # independent anchor from SURVEY.md 8(c): untiled f32, numpy default_rng(1234).random(float32), cfg3 filter
    f32 = oracle.apply_filter(out["cfg3_gauss2_xy_64/input"], CASES["cfg3_gauss2_xy_64"][2], True)
    first, last, centre, total = rc.CFG3_RANDOM64
    got = (float(f32[0, 0]), float(f32[-1, -1]), float(f32[32, 32]), float(f32.sum(dtype=np.float64)))
    for g, w in zip(got, (first, last, centre, total)):
        if abs(g - w) > 2e-6 * max(1.0, abs(w)):
            raise SystemExit(f"oracle disagrees with the SURVEY anchor: {got} vs {rc.CFG3_RANDOM64}")
    return out


if __name__ == "__main__":
    np.savez_compressed(os.path.join(HERE, "vectors.npz"), **build())
    print("wrote", os.path.join(HERE, "vectors.npz"), os.path.getsize(os.path.join(HERE, "vectors.npz")), "bytes")
