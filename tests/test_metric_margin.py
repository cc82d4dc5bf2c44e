"""Every f32 parity assertion the GPU suite recorded (tests/golden/parity_assertions.jsonl) must be one the f32 reference operator
itself passes: a quick replay (two seeds per assertion; the 200-seed run is profiles/r6/metric_margin_200_seeds.txt) through
tests/metric_margin.py finds no ill-conditioned metric."""
import multiprocessing as mp
import os

import metric_margin as mm


def test_no_recorded_assertion_is_ill_conditioned():
    asserts, skipped = mm.load_assertions()
    assert len(asserts) > 500, "the record of the GPU suite's assertions is missing or truncated"
    with mp.get_context("spawn").Pool(min(4, os.cpu_count() or 1)) as pool:
        results = pool.map(mm._job, [(a, 2) for a in asserts], chunksize=4)
    bad = []
    for a, (worst, worst_norm, _shape) in zip(asserts, results):
        margin = 1e-4 / worst if worst > 0 else float("inf")
        if mm.verdict(margin, worst, worst_norm) == "ILL-CONDITIONED":
            bad.append((sorted(a["tests"])[0], worst, worst_norm))
        assert a["recorded_worst"] < 1e-4 or a["entry"] not in ("auto", "local", "scaled"), a      # what the HIP path recorded
    assert not bad, bad
