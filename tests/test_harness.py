"""tools/profile_app.py: option parsing mirrors the reference's Arguments (lib/recfilter_utils.cpp:31-112) and the
sweep of scripts/profile_app.sh; on a GPU every app runs and checks against the oracle."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import profile_app as pa


def test_arguments_defaults_and_rules():
    a = pa.Arguments(["gaussian_3xy"])
    assert (a.width, a.block, a.iterations, a.nocheck) == (4096, 32, 1, False) and a.widths == [4096]
    a = pa.Arguments(["gaussian_3xy", "-w", "0"])
    assert a.widths[0] == 64 and a.widths[-1] == 4096 and len(a.widths) == 64 and a.nocheck     # profile_app.sh sweep
    a = pa.Arguments(["gaussian_3xy", "-width", "1024", "-tile", "64", "-iter", "5"])
    assert a.widths == [1024] and a.block == 64 and a.nocheck                                   # iter > 1 forces nocheck
    with pytest.raises(SystemExit):
        pa.Arguments(["gaussian_3xy", "-w", "1000", "-t", "32"])      # "Width should be a multiple of block size"
    assert abs(pa.throughput(1.0, 1 << 20) - 1000.0) < 1e-9            # lib/timing.cpp:3-5


@pytest.mark.gpu
@pytest.mark.parametrize("app", sorted(pa.APPS))
def test_every_app_runs_and_checks(app, tmp_path, capsys):
    # box filters difference f32 summed-area tables (the reference's apps do, too): the table entries grow with the
    # image area, so they are checked on a small image and against a bar scaled to that cancellation
    w = "128" if app.startswith("box") or app == "diff_gauss" else "512"
    assert pa.main([app, "-w", w, "--outdir", str(tmp_path)]) == 0
    row = capsys.readouterr().out.strip().splitlines()[-1].split("\t")
    assert int(row[0]) == int(w) and float(row[1]) > 0
    err = float(row[3].split()[-1])
    if len(row) > 4:
        # An app whose expression cancels -- the unsharp mask (1 + w) I - w Blur(I), differences of summed-area tables whose
        # entries grow with the image area -- is not held to a bare loosened bar: profile_app evaluates the SAME expression
        # with the f32 oracle (the reference's own arithmetic: its apps are float pipelines) against the f64 evaluation,
        # and ours may carry at most four times that error (or the 1e-4 of every other app, whichever is larger).
        err_f32 = float(row[4].split()[-1])
        assert err < max(1e-4, 4.0 * err_f32), (app, err, err_f32)
        assert err < 5e-3                                     # ... and never beyond what round 3 accepted bare
    else:
        assert err < 1e-4, (app, err)


@pytest.mark.gpu
def test_audio_sweeps_and_width_sweep(tmp_path, capsys):
    assert pa.main(["audio_biquads", "-w", str(1 << 20), "--outdir", str(tmp_path)]) == 0
    rows = (tmp_path / "audio_biquads.tiled.perflog").read_text().strip().splitlines()
    assert len(rows) == 15 and all(len(r.split("\t")) == 3 for r in rows)
    assert pa.main(["audio_high_order", "-w", str(1 << 16), "--outdir", str(tmp_path)]) == 0
    out = capsys.readouterr().out
    assert "max rel err" in out
    lines = out.strip().splitlines()
    assert [int(l.split()[0]) for l in lines[-15:]] == list(range(1, 30, 2))      # the app's sweep: orders 1, 3, ... 29
    for line in lines[-15:]:
        # every order in its direct form (orders above 8 on the matrix path: the tail propagation as f32 GEMMs on the
        # matrix cores), against the f64 oracle at north_star's bar
        assert float(line.split()[-1]) < 1e-4, line
