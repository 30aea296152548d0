"""
A few seeds of every randomised differential run under tests/fuzz/ (the full sweeps — thousands of seeds, DESIGN.md §4 — are run
by hand on the GPU box): irregular label images, random tile geometry, random pipelines; each script asserts parity with the oracle
(or, for the runner, with one call per position) seed by seed.
"""

import runpy
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
SCRIPTS = Path(__file__).resolve().parent / "fuzz"


@pytest.mark.parametrize("script,args", [
    ("fuzz_features.py", ["0", "6"]),
    ("fuzz_features.py", ["100", "4", "extras"]),
    ("fuzz_dynamics.py", ["0", "15"]),
    ("fuzz_stager_tracker.py", ["0", "40"]),
    ("fuzz_tiles_drift.py", ["0", "8"]),
    ("fuzz_runner.py", ["0", "6"]),
    ("fuzz_runner.py", ["0", "5", "timelapse"]),
    ("fuzz_overlap.py", ["0", "20"]),
    ("fuzz_volume.py", ["540", "30"]),  # (seed 553: a stack without any object)
    ("fuzz_traps.py", ["108", "8"]),  # (seed 114: no usable template — both sides raise "No valid tiles found.")
])
def test_randomised_differential_runs(engine, monkeypatch, capsys, script, args):
    monkeypatch.setattr(sys, "argv", [str(SCRIPTS / script), *args])
    monkeypatch.chdir(SCRIPTS.parents[1])  # (the scripts put "." and "tests" on sys.path)
    runpy.run_path(str(SCRIPTS / script), run_name="__main__")
    out = capsys.readouterr().out
    assert out.count(": ok") == int(args[1]), out[-400:]
