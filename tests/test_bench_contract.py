"""CPU-side checks of the measurement contract: the committed bench line (profiles/rNN_bench.json, produced by bench.py on the GPU
box) carries the metric BASELINE.json names, the fields the driver reads, a roofline that is re-derivable from the committed PMC
summary, and the CPU baseline; bench.py's flags parse."""
import glob
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def latest(pattern):
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
    return files[-1] if files else None


def test_committed_bench_line_follows_the_contract():
    path = latest("r*_bench.json")
    if path is None:
        pytest.skip("no committed bench line")
    line = json.load(open(path))
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert line["metric"] == base["metric"]
    for key in ("value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert key in line, key
    assert line["higher_is_better"] is True and line["scaling"] in ("weak", "strong") and line["vs_baseline"] is None
    assert "workload" in line["config"] and line["dtype"] == "f32"
    r = line["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["frac"] is None or 0.0 < r["frac"] <= 1.0
    c = line["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] in ("port", "reference")
    assert line["parity_checked_restarts"] >= 1 and line["parity_mismatches"] == []


def test_roofline_is_rederivable_from_the_committed_counters():
    bench, pmc = latest("r*_bench.json"), latest("r*_valu_roofline.json")
    if bench is None or pmc is None:
        pytest.skip("no committed evidence")
    line, prof = json.load(open(bench)), json.load(open(pmc))
    r = line["roofline"]
    if r.get("frac") is None:
        pytest.skip("the committed line carries no fraction")
    # frac = SQ_INSTS_VALU / (SIMDs x kernel seconds x clock / 2)   (profiles/README.md)
    frac = prof["SQ_INSTS_VALU"] / (r["simds"] * r["kernel_ms_avg"] * 1e-3 * r["clock_mhz_live"] * 1e6 / 2)
    assert abs(frac - r["frac"]) <= 0.02 * r["frac"]


def test_bench_flags_parse():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup", "--restarts", "--restarts-total"):
        assert flag in out.stdout
