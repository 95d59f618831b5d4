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


def test_result_line_stays_within_the_size_the_driver_parses():
    # the round-4 line was 22 112 bytes and BENCH_r04.json.parsed came back null.  The LAST stdout line is now a
    # compact object (bench.compact_line) of at most 4096 bytes; the long material goes to EXTRAS lines + bench_extras.json.
    sys.path.insert(0, ROOT)
    import bench
    assert bench.LINE_LIMIT == 4096
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench.json"))):
        line = json.load(open(path))
        short = json.dumps(bench.compact_line(line), separators=(",", ":"))
        assert len(short.encode()) <= bench.LINE_LIMIT, (path, len(short))
        back = json.loads(short)
        for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                    "dtype", "data", "config", "roofline", "cpu_baseline"):
            assert key in back, (path, key)
        assert "extras" not in back
        rnd = int(os.path.basename(path)[1:3])
        if rnd >= 5:  # from round 5 on the committed line IS the compact line
            assert os.path.getsize(path) <= bench.LINE_LIMIT + 1, path
            assert os.path.exists(path.replace("_bench.json", "_bench_extras.json")), "the sidecar of the committed line"


def test_emit_prints_the_compact_line_last_and_refuses_an_oversized_one(tmp_path, monkeypatch):
    import io
    sys.path.insert(0, ROOT)
    import bench
    monkeypatch.chdir(tmp_path)
    full = {"metric": "m", "value": 1.0, "unit": "u", "n_gpus": 1, "steps": 1, "warmup": 0, "ms_per_step": 1.0, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic", "config": {"workload": "w" * 1000},
            "roofline": {"bound": "valu_issue", "achieved": 1.0, "peak": 2.0, "unit": "x", "frac": 0.5, "traffic": None, "formula": "f" * 5000},
            "cpu_baseline": {"value": 1.0, "unit": "u", "cores": 1, "kind": "port", "sample": "s" * 3000},
            "extras": {"a": {"note": "n" * 20000}, "b": {"x": 1}}}
    buf = io.StringIO()
    last = bench.emit(full, stream=buf)
    lines = buf.getvalue().splitlines()
    assert lines[-1] == last and lines[-1].startswith('{"metric"') and len(lines[-1].encode()) <= bench.LINE_LIMIT
    assert [ln[:10] for ln in lines[:-1]] == ["EXTRAS {\"a", "EXTRAS {\"b", "EXTRAS {\"h"]
    side = json.load(open(tmp_path / bench.SIDECAR))
    assert side["extras"]["a"]["note"] == "n" * 20000 and json.loads(last)["extras_keys"] == ["a", "b"]
    full["scaling_note"] = "z" * 5000
    with pytest.raises(SystemExit):
        bench.emit(full, stream=io.StringIO())


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
    for flag in ("--gpus", "--steps", "--warmup", "--restarts", "--restarts-total", "--single-process", "--dry-launch"):
        assert flag in out.stdout


def _bare_env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    return env


def test_bare_gpus_2_starts_two_ranks_itself():
    # `bench.py --gpus N` without an external launcher used to run ONE rank and print n_gpus 1.  Now the bare
    # call is the launcher: two gloo ranks (stub step, no GPU, no oracle), rank 0's line relayed, ranks_seen == 2.
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-launch", "--steps", "2"],
                         capture_output=True, text=True, timeout=300, env=_bare_env())
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1
    assert out.stdout.splitlines()[-1] == lines[0] and len(lines[0].encode()) <= 4096     # the result line is the LAST line, and short
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and len(line["ms_per_step_per_rank"]) == 2
    assert line["dry_launch"] is True and line["value"] is None          # a rehearsal never carries a number
    assert line["best_restart"] == 5                                      # the stub's winner lives on rank 1: its tour crossed ranks
    assert line["launcher"] == "torch.distributed.run"
    # both readings of configs[3] in ONE line — R per GPU (weak, the headline) and 256 in all (strong) — and
    # the note that says the strong one is flat by design
    assert line["scaling"] == "weak"
    assert line["weak_per_gpu"]["restarts_per_rank"] == 4 and line["weak_per_gpu"]["ms_per_step"] > 0
    st = line["strong_256_total"]
    assert st["restarts_per_rank"] == 128 and st["ms_per_step"] > 0 and st["stub_units_all_ranks"] == 256 * 2  # 2 steps
    assert st["value"] is None and line["weak_per_gpu"]["value"] is None
    assert "flat by design" in line["scaling_note"] and "r02_xcu_sync_probe" in line["scaling_note"]
    assert st["cross_cu_round_us"] == [1.4, 2.1]


def test_rank_count_mismatch_is_an_error_not_a_one_gpu_run():
    env = _bare_env()
    env.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-launch"],
                         capture_output=True, text=True, timeout=120, env=env)
    assert out.returncode != 0 and "refusing" in (out.stderr + out.stdout)
