"""GPU tests of the C++ host mirror + CLI façade (teeline_amd/host_cpp): same output format as teeline-cli's
print_solution (teeline-cli/src/main.rs:645-652) and the reference's published / golden numbers."""
import json
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cli():
    from teeline_amd import build
    return build.build_cli()


@pytest.fixture(scope="module")
def goldens(golden_dir):
    with open(os.path.join(golden_dir, "goldens.json")) as fh:
        return json.load(fh)


def run(cli, *args):
    r = subprocess.run([cli, *args], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    head, route = r.stdout.strip().split("\n")
    cost, flag = head.split()
    return cost, flag, [int(v) for v in route.split()]


def test_cli_berlin52_matches_reference_numbers(cli, tsplib_dir, goldens):
    f = os.path.join(tsplib_dir, "berlin52.tsp")
    g = goldens["berlin52"]
    cost, flag, route = run(cli, "solve", "nn", "-i", f)
    assert (cost, flag) == ("8980.91797", "0")                                  # bench/baseline-solvers.tsv:2-6
    cost, flag, route = run(cli, "solve", "2opt", "-i", f)                       # `fast` preset nn -> 2opt (README.md:385)
    assert cost == "8384.18848" and route == g["nn_two_opt"]["route_ids"]
    cost, flag, route = run(cli, "solve", "2opt", "--no-seed", "-i", f)          # docs/benchmarks.md:28 (9368.32)
    assert cost == "9368.31836" and route == g["identity_two_opt"]["route_ids"]
    cost, flag, route = run(cli, "solve", "3opt", "-i", f)                       # docs/benchmarks.md:29 (7742.65)
    assert cost == "7742.64697" and route == g["nn_three_opt"]["route_ids"]
    cost, flag, route = run(cli, "solve", "oropt", "-i", f)                      # docs/benchmarks.md:48 (8 097.48)
    assert cost == "8097.47607" and route == g["nn_or_opt"]["route_ids"]
    cost, flag, route = run(cli, "solve", "lk", "-i", f, "--seed", "1")          # bench/baseline-solvers.tsv:17-21
    assert cost == "7544.36572" and sorted(route) == list(range(1, 53))


def test_cli_explicit_and_errors(cli, tsplib_dir, goldens):
    cost, flag, route = run(cli, "solve", "2opt", "-i", os.path.join(tsplib_dir, "gr17.tsp"))
    assert cost == goldens["gr17"]["identity_two_opt"]["cost"]
    assert [r - 1 for r in route] == goldens["gr17"]["identity_two_opt"]["route_pos"]
    r = subprocess.run([cli, "solve", "2opt", "-i", "/nonexistent.tsp"], capture_output=True, text=True)
    assert r.returncode == 1 and "tsplib: failed to read file" in r.stderr
    r = subprocess.run([cli, "solve", "sa", "-i", os.path.join(tsplib_dir, "berlin52.tsp")], capture_output=True, text=True)
    assert r.returncode == 1 and "unknown solver" in r.stderr
    tiny = "NAME: t\nTYPE: TSP\nDIMENSION: 2\nEDGE_WEIGHT_TYPE: EUC_2D\nNODE_COORD_SECTION\n1 0 0\n2 1 1\nEOF\n"
    r = subprocess.run([cli, "solve", "2opt", "--no-seed"], input=tiny, capture_output=True, text=True)
    assert r.returncode == 101 and "panicked" in r.stderr  # the reference panics on n < 3 (two_opt.rs:17,29)


def test_python_pipeline_facade(tsplib_dir, goldens):
    # pipeline.rs:170-187 (nn -> 2opt never worse than nn) and the `fast` preset numbers (README.md:385)
    import teeline_amd as TA
    prob = TA.tsplib.read_from_file(os.path.join(tsplib_dir, "berlin52.tsp")).problem()
    out = TA.pipeline.run_pipeline_stages(prob, ["nn", "2opt", "oropt"])
    assert [o.name for o in out] == ["nn", "2opt", "oropt"]
    assert f"{float(out[0].solution.total):.5f}" == "8980.91797" and f"{float(out[1].solution.total):.5f}" == "8384.18848"
    assert out[1].solution.total <= out[0].solution.total * 1.001 and out[2].solution.total <= out[1].solution.total
    assert all(TA.validate_tour(o.solution.route(), prob) and o.duration_ms >= 0 for o in out)
    with pytest.raises(ValueError):
        TA.pipeline.run_pipeline_stages(prob, ["sa"])
