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
    ids = [int(v) for v in route.split()]
    # print_solution (teeline-cli/src/main.rs:645-652): "{:.5} {flag}\n", every id followed by ONE space, "\n" — whole stdout
    assert r.stdout == f"{cost} {flag}\n" + "".join(f"{v} " for v in ids) + "\n"
    return cost, flag, ids


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
    import numpy as np
    import _oracle as O
    import _tsplib as T
    cost, flag, route = run(cli, "solve", "2opt", "--no-seed", "-i", os.path.join(tsplib_dir, "gr17.tsp"))
    assert cost == goldens["gr17"]["identity_two_opt"]["cost"]
    assert [r - 1 for r in route] == goldens["gr17"]["identity_two_opt"]["route_pos"]
    # EXPLICIT / GEO files are auto-seeded like every other (`solve 2opt` = pipeline(nn, 2opt), main.rs:387-389): the NN
    # stage walks the matrix (nearest_neighbor.rs:44-63 over distance_matrix.rs:259-297)
    for name in ("gr17", "bays29", "burma14"):
        e = T.parse_tsplib(os.path.join(tsplib_dir, f"{name}.tsp"))
        packed = e["packed"] if e["packed"] is not None else O.dm_build_packed(e["xy"], geo=True)
        rc, nn, cnn = O.nearest_neighbor(None, packed, e["n"], 3)
        rc, p, c, st = O.two_opt(None, packed, e["n"], init=nn)
        cost, flag, route = run(cli, "solve", "2opt", "-i", os.path.join(tsplib_dir, f"{name}.tsp"))
        assert cost == f"{float(c):.5f}" and route == e["ids"][p].tolist()
        cost, flag, route = run(cli, "solve", "nn", "-i", os.path.join(tsplib_dir, f"{name}.tsp"))
        assert cost == f"{float(cnn):.5f}" and route == e["ids"][nn].tolist()
    r = subprocess.run([cli, "solve", "2opt", "-i", "/nonexistent.tsp"], capture_output=True, text=True)
    assert r.returncode == 1 and "File doesnt exists" in r.stderr and r.stdout == ""
    r = subprocess.run([cli, "solve", "sa", "-i", os.path.join(tsplib_dir, "berlin52.tsp")], capture_output=True, text=True)
    assert r.returncode == 1 and "not accelerated by this build" in r.stderr
    r = subprocess.run([cli, "solve", "bogus", "-i", os.path.join(tsplib_dir, "berlin52.tsp")], capture_output=True, text=True)
    assert r.returncode == 2 and "unknown solver" in r.stderr
    tiny = "NAME: t\nTYPE: TSP\nDIMENSION: 2\nEDGE_WEIGHT_TYPE: EUC_2D\nNODE_COORD_SECTION\n1 0 0\n2 1 1\nEOF\n"
    r = subprocess.run([cli, "solve", "2opt", "--no-seed"], input=tiny, capture_output=True, text=True)
    assert r.returncode == 101 and "panicked" in r.stderr  # the reference panics on n < 3 (two_opt.rs:17,29)


def test_cli_pipeline_json_optimal_tour_and_shuffle(cli, tsplib_dir, goldens):
    """`pipeline --steps`, `--output-format json` (main.rs:700-712), `--optimal-tour` (main.rs:660-698, opt_tour.rs) and the
    seeded `shuffle` stage: whole stdout / stderr bytes."""
    import numpy as np
    import _oracle as O
    import _tsplib as T
    f = os.path.join(tsplib_dir, "berlin52.tsp")
    opt = os.path.join(tsplib_dir, "berlin52.opt.tour")
    g = goldens["berlin52"]
    # pipeline --steps=nn,2opt == solve 2opt == the `fast` preset
    a = subprocess.run([cli, "pipeline", "--steps=nn,2opt", "-i", f], capture_output=True, text=True)
    b = subprocess.run([cli, "solve", "fast", "-i", f], capture_output=True, text=True)
    c = subprocess.run([cli, "solve", "2opt", "-i", f], capture_output=True, text=True)
    want = "8384.18848 0\n" + "".join(f"{v} " for v in g["nn_two_opt"]["route_ids"]) + "\n"
    assert a.returncode == b.returncode == c.returncode == 0 and a.stdout == b.stdout == c.stdout == want
    # three stages, warm-started in turn; `--steps nn,2opt,or_opt` with a space also parses
    r = subprocess.run([cli, "pipeline", "--steps", "nn,2opt,or_opt", "-i", f], capture_output=True, text=True)
    e = T.parse_tsplib(f)
    rc, nn, _ = O.nearest_neighbor(e["xy"], None, 52, 3)
    rc, p2, _, _ = O.two_opt(e["xy"], None, 52, init=nn)
    rc, p3, c3, _ = O.or_opt(e["xy"], None, 52, init=p2)
    assert r.stdout == f"{float(c3):.5f} 0\n" + "".join(f"{v} " for v in e["ids"][p3]) + "\n"
    # JSON: keys sorted, compact, f32 widened to f64 and printed shortest (serde_json)
    r = subprocess.run([cli, "solve", "3opt", "-i", f, "--output-format", "json"], capture_output=True, text=True)
    c32 = np.float32(7742.64697)
    assert r.stdout == '{"cost":' + repr(float(c32)) + ',"optimized":false,"route":[' + ",".join(str(v) for v in g["nn_three_opt"]["route_ids"]) + "]}\n"
    assert json.loads(r.stdout)["cost"] == float(c32)
    # --optimal-tour: text mode prints the comparison to stderr, JSON mode adds optimal_cost / gap_pct
    r = subprocess.run([cli, "solve", "2opt", "-i", f, "--optimal-tour", opt], capture_output=True, text=True)
    assert r.stdout == want
    assert r.stderr.endswith("--- Comparison ---\nOptimal  : 7544.36572  (from BERLIN52.OPT.TOUR)\nSolver   : 8384.18848\nGap      : +11.13 %\n")
    r = subprocess.run([cli, "solve", "2opt", "-i", f, "--optimal-tour", opt, "--output-format=json"], capture_output=True, text=True)
    j = json.loads(r.stdout)
    gap = (np.float32(8384.18848) - np.float32(7544.36572)) / np.float32(7544.36572) * np.float32(100.0)
    assert list(j) == ["cost", "gap_pct", "optimal_cost", "optimized", "route"]
    assert j["optimal_cost"] == float(np.float32(7544.36572)) and j["gap_pct"] == float(np.float32(gap))
    r = subprocess.run([cli, "solve", "lk", "-i", f, "--seed", "1", "--optimal-tour", opt], capture_output=True, text=True)
    assert "Gap      : 0.00 % (matches optimal)" in r.stderr and r.stdout.startswith("7544.36572 0\n")
    # a broken .opt.tour is reported and ignored (main.rs:486-493)
    r = subprocess.run([cli, "solve", "nn", "-i", f, "--optimal-tour", f], capture_output=True, text=True)
    assert r.returncode == 0 and "--optimal-tour: opt_tour: expected TYPE : TOUR, found TYPE : TSP" in r.stderr
    # shuffle stage, seeded: the Fisher-Yates stream of restart 0 (oracle tlo_restart_perm); shuffle -> 2opt == oracle from it
    r = subprocess.run([cli, "solve", "shuffle", "-i", f, "--seed", "77"], capture_output=True, text=True)
    rp = O.restart_perm(52, 77, 0)
    assert r.stdout == f"{float(O.tour_length(e['xy'], None, rp)):.5f} 0\n" + "".join(f"{v} " for v in e["ids"][rp]) + "\n"
    r = subprocess.run([cli, "pipeline", "--steps=shuffle,2opt", "-i", f, "--seed", "77"], capture_output=True, text=True)
    rc, p, c, _ = O.two_opt(e["xy"], None, 52, init=rp)
    assert r.stdout == f"{float(c):.5f} 0\n" + "".join(f"{v} " for v in e["ids"][p]) + "\n"
    # --distance-type geo on a coordinate file (main.rs:461-471): burma14 read as EUC_2D vs GEO
    b14 = os.path.join(tsplib_dir, "burma14.tsp")
    r1 = subprocess.run([cli, "solve", "nn", "-i", b14, "--distance-type", "euc_2d"], capture_output=True, text=True)
    x = T.parse_tsplib(b14)
    rc, nn, cnn = O.nearest_neighbor(x["xy"], None, 14, 3)
    assert r1.stdout.startswith(f"{float(cnn):.5f} 0\n")
    r = subprocess.run([cli, "solvers", "--short"], capture_output=True, text=True)
    assert r.stdout.split() == ["nn", "2opt", "3opt", "or-opt", "lk", "shuffle"]


def test_python_facade_formats_and_comparison(ctx, tsplib_dir, goldens):
    import numpy as np
    import teeline_amd as TA
    prob = TA.tsplib.read_from_file(os.path.join(tsplib_dir, "berlin52.tsp")).problem()
    out = TA.pipeline.run_pipeline_stages(prob, TA.pipeline.steps_for_solve("2opt"), ctx=ctx)
    sol = out[-1].solution
    assert TA.pipeline.format_solution(sol) == "8384.18848 0\n" + "".join(f"{v} " for v in goldens["berlin52"]["nn_two_opt"]["route_ids"]) + "\n"
    ot = TA.opt_tour.read_from_file(os.path.join(tsplib_dir, "berlin52.opt.tour"))
    cmp = TA.opt_tour.compute_optimal_comparison(sol.total, prob, ot, ctx=ctx)
    assert f"{float(cmp[0]):.5f}" == "7544.36572" and f"{float(cmp[1]):+.2f}" == "+11.13" and cmp[2] == "BERLIN52.OPT.TOUR"
    j = json.loads(TA.pipeline.format_solution_json(sol, False, cmp))
    assert list(j) == ["cost", "gap_pct", "optimal_cost", "optimized", "route"] and j["cost"] == float(np.float32(8384.18848))
    sh = TA.pipeline.run_pipeline_stages(prob, ["shuffle", "2opt"], ctx=ctx, lk_seed=77)
    import _oracle as O
    rp = O.restart_perm(52, 77, 0)
    assert sh[0].solution.route() == prob.ids[rp].tolist()
    with pytest.raises(ValueError):
        TA.pipeline.steps_for_solve("sa")
    assert TA.pipeline.steps_for_solve("lk", no_seed=True) == ["lk"] and TA.pipeline.steps_for_solve("fast") == ["nn", "2opt"]


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


def _digest(messages):
    # the CLI's --progress-digest: FNV-1a 64 over (kind, id count u32, ids u64, f32 bits) per message; Done: the kind only
    import struct
    h, n = 1469598103934665603, [0, 0, 0]

    def feed(b):
        nonlocal h
        for x in b:
            h = ((h ^ x) * 1099511628211) & 0xFFFFFFFFFFFFFFFF

    for kind, payload in messages:
        k = {"PathUpdate": 0, "CityChange": 1, "Done": 2}[kind]
        n[k] += 1
        feed(bytes([k]))
        if k == 2:
            continue
        ids, v = (payload[0], payload[1]) if k == 0 else ([payload], 0.0)
        feed(struct.pack("<I", len(ids)) + b"".join(struct.pack("<Q", int(i)) for i in ids))
        feed(struct.pack("<f", v) if k == 0 else struct.pack("<I", 0))
    return n, h


def test_cpp_mirror_replays_the_same_progress_messages_as_the_python_mirror(cli, ctx, tsplib_dir):
    # The C++ mirror's progress replays (2-opt, 3-opt, Or-opt, LK, NN through the *_trace entries) against the Python mirror's, which the
    # per-solver tests compare with the reference's loops restated: same counts per kind, same digest over every id and f32 bit.
    import re
    import teeline_amd as TA
    f = os.path.join(tsplib_dir, "berlin52.tsp")
    prob = TA.tsplib.read_from_file(f).problem()
    for name, extra in (("nn", []), ("2opt", []), ("3opt", []), ("oropt", []), ("lk", ["--seed", "5", "--epochs", "40", "--platoo_epochs", "10", "--n_nearest", "5"])):
        r = subprocess.run([cli, "solve", name, "-i", f, "--progress-digest", *extra], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        m = re.search(r"progress: path_updates=(\d+) city_changes=(\d+) done=(\d+) digest=([0-9a-f]{16})", r.stderr)
        assert m, r.stderr
        got = []
        tx = lambda kind, payload: got.append((kind, payload))  # noqa: E731
        seed = TA.nearest_neighbor.solve(prob, TA.HeuristicOptions(n_nearest=3), tx, ctx=ctx)  # `solve X` = nn -> X (auto_expand_with_nn)
        if name == "2opt":
            TA.two_opt.solve(prob, None, tx, seed.route(), ctx=ctx)
        elif name == "3opt":
            TA.three_opt.solve(prob, None, tx, seed.route(), ctx=ctx)
        elif name == "oropt":
            TA.or_opt.solve(prob, None, tx, seed.route(), ctx=ctx)
        elif name == "lk":
            TA.lin_kernighan.solve(prob, TA.LKOptions(TA.HeuristicOptions(epochs=40, platoo_epochs=10, n_nearest=5), 5), tx, seed.route(), ctx=ctx, seed=5)
        n, h = _digest(got)
        assert [int(m.group(1)), int(m.group(2)), int(m.group(3))] == n, name
        assert m.group(4) == f"{h:016x}", name
