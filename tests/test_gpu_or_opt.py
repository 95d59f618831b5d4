"""GPU parity tests for Or-opt (-m gpu): tl_or_opt / tl_or_opt_find_best_move through the C ABI vs the oracle
(reference: src/tsp/or_opt.rs).  Same move (delta bits, i, j, seg_len, reversed), same tours, costs and counters."""
import json
import os

import numpy as np
import pytest

import _oracle as O
import _tsplib as T

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def goldens(golden_dir):
    with open(os.path.join(golden_dir, "goldens.json")) as fh:
        return json.load(fh)


def f5(x):
    return f"{float(x):.5f}"


def problem(xy, packed, n):
    import teeline_amd as TA
    return TA.TspProblem(np.arange(n), xy if xy is not None else np.zeros((n, 2), np.float32),
                         None if packed is None else TA.distance_matrix.DistanceMatrix(n, packed, np.arange(n), "explicit"))


def gpu_or_opt(ctx, xy, packed, n, init=None):
    import teeline_amd as TA
    sol = TA.or_opt.solve(problem(xy, packed, n), None, None, None if init is None else [int(v) for v in init], ctx=ctx)
    return np.asarray(sol.route(), dtype=np.uint32), sol.total, sol.stats


def assert_same(g, o):
    route, cost, st = g
    rc, oroute, ocost, ost = o
    assert rc == 0 and route.tolist() == oroute.tolist(), "tour differs from the oracle"
    assert np.float32(cost).tobytes() == np.float32(ocost).tobytes()
    assert (st["sweeps"], st["moves"], st["candidates"]) == (ost["sweeps"], ost["moves"], ost["candidates"])


def test_reference_unit_cases(ctx):
    import teeline_amd as TA
    detour = np.array([[0, 0], [1, 0], [5, 5], [2, 0], [3, 0]], np.float32)
    square = np.array([[0, 0], [1, 0], [1, 1], [0, 1]], np.float32)
    mv = TA.or_opt.find_best_move(problem(detour, None, 5), np.arange(5), ctx=ctx)
    omv = O.or_opt_find_best_move(detour, None, np.arange(5))
    assert mv is not None and mv[1:] == omv[1:] and mv[0].tobytes() == omv[0].tobytes() and mv[0] < 0
    assert TA.or_opt.find_best_move(problem(square, None, 4), [0, 1, 2, 3], ctx=ctx) is None
    assert_same(gpu_or_opt(ctx, detour, None, 5), O.or_opt(detour, None, 5))
    assert abs(gpu_or_opt(ctx, square, None, 4, [0, 1, 2, 3])[1] - 4.0) < 1e-2
    assert gpu_or_opt(ctx, square[:3], None, 3)[0].tolist() == [0, 1, 2]
    six = np.array([[0, 0], [1, 0], [5, 5], [2, 0], [3, 0], [4, 1]], np.float32)
    assert_same(gpu_or_opt(ctx, six, None, 6), O.or_opt(six, None, 6))


@pytest.mark.parametrize("name", ["berlin52", "att48", "a280"])
def test_tsplib_matches_goldens(ctx, name, tsplib_dir, goldens):
    d = T.parse_tsplib(os.path.join(tsplib_dir, f"{name}.tsp"))
    xy, n, ids = d["xy"], d["n"], d["ids"]
    rc, nn, _ = O.nearest_neighbor(xy, None, n, 3)
    for key, init in (("nn_or_opt", nn), ("identity_or_opt", None)):
        g = gpu_or_opt(ctx, xy, None, n, init)
        assert f5(g[1]) == goldens[name][key]["cost"] and ids[g[0]].tolist() == goldens[name][key]["route_ids"]
        assert g[2]["moves"] == goldens[name][key]["stats"]["moves"] and g[2]["candidates"] == goldens[name][key]["stats"]["candidates"]
    if name == "berlin52":
        assert f5(gpu_or_opt(ctx, xy, None, n, nn)[1]) == "8097.47607"  # docs/benchmarks.md:48 publishes 8 097.48


@pytest.mark.parametrize("name", ["gr17", "ring6_explicit", "bays29"])
def test_explicit_matrix(ctx, name, tsplib_dir, goldens):
    d = T.parse_tsplib(os.path.join(tsplib_dir, f"{name}.tsp"))
    g = gpu_or_opt(ctx, None, d["packed"], d["n"])
    assert_same(g, O.or_opt(None, d["packed"], d["n"]))
    assert g[0].tolist() == goldens[name]["identity_or_opt"]["route_pos"]


@pytest.mark.parametrize("n,seed", [(4, 1), (5, 2), (6, 3), (7, 4), (64, 5), (65, 6), (200, 7), (600, 8)])
def test_full_solve_sizes(ctx, n, seed):
    xy = O.synth_xy(n, seed=seed)
    assert_same(gpu_or_opt(ctx, xy, None, n), O.or_opt(xy, None, n))
    rp = O.restart_perm(n, 3, seed)
    assert_same(gpu_or_opt(ctx, xy, None, n, rp), O.or_opt(xy, None, n, init=rp))


def test_ties_and_large_scan(ctx):
    import teeline_amd as TA
    g = np.stack(np.meshgrid(np.arange(9, dtype=np.float32), np.arange(9, dtype=np.float32)), -1).reshape(-1, 2)
    pts = np.ascontiguousarray(g[np.random.default_rng(5).permutation(len(g))])
    assert_same(gpu_or_opt(ctx, pts, None, len(pts)), O.or_opt(pts, None, len(pts)))
    n = 5000
    xy = O.synth_xy(n)
    rc, nn, _ = O.nearest_neighbor(xy, None, n, 3)
    mv = TA.or_opt.find_best_move(problem(xy, None, n), nn, ctx=ctx)
    omv = O.or_opt_find_best_move(xy, None, nn)
    assert mv[1:] == omv[1:] and mv[0].tobytes() == omv[0].tobytes()
