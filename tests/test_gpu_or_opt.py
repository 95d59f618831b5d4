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


def _or_opt_messages(xy, packed, ids, init):
    # or_opt.rs:36-72 restated as its message stream, with the oracle's find_best_move / apply_relocation / tour_length
    n = len(ids)
    tour = np.arange(n, dtype=np.uint32) if init is None else np.asarray(init, dtype=np.uint32).copy()
    msgs = [("PathUpdate", ([int(ids[v]) for v in tour], 0.0))]
    moves = []
    while True:
        mv = O.or_opt_find_best_move(xy, packed, tour)
        if mv is None:
            break
        _, i, j, seg, rev = mv
        rc, tour = O.apply_relocation(tour, i, seg, j, rev)
        moves.append((i, j, seg, int(rev)))
        msgs.append(("PathUpdate", ([int(ids[v]) for v in tour], float(O.tour_length(xy, packed, tour)))))
    msgs.append(("Done", None))
    return msgs, moves, tour


def test_progress_channel_replays_the_reference_messages(ctx, tsplib_dir):
    # or_opt.rs:40-42,62-67,70-72: PathUpdate(start, 0.0), PathUpdate(path, distances.tour_length(path)) after every
    # apply_relocation, Done.  With a progress callback or_opt::solve goes through tl_or_opt_trace and replays exactly that.
    import ctypes as C
    import teeline_amd as TA
    from teeline_amd import _capi
    d = T.parse_tsplib(os.path.join(tsplib_dir, "berlin52.tsp"))
    g = T.parse_tsplib(os.path.join(tsplib_dir, "gr17.tsp"))
    xs = O.synth_xy(150, seed=5)
    cases = [(d["xy"], None, d["ids"], None), (d["xy"], None, d["ids"], O.restart_perm(d["n"], 11, 0)), (g["xy"], g["packed"], g["ids"], None),
             (xs, None, np.arange(150), O.restart_perm(150, 3, 1))]
    for xy, packed, ids, init in cases:
        n = len(ids)
        want, moves, tour = _or_opt_messages(xy, packed, ids, init)
        got = []
        dmx = None if packed is None else TA.distance_matrix.DistanceMatrix(n, np.ascontiguousarray(packed, dtype=np.float32), ids, "explicit")
        sol = TA.or_opt.solve(TA.TspProblem(ids, xy, dmx), None, lambda kind, payload: got.append((kind, payload)),
                              None if init is None else [int(ids[v]) for v in init], ctx=ctx)
        assert len(got) == len(want) and [m[0] for m in got] == [m[0] for m in want]
        for a, b in zip(got, want):
            if a[0] == "PathUpdate":
                assert a[1][0] == b[1][0] and np.float32(a[1][1]).tobytes() == np.float32(b[1][1]).tobytes()
        assert list(sol.route()) == [int(ids[v]) for v in tour] and sol.stats["moves"] == len(moves)
        out = np.empty(n, dtype=np.uint32)
        c, st, ln = C.c_float(), _capi.TlStats(), C.c_uint32()
        cap = max(len(moves) // 2, 1)
        log = np.full((cap, 4), 0x12345678, dtype=np.uint32)
        xyc = np.ascontiguousarray(xy, dtype=np.float32)
        pk = None if packed is None else np.ascontiguousarray(packed, dtype=np.float32)
        ip = None if init is None else np.ascontiguousarray(init, dtype=np.uint32)
        ctx.check(ctx.lib.tl_or_opt_trace(ctx.handle, xyc.ctypes.data_as(C.c_void_p), n, None if pk is None else pk.ctypes.data_as(C.c_void_p),
                                          None if ip is None else ip.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), C.byref(c), C.byref(st),
                                          log.ctypes.data_as(C.c_void_p), cap, C.byref(ln)))
        assert ln.value == len(moves) == st.moves
        m = min(cap, len(moves))
        assert log[:m].tolist() == [list(v) for v in moves[:m]] and (log[m:] == 0x12345678).all() and out.tolist() == tour.tolist()


def test_nearest_neighbor_progress_follows_from_the_walk(ctx, tsplib_dir):
    # nearest_neighbor.rs:32-34,40-42,67-69,72-74: PathUpdate([start], 0.0); per step CityChange(current) + PathUpdate(path so far, 0.0); Done
    import teeline_amd as TA
    d = T.parse_tsplib(os.path.join(tsplib_dir, "berlin52.tsp"))
    got = []
    sol = TA.nearest_neighbor.solve(TA.TspProblem(d["ids"], d["xy"]), TA.HeuristicOptions(n_nearest=3), lambda k, p: got.append((k, p)), ctx=ctx)
    route = [int(v) for v in sol.route()]
    rc, oroute, oc = O.nearest_neighbor(d["xy"], None, d["n"], 3)
    assert route == [int(d["ids"][v]) for v in oroute]
    want = [("PathUpdate", (route[:1], 0.0))]
    for t in range(1, len(route)):
        want += [("CityChange", route[t - 1]), ("PathUpdate", (route[:t + 1], 0.0))]
    want.append(("Done", None))
    assert got == want
