"""GPU parity tests for 3-opt (-m gpu): tl_three_opt / tl_three_opt_find_best_move through the C ABI vs the oracle
(reference: src/tsp/three_opt.rs).  Bit-exact: same move (i,j,k,case), same f32 savings, same tours and costs."""
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


def gpu_three_opt(ctx, xy, packed, n, init=None):
    import teeline_amd as TA
    sol = TA.three_opt.solve(problem(xy, packed, n), None, None, None if init is None else [int(v) for v in init], ctx=ctx)
    return np.asarray(sol.route(), dtype=np.uint32), sol.total, sol.stats


def assert_same(gpu, ora):
    route, cost, st = gpu
    rc, oroute, ocost, ost = ora
    assert rc == 0
    assert route.tolist() == oroute.tolist(), "tour differs from the oracle"
    assert np.float32(cost).tobytes() == np.float32(ocost).tobytes()
    assert st["moves"] == ost["moves"] and st["sweeps"] == ost["sweeps"] and st["candidates"] == ost["candidates"]


def test_reference_unit_cases(ctx):
    import teeline_amd as TA
    square = np.array([[0, 0], [1, 0], [1, 1], [0, 1]], np.float32)
    tsp5 = np.array([[0.0, 0.0], [0.0, 0.5], [0.0, 1.0], [1.0, 1.0], [1.0, 0.0]], np.float32)
    assert TA.three_opt.find_best_move(problem(square, None, 4), [0, 1, 2, 3], ctx=ctx) is None      # three_opt.rs:333-339
    mv = TA.three_opt.find_best_move(problem(tsp5, None, 5), [0, 2, 4, 1, 3], ctx=ctx)               # :341-362
    omv = O.three_opt_find_best_move(tsp5, None, [0, 2, 4, 1, 3])
    assert mv is not None and mv[:4] == omv[:4] and mv[4].tobytes() == omv[4].tobytes()
    for pts in (square, tsp5):                                                                       # :416-457
        route, cost, st = gpu_three_opt(ctx, pts, None, len(pts))
        assert abs(cost - 4.0) < 1e-4 and O.validate_tour(route)
        assert_same((route, cost, st), O.three_opt(pts, None, len(pts)))
    route, cost, st = gpu_three_opt(ctx, tsp5, None, 5, init=[0, 1, 2, 3, 4])                       # :266-277
    assert route.tolist() == [0, 1, 2, 3, 4]
    for n in (2, 3):                                                                                 # :25-28 n < 4
        route, cost, st = gpu_three_opt(ctx, tsp5[:n], None, n, init=list(range(n))[::-1])
        assert route.tolist() == list(range(n))
        assert cost == O.tour_length(tsp5[:n], None, np.arange(n))


@pytest.mark.parametrize("name", ["berlin52", "att48"])
def test_tsplib_matches_goldens(ctx, name, tsplib_dir, goldens):
    d = T.parse_tsplib(os.path.join(tsplib_dir, f"{name}.tsp"))
    xy, n, ids = d["xy"], d["n"], d["ids"]
    rc, nn, _ = O.nearest_neighbor(xy, None, n, 3)
    for key, init in (("nn_three_opt", nn), ("identity_three_opt", None)):
        g = gpu_three_opt(ctx, xy, None, n, init)
        assert f5(g[1]) == goldens[name][key]["cost"]
        assert ids[g[0]].tolist() == goldens[name][key]["route_ids"]
        assert g[2]["moves"] == goldens[name][key]["stats"]["moves"]
        assert g[2]["candidates"] == goldens[name][key]["stats"]["candidates"]
    if name == "berlin52":
        assert f5(gpu_three_opt(ctx, xy, None, n, nn)[1]) == "7742.64697"  # docs/benchmarks.md:29 publishes 7742.65


@pytest.mark.parametrize("name", ["gr17", "ring6_explicit", "bays29"])
def test_explicit_matrix(ctx, name, tsplib_dir, goldens):
    d = T.parse_tsplib(os.path.join(tsplib_dir, f"{name}.tsp"))
    g = gpu_three_opt(ctx, None, d["packed"], d["n"])
    assert_same(g, O.three_opt(None, d["packed"], d["n"]))
    assert g[0].tolist() == goldens[name]["identity_three_opt"]["route_pos"]


@pytest.mark.parametrize("n,seed", [(4, 1), (5, 2), (6, 3), (7, 4), (17, 5), (64, 6), (65, 7), (120, 8)])
def test_full_solve_small_sizes(ctx, n, seed):
    xy = O.synth_xy(n, seed=seed)
    assert_same(gpu_three_opt(ctx, xy, None, n), O.three_opt(xy, None, n))
    rp = O.restart_perm(n, 9, seed)
    assert_same(gpu_three_opt(ctx, xy, None, n, rp), O.three_opt(xy, None, n, init=rp))


def test_ties_on_a_lattice(ctx):
    # many exactly equal savings: the reference keeps the FIRST (i,j,k) in loop order and the lowest case per triple
    g = np.stack(np.meshgrid(np.arange(7, dtype=np.float32), np.arange(7, dtype=np.float32)), -1).reshape(-1, 2)
    rng = np.random.default_rng(3)
    pts = np.ascontiguousarray(g[rng.permutation(len(g))])
    import teeline_amd as TA
    mv = TA.three_opt.find_best_move(problem(pts, None, len(pts)), np.arange(len(pts)), ctx=ctx)
    omv = O.three_opt_find_best_move(pts, None, np.arange(len(pts)))
    assert mv[:4] == omv[:4] and mv[4].tobytes() == omv[4].tobytes()
    assert_same(gpu_three_opt(ctx, pts, None, len(pts)), O.three_opt(pts, None, len(pts)))


@pytest.mark.parametrize("n", [300, 1002])
def test_find_best_move_large(ctx, n):
    # one full O(n^3) scan (1.67e8 triples at n = 1002) against the oracle: same move, same f32 savings
    import teeline_amd as TA
    xy = O.synth_xy(n)
    rc, nn, _ = O.nearest_neighbor(xy, None, n, 3)
    for path in (nn, O.restart_perm(n, 4, 0)):
        mv = TA.three_opt.find_best_move(problem(xy, None, n), path, ctx=ctx)
        omv = O.three_opt_find_best_move(xy, None, path)
        assert mv is not None and mv[:4] == omv[:4] and mv[4].tobytes() == omv[4].tobytes()
    # matrix form gives the same move
    packed = O.dm_build_packed(xy)
    mv2 = TA.three_opt.find_best_move(problem(None, packed, n), nn, ctx=ctx)
    omv_nn = O.three_opt_find_best_move(xy, None, nn)
    assert mv2[:4] == omv_nn[:4] and mv2[4].tobytes() == omv_nn[4].tobytes()


@pytest.fixture(scope="module")
def goldens3(golden_dir):
    with open(os.path.join(golden_dir, "goldens_three_opt.json")) as fh:
        return json.load(fh)


def _crc(p):
    p = np.asarray(p, dtype=np.uint32)
    return int(np.bitwise_xor.reduce(p * np.arange(1, len(p) + 1, dtype=np.uint32)))


def _check_descent(g, gold):
    route, cost, st = g
    assert int(np.float32(cost).view(np.uint32)) == gold["cost_bits"], (float(cost), gold["cost"])
    assert _crc(route) == gold["route_crc"] and route[:12].tolist() == gold["route_head"]
    assert st["moves"] == gold["stats"]["moves"] and st["sweeps"] == gold["stats"]["sweeps"]
    assert st["candidates"] == gold["stats"]["candidates"]


def test_a280_full_descent_from_the_nn_seed(ctx, tsplib_dir, goldens3):
    # full 3-opt descents were oracle-checked only up to n = 120.  a280 from the NN seed: 32 moves, among them
    # apply_3opt's segment-swap cases 4-7 (three_opt.rs:186-218) over long segments; coordinates and matrix form.
    d = T.parse_tsplib(os.path.join(tsplib_dir, "a280.tsp"))
    xy, n = d["xy"], d["n"]
    rc, nn, _ = O.nearest_neighbor(xy, None, n, 3)
    _check_descent(gpu_three_opt(ctx, xy, None, n, nn), goldens3["a280_nn_three_opt"])
    _check_descent(gpu_three_opt(ctx, None, O.dm_build_packed(xy), n, nn), goldens3["a280_nn_three_opt_matrix"])


def test_synthetic_300_full_descent_from_a_random_permutation(ctx, goldens3):
    # 225 moves from a seeded random permutation (1.0e9 triples): every pass moves a long segment
    n = 300
    xy = O.synth_xy(n)
    rp = O.restart_perm(n, 4, 0)
    assert rp[:8].tolist() == goldens3["synth300_perm_seed4_three_opt"]["perm_head"]
    _check_descent(gpu_three_opt(ctx, xy, None, n, rp), goldens3["synth300_perm_seed4_three_opt"])
    _check_descent(gpu_three_opt(ctx, None, O.dm_build_packed(xy), n, rp), goldens3["synth300_perm_seed4_three_opt_matrix"])


def _three_opt_messages(xy, packed, ids, init):
    # three_opt.rs:30-49 restated as its message stream, with the oracle's find_best_move / apply_3opt
    n = len(ids)
    path = np.arange(n, dtype=np.uint32) if init is None else np.asarray(init, dtype=np.uint32).copy()
    msgs = [("PathUpdate", ([int(ids[v]) for v in path], 0.0))]
    moves = []
    while True:
        mv = O.three_opt_find_best_move(xy, packed, path)
        if mv is None:
            break
        i, j, k, case, _ = mv
        rc, path = O.apply_3opt(path, i, j, k, case)
        moves.append((i, j, k, case))
        msgs.append(("PathUpdate", ([int(ids[v]) for v in path], 0.0)))
    msgs.append(("Done", None))
    return msgs, moves, path


def test_progress_channel_replays_the_reference_messages(ctx, tsplib_dir):
    # three_opt.rs:34,42,47-49: PathUpdate(path, 0.0) for the start path and after every apply_3opt, then Done.  With a progress
    # callback three_opt::solve goes through tl_three_opt_trace (moves i, j, k, case in order) and replays exactly that.
    import ctypes as C
    import teeline_amd as TA
    from teeline_amd import _capi
    cases = []
    d = T.parse_tsplib(os.path.join(tsplib_dir, "berlin52.tsp"))
    cases.append((d["xy"], None, d["ids"], None))
    cases.append((d["xy"], None, d["ids"], O.restart_perm(d["n"], 11, 0)))
    g = T.parse_tsplib(os.path.join(tsplib_dir, "gr17.tsp"))
    cases.append((g["xy"], g["packed"], g["ids"], None))
    xs = O.synth_xy(90, seed=5)
    cases.append((xs, None, np.arange(90), O.restart_perm(90, 3, 1)))
    for xy, packed, ids, init in cases:
        n = len(ids)
        want, moves, path = _three_opt_messages(xy, packed, ids, init)
        got = []
        dmx = None if packed is None else TA.distance_matrix.DistanceMatrix(n, np.ascontiguousarray(packed, dtype=np.float32), ids, "explicit")
        prob = TA.TspProblem(ids, xy, dmx)
        sol = TA.three_opt.solve(prob, None, lambda kind, payload: got.append((kind, payload)),
                                 None if init is None else [int(ids[v]) for v in init], ctx=ctx)
        assert got == want
        assert list(sol.route()) == [int(ids[v]) for v in path] and sol.stats["moves"] == len(moves)
        # the raw list, and a short buffer: the prefix + the full count
        out = np.empty(n, dtype=np.uint32)
        c, st, ln = C.c_float(), _capi.TlStats(), C.c_uint32()
        cap = max(len(moves) // 2, 1)
        log = np.full((cap, 4), 0x12345678, dtype=np.uint32)
        xyc = np.ascontiguousarray(xy, dtype=np.float32)
        pk = None if packed is None else np.ascontiguousarray(packed, dtype=np.float32)
        ip = None if init is None else np.ascontiguousarray(init, dtype=np.uint32)
        ctx.check(ctx.lib.tl_three_opt_trace(ctx.handle, xyc.ctypes.data_as(C.c_void_p), n, None if pk is None else pk.ctypes.data_as(C.c_void_p),
                                             None if ip is None else ip.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), C.byref(c), C.byref(st),
                                             log.ctypes.data_as(C.c_void_p), cap, C.byref(ln)))
        assert ln.value == len(moves) == st.moves
        m = min(cap, len(moves))
        assert log[:m].tolist() == [list(v) for v in moves[:m]]
        assert (log[m:] == 0x12345678).all() and out.tolist() == path.tolist()


def test_no_messages_below_four_cities(ctx):
    # three_opt.rs:25-28 returns before its first message
    import teeline_amd as TA
    xy = O.synth_xy(3, seed=2)
    got = []
    sol = TA.three_opt.solve(TA.TspProblem(np.arange(3), xy), None, lambda k, p: got.append((k, p)), None, ctx=ctx)
    assert got == [] and list(sol.route()) == [0, 1, 2]


def test_an_instance_on_which_the_reference_does_not_terminate(ctx):
    # three_opt.rs:61,121 accepts every move with savings > 0.0 in f32; on some lattices a cycle of neutral moves rounds to positive
    # savings each time and `while improved` never ends (found by tests/probes/fuzz_campaign_trace.py, seed 39: a 117-city lattice read
    # through its matrix).  The oracle — the same loop — is still moving after 2 000 moves (and after 64 n + 1024); the
    # library gives up after 64 n + 1024 passes with TL_ERR_NO_CONVERGE instead of hanging its caller.
    import teeline_amd as TA
    rng = np.random.default_rng(39)
    n = int(rng.integers(4, 120))
    xy = np.ascontiguousarray(rng.integers(0, int(rng.integers(2, 30)), (n, 2)), dtype=np.float32)
    rng.integers(0, 50)
    init = O.restart_perm(n, 39, 0)
    packed = O.dm_build_packed(xy)
    still_moving = 2000  # (descents of this size take ~100 moves; the full 64 n + 1024 = 8 512 in the oracle would take 20 s here)
    assert O.three_opt(xy, packed, n, init=init, max_moves=still_moving)[3]["moves"] >= still_moving
    with pytest.raises(TA._capi.TeelineGpuError) as e:
        TA.three_opt.solve(problem(xy, packed, n), None, None, [int(v) for v in init], ctx=ctx)
    assert e.value.code == TA._capi.TL_ERR_NO_CONVERGE
