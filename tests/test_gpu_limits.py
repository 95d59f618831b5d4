"""Size limits and degenerate inputs of every entry point (-m gpu): the largest n each kernel form takes, one past it
(TL_ERR_UNSUPPORTED, never a HIP error or a wrong answer), and the empty / tiny inputs the reference special-cases.
Where the oracle would need minutes, size-independent properties stand in: valid tour, endpoints of the open path fixed,
fixed point of one more reference sweep, cost == tour_length, reported delta == cost difference."""
import os
import sys

import numpy as np
import pytest

import _oracle as O

pytestmark = pytest.mark.gpu


def len64(xy, tour):
    p = np.asarray(xy, dtype=np.float64)[np.asarray(tour, dtype=np.int64)]
    d = p - np.roll(p, -1, axis=0)
    return float(np.sqrt((d * d).sum(axis=1)).sum())


def prob(xy, packed=None):
    import teeline_amd as TA
    n = len(xy)
    dm = None if packed is None else TA.distance_matrix.DistanceMatrix(n, packed, np.arange(n), "explicit")
    return TA.TspProblem(np.arange(n), xy, dm)


def test_two_opt_size_limits(ctx):
    import teeline_amd as TA
    nmax = ctx.two_opt_lds_max_n()
    assert 15000 <= nmax <= 16384
    # the LDS kernel at its limit and the HBM-resident form one city later: both from the NN seed, checked as fixed points
    for n in (nmax, nmax + 1):
        xy = O.synth_xy(n, seed=11)
        rc, nn, cnn = O.nearest_neighbor(xy, None, n, 3)
        sol = TA.two_opt.solve(prob(xy), None, None, [int(v) for v in nn], ctx=ctx)
        route = np.asarray(sol.route(), dtype=np.uint32)
        assert O.validate_tour(route) and route[0] == nn[0] and route[-1] == nn[-1] and sol.total < cnn
        rc, again, c2, st2 = O.two_opt(xy, None, n, init=route, max_candidates=1)   # one reference sweep: nothing left to improve
        assert st2["moves"] == 0 and np.float32(c2).tobytes() == np.float32(sol.total).tobytes()
        assert sol.stats["candidates"] == sol.stats["sweeps"] * ((n - 3) * (n - 2) // 2)
    # (the HBM-resident form has had 64-bit (i, j) keys since round 4: no 65 535 limit any more — test_two_opt_beyond_65535_cities)
    # multi-start / population beyond the LDS-resident descent (round 5): the descents run one after the other through
    # the HBM form.  Population: two nearly optimal tours (the NN seed and the NN seed with a stretch reversed) against tl_two_opt on each;
    # multi-start: ONE seeded restart at the limit + 1 (a random start of 16 K cities is seconds of descent) — its start permutation is the
    # kernels' own stream (checked against the oracle's tlo_restart_perm through the result of the single descent from it).
    n = nmax + 1
    xy = O.synth_xy(n, seed=11)
    rc, nn, cnn = O.nearest_neighbor(xy, None, n, 3)
    t2 = nn.copy()
    t2[100:900] = t2[100:900][::-1].copy()
    sols = TA.two_opt.solve_population(prob(xy), [[int(v) for v in nn], [int(v) for v in t2]], ctx=ctx)
    for tour, sol in zip((nn, t2), sols):
        one = TA.two_opt.solve(prob(xy), None, None, [int(v) for v in tour], ctx=ctx)
        assert list(sol.route()) == list(one.route()) and np.float32(sol.total).tobytes() == np.float32(one.total).tobytes()
    ms, costs = TA.two_opt.multistart(prob(xy), 1, seed=7, first=3, ctx=ctx, return_costs=True)
    one = TA.two_opt.solve(prob(xy), None, None, [int(v) for v in O.restart_perm(n, 7, 3)], ctx=ctx)
    assert list(ms.route()) == list(one.route()) and np.float32(ms.total).tobytes() == np.float32(one.total).tobytes()
    assert ms.stats["best_restart"] == 3 and np.float32(costs[0]).tobytes() == np.float32(one.total).tobytes()


def test_two_opt_beyond_65535_cities(ctx):
    """the reference takes any n (`usize` indices, two_opt.rs:26-61); the HBM-resident form packed (i, j) into 32 bits
    and stopped at 65 535.  With the 64-bit key: a 257 x 257 lattice (n = 66 049, every distance exact in f32) walked as a snake — an
    optimal open path — with segments reversed near both ends of the tour, so that improving candidates sit at rows beyond 65 535 as
    well as below; tour (CRC-32), cost bits, sweeps, moves and reversal count against the oracle's committed result."""
    import json
    import zlib
    import teeline_amd as TA
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_goldens_limits as ML
    # the oracle needs ~30-60 s for this descent: its result is a committed golden (tests/golden/make_goldens_limits.py writes
    # goldens_limits.json from the oracle), so the GPU suite stays inside its time budget
    g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "goldens_limits.json")))["lattice257_two_opt"]
    xy, init = ML.lattice257()
    n = len(xy)
    assert n == g["n"] == 257 * 257 and ML.crc(init) == g["init_crc32"]
    sol = TA.two_opt.solve(prob(xy), None, None, [int(v) for v in init], ctx=ctx)
    st = g["stats"]
    assert st["moves"] >= 5
    assert ML.crc(np.asarray(sol.route(), dtype=np.uint32)) == g["route_crc32"]
    assert int(np.float32(sol.total).view(np.uint32)) == g["cost_bits"]
    assert (sol.stats["sweeps"], sol.stats["moves"], sol.stats["reversed"], sol.stats["candidates"]) == (st["sweeps"], st["moves"], st["reversed"], st["candidates"])


def test_matrix_form_size_limit(ctx):
    import teeline_amd as TA
    n = 19000  # 8 B of LDS per city: below the ~19 900 limit; 1.4 GB full matrix in the workspace
    xy = O.synth_xy(n, seed=3)
    packed = TA.distance_matrix.build(np.arange(n), xy, ctx=ctx).items
    rc, nn, cnn = O.nearest_neighbor(xy, None, n, 3)
    sol = TA.two_opt.solve(prob(xy, packed), None, None, [int(v) for v in nn], ctx=ctx)
    ref = TA.two_opt.solve(prob(xy), None, None, [int(v) for v in nn], ctx=ctx)      # the coordinate form is the cross-check
    assert sol.route() == ref.route() and sol.total == ref.total and sol.stats["moves"] == ref.stats["moves"]
    with pytest.raises(TA.TeelineGpuError) as e:
        big = 21000
        xyb = O.synth_xy(big, seed=4)
        TA.two_opt.solve(prob(xyb, TA.distance_matrix.build(np.arange(big), xyb, ctx=ctx).items), None, None, None, ctx=ctx)
    assert e.value.code == TA._capi.TL_ERR_UNSUPPORTED


def test_three_opt_and_or_opt_limits(ctx):
    import teeline_amd as TA
    # 3-opt: (i, j) and (k, case) travel as packed 16-bit fields: n <= 65 535; one past it is refused up front.  (Until round 4 the
    # pick kernel staged the move's segments in LDS and the limit was (160 KB - 2 KB) / 4 = 40 448; they go through the workspace
    # now from 257 cities on, which the full-descent parity tests at a280 / n = 300 run.)
    with pytest.raises(TA.TeelineGpuError) as e:
        TA.three_opt.find_best_move(prob(O.synth_xy(65536, seed=5)), np.arange(65536), ctx=ctx)
    assert e.value.code == TA._capi.TL_ERR_UNSUPPORTED
    # a large scan the oracle cannot afford: the reported savings must be the cost difference of the applied move
    n = 6000
    xy = O.synth_xy(n, seed=6)
    rc, nn, cnn = O.nearest_neighbor(xy, None, n, 3)
    mv = TA.three_opt.find_best_move(prob(xy), nn, ctx=ctx)
    assert mv is not None
    rc, after = O.apply_3opt(nn, *mv[:4])
    assert rc == 0 and O.validate_tour(after)
    assert abs(len64(xy, nn) - len64(xy, after) - float(mv[4])) < 0.01 and mv[4] > 0
    # Or-opt (round 4): the loop-order key 6 n^2 no longer has to fit 32 bits (n <= 26 754 before) and the pick kernel stages the tour
    # in the workspace where it does not fit the LDS (n > ~40 700).  Limit + 1 against the oracle: an instance whose best move has a
    # loop-order index beyond 2^32 — the NN tour of a random instance with a displaced TRIPLE of cities near the tour's end (the
    # index is ((seg_len - 1) n + i) n + j) * 2: only seg_len 3 rows with i near n pass 2^32 at this size) — then a size beyond
    # the LDS staging with an apply, checked through what the move must do.
    n = 26800
    xy = O.synth_xy(n, seed=8)
    sol_nn = TA.nearest_neighbor.solve(prob(xy), ctx=ctx)
    nn = np.asarray(sol_nn.route(), dtype=np.uint32)
    path = np.concatenate([nn[:n - 40], nn[100:103], nn[n - 40:]])
    path = np.concatenate([path[:100], path[103:]]).astype(np.uint32)  # cities nn[100..102] now sit 40 from the end
    assert O.validate_tour(path)
    mv = TA.or_opt.find_best_move(prob(xy), path, ctx=ctx)
    omv = O.or_opt_find_best_move(xy, None, path)
    assert mv is not None and omv is not None
    assert np.float32(mv[0]).tobytes() == np.float32(omv[0]).tobytes() and tuple(mv[1:]) == tuple(omv[1:])
    order = (((mv[3] - 1) * n + mv[1]) * n + mv[2]) * 2 + int(mv[4])
    assert order >= 2 ** 32, "the winning move's loop-order index needs the wide key"
    # beyond the LDS staging of k_or_pick (4 n bytes): a convex instance (cities on a circle, in order) with one displaced pair —
    # the best relocation puts it back, after which nothing improves: one move, two passes, the tour is the circle again
    n = 41000
    ang = (np.arange(n, dtype=np.float64) * (2.0 * np.pi / n))
    cxy = np.ascontiguousarray(np.stack([1e5 * np.cos(ang), 1e5 * np.sin(ang)], 1), dtype=np.float32)
    ident = np.arange(n, dtype=np.uint32)
    moved = np.concatenate([ident[:100], ident[102:40000], ident[100:102], ident[40000:]]).astype(np.uint32)
    sol = TA.or_opt.solve(prob(cxy), None, None, [int(v) for v in moved], ctx=ctx)
    assert list(sol.route()) == ident.tolist() and sol.stats["moves"] == 1 and sol.stats["sweeps"] == 2
    assert np.float32(sol.total).tobytes() == O.tour_length(cxy, None, ident).tobytes()


def test_candidate_lists_and_nn_seed_at_large_n(ctx):
    import teeline_amd as TA
    n = 150000
    xy = O.synth_xy(n, seed=9)
    got = TA.lin_kernighan.build_candidates(prob(xy), 8, ctx=ctx)
    want, _ = O.build_candidates_kdtree(xy, 8)
    assert np.array_equal(got, want)
    k16 = TA.lin_kernighan.build_candidates(prob(xy[:5000]), 16, ctx=ctx)
    assert np.array_equal(k16, O.build_candidates_kdtree(xy[:5000], 16)[0])
    # round 5: lists of up to 64 (the reference's n_nearest is an unbounded usize) — 17 / 33 / 64 against the oracle's
    # kd-tree, brute force the same, one past the limit refused with a message that names it
    for k in (17, 33, 64):
        got = TA.lin_kernighan.build_candidates(prob(xy[:3000]), k, ctx=ctx)
        assert np.array_equal(got, O.build_candidates_kdtree(xy[:3000], k)[0]), k
    with TA.Context(0, TA.TL_FLAG_KNN_BRUTE) as cb:
        assert np.array_equal(TA.lin_kernighan.build_candidates(prob(xy[:3000]), 40, ctx=cb), O.build_candidates_kdtree(xy[:3000], 40)[0])
    with pytest.raises(TA.TeelineGpuError) as e:
        TA.lin_kernighan.build_candidates(prob(xy[:100]), 65, ctx=ctx)
    assert e.value.code == TA._capi.TL_ERR_UNSUPPORTED and "unbounded" in str(e.value)
    # ... and through the solvers that read them: the NN seed with n_nearest = 40, LK with n_nearest = 20 (chip-wide scans: the LDS form takes k <= 16)
    sol = TA.nearest_neighbor.solve(prob(xy[:2000]), TA.HeuristicOptions(n_nearest=40), ctx=ctx)
    rc, route, c = O.nearest_neighbor(xy[:2000], None, 2000, 40)
    assert list(sol.route()) == route.tolist() and np.float32(sol.total).tobytes() == np.float32(c).tobytes()
    h20 = TA.HeuristicOptions(epochs=3, platoo_epochs=3, n_nearest=20)
    s20 = TA.lin_kernighan.solve(prob(xy[:300]), TA.LKOptions(h20, 3), None, None, ctx=ctx, seed=5)
    rc, r20, c20, st20 = O.lin_kernighan(xy[:300], seed=5, epochs=3, platoo_epochs=3, n_nearest=20, max_depth=3)
    assert list(s20.route()) == r20.tolist() and np.float32(s20.total).tobytes() == np.float32(c20).tobytes() and s20.stats["sweeps"] == st20["sweeps"]
    n2 = 40000
    sol = TA.nearest_neighbor.solve(prob(xy[:n2]), ctx=ctx)
    rc, route, c = O.nearest_neighbor(xy[:n2], None, n2, 3)
    assert list(sol.route()) == route.tolist() and np.float32(sol.total).tobytes() == np.float32(c).tobytes()
    with pytest.raises(TA.TeelineGpuError) as e:   # visited flags live in one CU's LDS
        TA.nearest_neighbor.solve(prob(O.synth_xy(170000, seed=10)), ctx=ctx)
    assert e.value.code == TA._capi.TL_ERR_UNSUPPORTED


def test_lk_at_a_size_beyond_the_oracle(ctx):
    # n = 40 000, one epoch: the oracle would need ~20 min; properties only
    import teeline_amd as TA
    n = 40000
    xy = O.synth_xy(n, seed=14)
    h = TA.HeuristicOptions(epochs=1, platoo_epochs=1, n_nearest=5)
    sol = TA.lin_kernighan.solve(prob(xy), TA.LKOptions(h, 5), None, None, ctx=ctx, seed=3)
    route = np.asarray(sol.route(), dtype=np.uint32)
    rc, nn, cnn = O.nearest_neighbor(xy, None, n, 3)
    assert O.validate_tour(route) and sol.total < cnn and sol.stats["moves"] > 0
    assert np.float32(sol.total).tobytes() == O.tour_length(xy, None, route).tobytes()
    # the result of an lk_pass is LK-optimal for its own candidate lists: one more pass of the oracle finds nothing ... from
    # the tour the LAST pass ended in, which is `route` only if the kick was rejected; so only when no kick improved
    with pytest.raises(TA.TeelineGpuError) as e:
        TA.lin_kernighan.solve(prob(xy[:100]), TA.LKOptions(h, 17), None, None, ctx=ctx)   # max_depth > 16 (the deep build's chain capacity)
    assert e.value.code == TA._capi.TL_ERR_UNSUPPORTED and "unbounded" in str(e.value)


def test_lk_max_depth_beyond_the_register_build(ctx):
    """the reference's max_depth is an unbounded usize (mod.rs:1252, recursion lin_kernighan.rs:265-340); up to 6
    the chain lives in registers (lk.hip), 7..16 run the same kernels built with larger chain arrays (lk_deep.hip).  Limit + 1 and
    beyond against the oracle on small instances with large options: every form of the scan (fused pick with parked walks at
    k (k+1)^2 <= 1024, the flat form beyond, the chip-wide step from n = 1500, the single-workgroup cross-check)."""
    import teeline_amd as TA
    from test_gpu_lk import assert_same, gpu_lk, lattice
    xy = O.synth_xy(300, seed=21)
    for depth, k, epochs in ((7, 5, 6), (10, 4, 4), (16, 2, 6), (9, 10, 1)):  # (the last one: k (k+1)^2 > 1024 lanes, the flat scan form)
        assert_same(gpu_lk(ctx, xy, seed=2, epochs=epochs, max_depth=depth, n_nearest=k),
                    O.lin_kernighan(xy, seed=2, epochs=epochs, max_depth=depth, n_nearest=k))
    lat = lattice(9, 3)  # ties everywhere: long chains of equal-gain exchanges
    assert_same(gpu_lk(ctx, lat, seed=5, epochs=10, max_depth=9, n_nearest=6), O.lin_kernighan(lat, seed=5, epochs=10, max_depth=9, n_nearest=6))
    big = O.synth_xy(1600, seed=22)  # chip-wide step (n >= 1500)
    assert_same(gpu_lk(ctx, big, seed=3, epochs=2, max_depth=7), O.lin_kernighan(big, seed=3, epochs=2, max_depth=7))
    with TA.Context(0, TA.TL_FLAG_LK_ONE_WORKGROUP) as c1:
        assert_same(gpu_lk(c1, xy, seed=2, epochs=3, max_depth=7), O.lin_kernighan(xy, seed=2, epochs=3, max_depth=7))


def test_empty_and_tiny_inputs_of_every_entry_point(ctx):
    import teeline_amd as TA
    one = np.array([[3.0, 4.0]], np.float32)
    two = np.array([[0, 0], [3, 4]], np.float32)
    three = np.array([[0, 0], [3, 4], [6, 0]], np.float32)
    # two_opt: n < 3 underflows in the reference (two_opt.rs:17,29) -> ReferencePanics; n = 3: identity, one empty sweep
    for xy in (one, two):
        with pytest.raises(TA.ReferencePanics):
            TA.two_opt.solve(prob(xy), None, None, None, ctx=ctx)
    s = TA.two_opt.solve(prob(three), None, None, [2, 0, 1], ctx=ctx)
    assert s.route() == [2, 0, 1] and s.stats["sweeps"] == 1 and s.stats["candidates"] == 0 and s.total == np.float32(16.0)
    # three_opt / or_opt: n < 4 returns the cities order whatever the seed (three_opt.rs:25-28, or_opt.rs:31-34)
    for mod in (TA.three_opt, TA.or_opt):
        for xy in (one, two, three):
            s = mod.solve(prob(xy), None, None, list(range(len(xy)))[::-1], ctx=ctx)
            assert s.route() == list(range(len(xy)))
        assert mod.find_best_move(prob(three), [0, 1, 2], ctx=ctx) is None
    # nearest_neighbor: one city is its own tour; lin_kernighan: fewer than 4 cities -> the seed untouched (lin_kernighan.rs:57-59)
    assert TA.nearest_neighbor.solve(prob(one), ctx=ctx).route() == [0]
    assert TA.nearest_neighbor.solve(prob(two), ctx=ctx).route() == [0, 1]
    for xy in (one, two, three):
        s = TA.lin_kernighan.solve(prob(xy), None, None, None, ctx=ctx)
        rc, r, c, st = O.lin_kernighan(xy)
        assert s.route() == r.tolist() and np.float32(s.total).tobytes() == np.float32(c).tobytes()
    # distance matrix: at least two points (distance_matrix.rs:124-126); tour_length of fewer than two cities is 0 (:236-238)
    with pytest.raises(ValueError):
        TA.distance_matrix.build([0], one, ctx=ctx)
    dm = TA.distance_matrix.build([0, 1], two, ctx=ctx)
    assert dm.items.tolist() == [5.0] and dm.tour_length([0], ctx=ctx) == 0.0 and dm.tour_length([0, 1], ctx=ctx) == np.float32(10.0)
    # candidate lists: k clamps to n - 1 (lin_kernighan.rs:14); a single city has none
    assert TA.lin_kernighan.build_candidates(prob(two), 5, ctx=ctx).tolist() == [[1], [0]]
    assert TA.lin_kernighan.build_candidates(prob(one), 5, ctx=ctx).shape == (1, 0)
    # multi-start: at least one restart; a population needs valid permutations
    with pytest.raises(TA.TeelineGpuError):
        TA.two_opt.multistart(prob(O.synth_xy(50, seed=1)), 0, ctx=ctx)
    with pytest.raises(TA.TeelineGpuError):
        TA.two_opt.solve_population(prob(O.synth_xy(5, seed=1)), [[0, 1, 2, 3, 3]], ctx=ctx)


def test_forced_kernel_forms_fall_back_beyond_their_sizes(ctx):
    # TL_FLAG_2OPT_FX / _NT512 / _NT256 force a form only where it exists: the grid form up to n = 10 240, the 8- / 4-wave
    # forms up to 15 x 512 / 15 x 256 cities (a flush holds 15 elements per thread).  Beyond, the library runs the wide
    # float2 form: same tours as the default context.
    import teeline_amd as TA
    n = 12000
    rng = np.random.default_rng(12)
    xy = (rng.integers(0, 1000000, (n, 2)).astype(np.float32) / np.float32(1000.0)).astype(np.float32)
    prob = TA.TspProblem(np.arange(n), xy)
    nn = TA.nearest_neighbor.solve(prob, ctx=ctx).route()
    ref = TA.two_opt.solve(prob, None, None, nn, ctx=ctx)
    for flag in (TA.TL_FLAG_2OPT_FX, TA.TL_FLAG_2OPT_NT512, TA.TL_FLAG_2OPT_NT256):
        with TA.Context(0, flag) as c2:
            sol = TA.two_opt.solve(prob, None, None, nn, ctx=c2)
            assert list(sol.route()) == list(ref.route()) and sol.total == ref.total
            assert sol.stats["moves"] == ref.stats["moves"] and sol.stats["sweeps"] == ref.stats["sweeps"]
