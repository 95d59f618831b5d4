"""GPU parity tests for the LK row (-m gpu): candidate lists, NN seed and lin_kernighan::solve through the C ABI
against the oracle (reference: src/tsp/lin_kernighan.rs, nearest_neighbor.rs).  Kicks are seeded (same splitmix64
stream in both), so tours are bit-identical, not just costs."""
import os

import numpy as np
import pytest

import _oracle as O
import _tsplib as T

pytestmark = pytest.mark.gpu


def f5(x):
    return f"{float(x):.5f}"


def prob(xy):
    import teeline_amd as TA
    return TA.TspProblem(np.arange(len(xy)), xy)


def lattice(m, seed):
    g = np.stack(np.meshgrid(np.arange(m, dtype=np.float32), np.arange(m, dtype=np.float32)), -1).reshape(-1, 2)
    return np.ascontiguousarray(g[np.random.default_rng(seed).permutation(len(g))])


@pytest.mark.parametrize("k", [1, 3, 5, 10])
def test_candidate_lists(ctx, k, tsplib_dir):
    import teeline_amd as TA
    sets = [T.parse_tsplib(os.path.join(tsplib_dir, "berlin52.tsp"))["xy"], O.synth_xy(1000, seed=3), lattice(9, 1),
            O.synth_xy(4, seed=2)]
    sets.append(T.parse_tsplib(os.path.join(tsplib_dir, "a280.tsp"))["xy"])       # lattice: most rows hold distance ties
    sets.append(np.concatenate([lattice(6, 2), lattice(6, 2)[:9]]).astype(np.float32))  # duplicates
    sets.append(np.array([[0, 0], [0, 0], [0, 0]], np.float32))                     # test_kdtree_duplicate_coordinates
    for xy in sets:
        # the reference's lists: kd-tree k-NN buffer per city (kdtree.rs:193-212), equal distances in visiting order
        got = TA.lin_kernighan.build_candidates(prob(xy), k, ctx=ctx)
        want, _ = O.build_candidates_kdtree(xy, k)
        assert got.shape == want.shape and np.array_equal(got, want)


def _on_tune_build():
    import teeline_amd as TA
    return TA._capi.load().tl_version().decode().endswith("+tune")


TUNE_ONLY = pytest.mark.skipif("not _on_tune_build()", reason="rejected kernel forms: only libteeline_gpu_tune.so carries them "
                                                              "(test_rejected_forms_on_the_tune_build runs these in a child process)")


@pytest.mark.parametrize("flag", ["TL_FLAG_KNN_BRUTE", pytest.param("TL_FLAG_KNN_4LANES", marks=TUNE_ONLY), pytest.param("TL_FLAG_KNN_1LANE", marks=TUNE_ONLY)])
def test_candidate_list_builders_agree(flag):
    # the brute-force builders (TL_FLAG_KNN_BRUTE: sixteen lanes per city; _4LANES; _1LANE) scan in position order: the
    # oracle's scan lists (ascending f32 distance, ties -> lowest position), duplicates included — and the kd-tree walk's
    # lists wherever no distances tie
    import teeline_amd as TA
    dup = np.concatenate([lattice(7, 3), lattice(7, 3)[:20]]).astype(np.float32)
    with TA.Context(0, getattr(TA, flag)) as c2:
        for xy in (O.synth_xy(777, seed=8), dup, O.synth_xy(5, seed=1)):
            for k in (1, 4, 7, 16):
                got = TA.lin_kernighan.build_candidates(prob(xy), k, ctx=c2)
                assert np.array_equal(got, O.build_candidates(xy, k))
        rnd = O.synth_xy(3000, seed=4)
        assert np.array_equal(TA.lin_kernighan.build_candidates(prob(rnd), 5, ctx=c2), O.build_candidates_kdtree(rnd, 5)[0])


def test_nearest_neighbor_seed(ctx, tsplib_dir):
    import teeline_amd as TA
    want = {"berlin52": "8980.91797", "att532": "112099.42188", "a280": "3148.10962"}  # bench/baseline-solvers.tsv:2-16
    for name, cost in want.items():
        d = T.parse_tsplib(os.path.join(tsplib_dir, f"{name}.tsp"))
        sol = TA.nearest_neighbor.solve(TA.TspProblem(d["ids"], d["xy"]), ctx=ctx)
        rc, route, c = O.nearest_neighbor(d["xy"], None, d["n"], 3)
        assert f5(sol.total) == cost == f5(c)
        assert list(sol.route()) == d["ids"][route].tolist()
    # sizes chosen to hit every LDS placement of the walk's state: lists + coordinates (10^4), lists only (13 509),
    # neither (20 000, k = 5), coordinates only (17 000, k = 16)
    for xy, k in ((O.synth_xy(10000), 3), (lattice(12, 4), 3), (O.synth_xy(300, seed=9), 1), (O.synth_xy(300, seed=9), 7),
                  (O.synth_xy(13509), 3), (O.synth_xy(20000, seed=2), 5), (O.synth_xy(17000, seed=4), 16)):
        sol = TA.nearest_neighbor.solve(prob(xy), TA.HeuristicOptions(n_nearest=k), ctx=ctx)
        rc, route, c = O.nearest_neighbor(xy, None, len(xy), k)
        assert list(sol.route()) == route.tolist() and np.float32(sol.total).tobytes() == np.float32(c).tobytes()


def gpu_lk(ctx, xy, init=None, seed=1, **kw):
    import teeline_amd as TA
    h = TA.HeuristicOptions(epochs=kw.get("epochs", 100), platoo_epochs=kw.get("platoo_epochs", 10), n_nearest=kw.get("n_nearest", 5))
    sol = TA.lin_kernighan.solve(prob(xy), TA.LKOptions(h, kw.get("max_depth", 5)), None,
                                 None if init is None else [int(v) for v in init], ctx=ctx, seed=seed)
    return np.asarray(sol.route(), dtype=np.uint32), sol.total, sol.stats


def assert_same(g, o):
    route, cost, st = g
    rc, oroute, ocost, ost = o
    assert rc == 0 and route.tolist() == oroute.tolist(), "tour differs from the oracle"
    assert np.float32(cost).tobytes() == np.float32(ocost).tobytes()
    assert (st["sweeps"], st["candidates"], st["moves"], st["reversed"]) == (ost["sweeps"], ost["candidates"], ost["moves"], ost["reversed"])


def test_lk_structural_cases(ctx):
    # lin_kernighan.rs:519-926: crossed square improves at depth 1; tiny inputs are returned as they are
    sq = np.array([[0, 0], [1, 1], [1, 0], [0, 1]], np.float32)  # visiting order 0,1,2,3 crosses
    g = gpu_lk(ctx, sq, init=[0, 1, 2, 3], max_depth=1, epochs=0)
    assert_same(g, O.lin_kernighan(sq, init=[0, 1, 2, 3], epochs=0, max_depth=1))
    assert abs(g[1] - 4.0) < 1e-5
    tri = np.array([[0, 0], [1, 0], [0.5, 1]], np.float32)
    g = gpu_lk(ctx, tri, init=[2, 0, 1])
    assert g[0].tolist() == [2, 0, 1]  # :57-59 len < 4


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5])
def test_lk_berlin52_default_options(ctx, seed, tsplib_dir):
    xy = T.parse_tsplib(os.path.join(tsplib_dir, "berlin52.tsp"))["xy"]
    assert_same(gpu_lk(ctx, xy, seed=seed), O.lin_kernighan(xy, seed=seed))


def test_lk_berlin52_cli_options_reach_published_optimum(ctx, tsplib_dir):
    # CLI defaults (mod.rs:596-613,1321-1325): epochs 10000, platoo 500, n_nearest 3, depth 5; bench/baseline-solvers.tsv:17-21
    xy = T.parse_tsplib(os.path.join(tsplib_dir, "berlin52.tsp"))["xy"]
    kw = dict(epochs=10000, platoo_epochs=500, n_nearest=3)
    g = gpu_lk(ctx, xy, seed=1, **kw)
    assert_same(g, O.lin_kernighan(xy, seed=1, **kw))
    assert f5(g[1]) == "7544.36572"


@pytest.mark.parametrize("case", ["a280", "synth1000", "lattice", "random_start", "depth_limits"])
def test_lk_other_instances(ctx, case, tsplib_dir):
    if case == "a280":
        xy = T.parse_tsplib(os.path.join(tsplib_dir, "a280.tsp"))["xy"]
        assert_same(gpu_lk(ctx, xy, seed=7, epochs=30), O.lin_kernighan(xy, seed=7, epochs=30))
    elif case == "synth1000":
        xy = O.synth_xy(1000, seed=5)
        assert_same(gpu_lk(ctx, xy, seed=3, epochs=12), O.lin_kernighan(xy, seed=3, epochs=12))
    elif case == "lattice":
        xy = lattice(10, 2)
        assert_same(gpu_lk(ctx, xy, seed=11, epochs=40), O.lin_kernighan(xy, seed=11, epochs=40))
    elif case == "random_start":
        xy = O.synth_xy(600, seed=8)
        init = O.restart_perm(600, 5, 0)
        assert_same(gpu_lk(ctx, xy, init=init, seed=2, epochs=5), O.lin_kernighan(xy, init=init, seed=2, epochs=5))
    else:
        xy = O.synth_xy(200, seed=4)
        for depth, k in ((1, 5), (2, 3), (3, 8), (6, 5)):
            assert_same(gpu_lk(ctx, xy, seed=1, epochs=20, max_depth=depth, n_nearest=k),
                        O.lin_kernighan(xy, seed=1, epochs=20, max_depth=depth, n_nearest=k))


def test_lk_wide_candidate_lists(ctx):
    # k(k+1)^2 > 1024 sub-searches per pair: the flat (divided) lane index instead of one workgroup per pair
    xy = O.synth_xy(300, seed=12)
    for k, depth in ((10, 5), (12, 3), (16, 4)):
        assert_same(gpu_lk(ctx, xy, seed=3, epochs=6, n_nearest=k, max_depth=depth),
                    O.lin_kernighan(xy, seed=3, epochs=6, n_nearest=k, max_depth=depth))


def test_product_library_refuses_tune_only_flags():
    # the forms DESIGN §4.6 measured and rejected are not in the product library
    import teeline_amd as TA
    if _on_tune_build():
        pytest.skip("this process runs on the tuning build")
    for flag in (TA.TL_FLAG_LK_NO_SPLIT, TA.TL_FLAG_LK_SPLIT2, TA.TL_FLAG_LK_NO_SUBCHAINS, TA.TL_FLAG_LK_SMALL, TA.TL_FLAG_LK_SEPARATE_PICK,
                 TA.TL_FLAG_LK_NO_GRAPH, TA.TL_FLAG_LK_SEPARATE_STEP, TA.TL_FLAG_KNN_4LANES, TA.TL_FLAG_KNN_1LANE):
        with pytest.raises(TA.TeelineGpuError) as e:
            TA.Context(0, flag)
        assert e.value.code == TA._capi.TL_ERR_UNSUPPORTED


def test_rejected_forms_on_the_tune_build():
    # the cross-checks of the rejected forms (the two tests marked TUNE_ONLY / test_lk_variants_are_identical) against the oracle,
    # in ONE child process bound to libteeline_gpu_tune.so (a process binds one library)
    import subprocess
    import sys
    if _on_tune_build():
        pytest.skip("already the child")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tune = os.path.join(root, "teeline_amd", "libteeline_gpu_tune.so")
    assert os.path.exists(tune), "built by __graft_entry__.build() / python -m teeline_amd.build"
    env = dict(os.environ, TEELINE_GPU_LIB=tune)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-x", "-q", "-k",
                        "variants_are_identical or builders_agree"], env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "skipped" not in r.stdout.splitlines()[-1], r.stdout[-500:]


@TUNE_ONLY
def test_lk_variants_are_identical(tsplib_dir):
    # default = scans spread over all CUs, each pair's chain search split into k*(k+1)^2 sub-searches (TL_FLAG_LK_SPLIT2: k*(k+1)), device-side control
    # state machine, the pair's first chain picked and validated by the scan workgroup itself (TL_FLAG_LK_SEPARATE_PICK: by a kernel of
    # its own, from kept sub-search chains).  The unsplit scan, the single persistent workgroup and the pick step that walks the
    # winning chain again must reproduce the same results (= the oracle's).
    xy = T.parse_tsplib(os.path.join(tsplib_dir, "berlin52.tsp"))["xy"]
    lat = lattice(10, 2)
    xy2 = O.synth_xy(200, seed=4)
    sq = np.array([[0, 0], [1, 1], [1, 0], [0, 1]], np.float32)
    tri = np.array([[0, 0], [1, 0], [0.5, 1]], np.float32)
    import teeline_amd as TA
    for flag in (TA.TL_FLAG_LK_NO_SPLIT, TA.TL_FLAG_LK_ONE_WORKGROUP, TA.TL_FLAG_LK_NO_SUBCHAINS, TA.TL_FLAG_LK_SPLIT2, TA.TL_FLAG_LK_SMALL,
                 TA.TL_FLAG_LK_SEPARATE_PICK, TA.TL_FLAG_LK_SEPARATE_PICK | TA.TL_FLAG_LK_SPLIT2, TA.TL_FLAG_LK_NO_GRAPH, TA.TL_FLAG_LK_SEPARATE_STEP, TA.TL_FLAG_LK_SCAN_PERSIST):
        with TA.Context(0, flag) as ctx:
            for seed in (1, 2):
                assert_same(gpu_lk(ctx, xy, seed=seed), O.lin_kernighan(xy, seed=seed))
            assert_same(gpu_lk(ctx, lat, seed=11, epochs=40), O.lin_kernighan(lat, seed=11, epochs=40))
            assert_same(gpu_lk(ctx, sq, init=[0, 1, 2, 3], max_depth=1, epochs=0), O.lin_kernighan(sq, init=[0, 1, 2, 3], epochs=0, max_depth=1))
            assert gpu_lk(ctx, tri, init=[2, 0, 1])[0].tolist() == [2, 0, 1]
            for depth, k in ((1, 5), (2, 3), (6, 5)):
                assert_same(gpu_lk(ctx, xy2, seed=1, epochs=20, max_depth=depth, n_nearest=k),
                            O.lin_kernighan(xy2, seed=1, epochs=20, max_depth=depth, n_nearest=k))
    big = O.synth_xy(2000, seed=6)
    want = O.lin_kernighan(big, seed=5, epochs=8)
    # n >= 1500: the chip-wide step kernel (state machine + move application in one launch) is the default;
    # TL_FLAG_LK_SEPARATE_STEP / _SEPARATE_PICK / _SPLIT2 run the two-kernel forms
    for flag in (0, TA.TL_FLAG_LK_SEPARATE_STEP, TA.TL_FLAG_LK_SEPARATE_PICK, TA.TL_FLAG_LK_SPLIT2, TA.TL_FLAG_LK_NO_GRAPH, TA.TL_FLAG_LK_SCAN_PERSIST):
        with TA.Context(0, flag) as ctx:
            assert_same(gpu_lk(ctx, big, seed=5, epochs=8), want)


@pytest.mark.parametrize("name", ["gr17", "bays29", "ring6_explicit", "burma14"])
def test_nn_seed_and_lk_on_explicit_and_geo_problems(ctx, name, tsplib_dir):
    """GEO / EXPLICIT: nearest_neighbor::solve walks problem.distances (nearest_neighbor.rs:44-63 over
    distance_matrix.rs:259-297); lin_kernighan::solve searches over the Euclidean matrix rebuilt from the city coordinates
    (:41) but seeds with that NN walk (:47-55) and reports problem.distances.tour_length (:99)."""
    import teeline_amd as TA
    e = T.parse_tsplib(os.path.join(tsplib_dir, f"{name}.tsp"))
    xy, n = e["xy"], e["n"]
    packed = e["packed"] if e["packed"] is not None else O.dm_build_packed(xy, geo=True)
    kind = "explicit" if e["packed"] is not None else "geo"
    p = TA.TspProblem(e["ids"], xy, TA.distance_matrix.DistanceMatrix(n, packed, e["ids"], kind))
    for k in (1, 3, 5):
        sol = TA.nearest_neighbor.solve(p, TA.HeuristicOptions(n_nearest=k), ctx=ctx)
        rc, route, c = O.nearest_neighbor(None, packed, n, k)
        assert list(sol.route()) == e["ids"][route].tolist() and np.float32(sol.total).tobytes() == np.float32(c).tobytes()
    for init in (None, list(e["ids"][::-1])):
        h = TA.HeuristicOptions(epochs=20, platoo_epochs=10, n_nearest=5)
        sol = TA.lin_kernighan.solve(p, TA.LKOptions(h, 5), None, init, ctx=ctx, seed=3)
        oinit = None if init is None else np.arange(n - 1, -1, -1, dtype=np.uint32)
        rc, route, c, st = O.lin_kernighan(xy, init=oinit, epochs=20, seed=3, packed=packed)
        assert list(sol.route()) == e["ids"][route].tolist()
        assert np.float32(sol.total).tobytes() == np.float32(c).tobytes()
        assert np.float32(sol.total) == O.tour_length(None, packed, route)   # the total is problem.distances', not Euclid


def test_lk_tiny_problems_return_the_initial_tour(ctx):
    # lin_kernighan.rs:57-59: fewer than 4 cities -> Solution::new(&best_tour) without a pass; n = 1 used to reach the GPU
    # with an empty candidate list
    for n in (1, 2, 3):
        xy = O.synth_xy(n, seed=n)
        g = gpu_lk(ctx, xy)
        rc, route, c, st = O.lin_kernighan(xy)
        assert g[0].tolist() == route.tolist() and np.float32(g[1]).tobytes() == np.float32(c).tobytes()
        assert g[2]["moves"] == 0
    g = gpu_lk(ctx, O.synth_xy(3, seed=9), init=[2, 0, 1])
    assert g[0].tolist() == [2, 0, 1]


def test_short_init_tours_are_rejected_on_the_host(ctx):
    # a tour crosses the C ABI as n u32 values: a shorter list must never reach the library
    import teeline_amd as TA
    xy = O.synth_xy(50, seed=2)
    p = prob(xy)
    short = list(range(40))
    for fn in (lambda: TA.two_opt.solve(p, None, None, short, ctx=ctx), lambda: TA.three_opt.solve(p, None, None, short, ctx=ctx),
               lambda: TA.or_opt.solve(p, None, None, short, ctx=ctx), lambda: TA.lin_kernighan.solve(p, None, None, short, ctx=ctx),
               lambda: TA.three_opt.find_best_move(p, short, ctx=ctx), lambda: TA.or_opt.find_best_move(p, short, ctx=ctx)):
        with pytest.raises(TA.TeelineGpuError):
            fn()


def test_lk_forms_at_the_chip_step_size():
    # n >= 1500 is where the chip-wide step kernel takes over; list lengths 1..10 cross from the one-workgroup-per-pair scan
    # (k (k+1)^2 <= 1024 lanes: k <= 9) to the flat form, depths 1..6 cover every cut of the split and of the parked walks
    import teeline_amd as TA
    # (round 5: the scan reads the PACKED view by default — candidates with their distances, successor records kept by the step kernel —
    #  and the classic cand -> xy -> next -> xy look-ups under TL_FLAG_LK_CLASSIC_VIEW: both against the oracle)
    xy = O.synth_xy(1600, seed=9)
    want = {}
    for flags in (0, TA.TL_FLAG_LK_CLASSIC_VIEW):
        with TA.Context(0, flags) as ctx:
            for k, depth, epochs in ((1, 5, 3), (2, 3, 3), (3, 6, 4), (5, 4, 4), (5, 2, 3), (5, 1, 2), (9, 3, 2), (10, 3, 2)):
                if (k, depth) not in want:
                    want[(k, depth)] = O.lin_kernighan(xy, seed=7, epochs=epochs, max_depth=depth, n_nearest=k)
                assert_same(gpu_lk(ctx, xy, seed=7, epochs=epochs, max_depth=depth, n_nearest=k), want[(k, depth)])


def test_progress_channel_carries_the_reference_messages(ctx, tsplib_dir):
    # lin_kernighan.rs:71,90: PathUpdate(best_tour, best_dist) after the first lk_pass and after every epoch that improves on it, no
    # Done.  With a progress callback lin_kernighan::solve goes through tl_lk_live (round 4): the device-side state machine files
    # exactly those tours and distances in a ring and the host hands them to the callback while the search runs (oracle:
    # tlo_lin_kernighan_trace, the same loop with the messages recorded); tl_lk_trace lists the same ones after the fact (below).
    import ctypes as C
    import teeline_amd as TA
    from teeline_amd import _capi
    b = T.parse_tsplib(os.path.join(tsplib_dir, "berlin52.tsp"))
    cases = [(b["xy"], b["ids"], None, dict(epochs=60, seed=1)), (b["xy"], b["ids"], None, dict(epochs=60, seed=7)),
             (O.synth_xy(400, seed=9), np.arange(400), O.restart_perm(400, 5, 0), dict(epochs=12, seed=3)),
             (O.synth_xy(2000, seed=4), np.arange(2000), None, dict(epochs=6, seed=2)),   # chip-wide step (n >= 1500)
             (lattice(8, 3), np.arange(64), None, dict(epochs=20, seed=5, n_nearest=6, max_depth=4))]
    for xy, ids, init, kw in cases:
        n = len(ids)
        rc, oroute, ocost, ost, osnaps = O.lin_kernighan_trace(xy, init=init, epochs=kw["epochs"], platoo_epochs=10, n_nearest=kw.get("n_nearest", 5),
                                                               max_depth=kw.get("max_depth", 5), seed=kw["seed"])
        assert rc == 0 and len(osnaps) >= 1
        got = []
        h = TA.HeuristicOptions(epochs=kw["epochs"], platoo_epochs=10, n_nearest=kw.get("n_nearest", 5))
        sol = TA.lin_kernighan.solve(TA.TspProblem(ids, xy), TA.LKOptions(h, kw.get("max_depth", 5)), lambda kind, payload: got.append((kind, payload)),
                                     None if init is None else [int(ids[v]) for v in init], ctx=ctx, seed=kw["seed"])
        assert [k for k, _ in got] == ["PathUpdate"] * len(osnaps)
        for (kind, (route, dist)), (spos, sdist) in zip(got, osnaps):
            assert route == [int(ids[v]) for v in spos] and np.float32(dist).tobytes() == np.float32(sdist).tobytes()
        assert list(sol.route()) == [int(ids[v]) for v in oroute] and np.float32(sol.total).tobytes() == np.float32(ocost).tobytes()
        assert got[-1][1][0] == list(sol.route())  # the last message carries the returned tour
    # tl_lk_trace: the same list after the fact, through the C ABI directly (the mirrors use tl_lk_live)
    xy, ids = b["xy"], b["ids"]
    n = len(ids)
    rc, oroute, ocost, ost, osnaps = O.lin_kernighan_trace(xy, epochs=60, seed=1)
    out = np.empty(n, dtype=np.uint32)
    snaps = np.zeros((64, n), dtype=np.uint32)
    dists = np.zeros(64, dtype=np.float32)
    c, st, ln = C.c_float(), _capi.TlStats(), C.c_uint32()
    o = _capi.TlLkOpts(60, 10, 5, 5)
    ctx.check(ctx.lib.tl_lk_trace(ctx.handle, xy.ctypes.data_as(C.c_void_p), n, None, None, C.byref(o), 1, out.ctypes.data_as(C.c_void_p), C.byref(c),
                                  C.byref(st), snaps.ctypes.data_as(C.c_void_p), dists.ctypes.data_as(C.c_void_p), 64, C.byref(ln)))
    assert ln.value == len(osnaps) and out.tolist() == oroute.tolist()
    for m, (spos, sdist) in enumerate(osnaps):
        assert snaps[m].tolist() == spos.tolist() and dists[m].tobytes() == np.float32(sdist).tobytes()
    # a short buffer holds the first snapshots and reports the full count
    xy, ids = b["xy"], b["ids"]
    n = len(ids)
    rc, oroute, ocost, ost, osnaps = O.lin_kernighan_trace(xy, epochs=60, seed=7)
    if len(osnaps) > 1:
        out = np.empty(n, dtype=np.uint32)
        snaps = np.full((1, n), 0xFFFFFFFF, dtype=np.uint32)
        dists = np.zeros(1, dtype=np.float32)
        c, st, ln = C.c_float(), _capi.TlStats(), C.c_uint32()
        o = _capi.TlLkOpts(60, 10, 5, 5)
        ctx.check(ctx.lib.tl_lk_trace(ctx.handle, xy.ctypes.data_as(C.c_void_p), n, None, None, C.byref(o), 7, out.ctypes.data_as(C.c_void_p), C.byref(c),
                                      C.byref(st), snaps.ctypes.data_as(C.c_void_p), dists.ctypes.data_as(C.c_void_p), 1, C.byref(ln)))
        assert ln.value == len(osnaps) and snaps[0].tolist() == osnaps[0][0].tolist() and out.tolist() == oroute.tolist()


def test_progress_messages_arrive_while_the_search_runs(ctx):
    """VERDICT r03 "missing 5": the reference sends while it runs (lin_kernighan.rs:71,90; teeline-qt watches the channel).  tl_lk_live
    calls back between two polls of the device-side search: on a run of a few hundred milliseconds the first message (the tour after
    the first lk_pass) must arrive long before the call returns, the messages must be spread over the run, and more than a ring's
    worth of them (64) must all arrive, in order — content against the oracle's record."""
    import time
    import teeline_amd as TA
    n = 1000
    xy = O.synth_xy(n, seed=12)
    stamps, got = [], []

    def tx(kind, payload):
        stamps.append(time.perf_counter())
        got.append((kind, payload))

    h = TA.HeuristicOptions(epochs=1500, platoo_epochs=1500, n_nearest=5)   # ~18 000 rounds, ~0.5 s
    t0 = time.perf_counter()
    sol = TA.lin_kernighan.solve(prob(xy), TA.LKOptions(h, 5), tx, None, ctx=ctx, seed=4)
    t1 = time.perf_counter()
    rc, oroute, ocost, ost, osnaps = O.lin_kernighan_trace(xy, epochs=1500, platoo_epochs=1500, seed=4, cap=512)
    assert rc == 0 and len(osnaps) > 64, "the instance should improve in more epochs than the ring holds"
    assert len(got) == len(osnaps)
    for (kind, (route, dist)), (spos, sdist) in zip(got, osnaps):
        assert kind == "PathUpdate" and route == spos.tolist() and np.float32(dist).tobytes() == np.float32(sdist).tobytes()
    assert list(sol.route()) == oroute.tolist()
    run = t1 - t0
    assert stamps[0] - t0 < 0.5 * run, (stamps[0] - t0, run)          # the first message is not held back until the end
    assert stamps[-1] - stamps[0] > 0.1 * run, (stamps[0] - t0, stamps[-1] - t0, run)  # ... and the others come spread over the run


def test_live_callback_cannot_re_enter_its_own_context(ctx):
    # the callback runs on the owning thread, so the owner guard used to let it back in — onto the stream and workspace of
    # the running search.  Now every entry into THIS context from inside the callback is TL_ERR_BUSY; another context is free; and the
    # search's result is untouched.
    import ctypes as C
    import teeline_amd as TA
    from teeline_amd import _capi
    n = 400
    xy = O.synth_xy(n, seed=31)
    seen = []
    out = np.empty(n, dtype=np.uint32)
    cost, st = C.c_float(), _capi.TlStats()
    with TA.Context(0) as other:
        @_capi.LK_PROGRESS_FN
        def cb(user, best_pos, nn, best_dist):
            ms = C.c_double()
            rc_same = ctx.lib.tl_last_kernel_ms(ctx.handle, C.byref(ms))
            tmp = np.empty(n, dtype=np.uint32)
            c2 = C.c_float()
            rc_same2 = ctx.lib.tl_nearest_neighbor(ctx.handle, xy.ctypes.data_as(C.c_void_p), None, n, 3, tmp.ctypes.data_as(C.c_void_p), C.byref(c2))
            rc_other = other.lib.tl_nearest_neighbor(other.handle, xy.ctypes.data_as(C.c_void_p), None, n, 3, tmp.ctypes.data_as(C.c_void_p), C.byref(c2))
            seen.append((rc_same, rc_same2, rc_other))
        o = _capi.TlLkOpts(40, 10, 5, 5)
        ctx.check(ctx.lib.tl_lk_live(ctx.handle, xy.ctypes.data_as(C.c_void_p), n, None, None, C.byref(o), 9, out.ctypes.data_as(C.c_void_p), C.byref(cost),
                                     C.byref(st), cb, None))
    assert seen and all(s == (_capi.TL_ERR_BUSY, _capi.TL_ERR_BUSY, 0) for s in seen), seen[:4]
    rc, oroute, ocost, ost = O.lin_kernighan_trace(xy, epochs=40, platoo_epochs=10, seed=9)[:4]
    assert out.tolist() == oroute.tolist() and np.float32(cost.value).tobytes() == np.float32(ocost).tobytes()
    ms = C.c_double()
    assert ctx.lib.tl_last_kernel_ms(ctx.handle, C.byref(ms)) == 0          # ... and the context is usable again afterwards


def test_no_progress_messages_below_four_cities(ctx):
    # lin_kernighan.rs:57-59 returns before its first send_progress
    import teeline_amd as TA
    got = []
    sol = TA.lin_kernighan.solve(prob(O.synth_xy(3, seed=2)), TA.LKOptions(TA.HeuristicOptions(epochs=5), 5), lambda k, p: got.append((k, p)), None, ctx=ctx)
    assert got == [] and len(sol.route()) == 3
