"""GPU parity tests (-m gpu): the HIP path, called through the C ABI, against the CPU oracle.

Bit-exact bar: tours element-for-element, costs bit-for-bit (f32), sweep/move/reversal counters equal.
"""
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


def gpu_two_opt(ctx, xy, packed, n, init=None, mode=0):
    import teeline_amd as TA
    prob = TA.TspProblem(np.arange(n), xy if xy is not None else np.zeros((n, 2), np.float32),
                         None if packed is None else TA.distance_matrix.DistanceMatrix(n, packed, np.arange(n), "explicit"))
    sol = TA.two_opt.solve(prob, None, None, None if init is None else [int(v) for v in init], ctx=ctx, mode=mode)
    return np.asarray(sol.route(), dtype=np.uint32), sol.total, sol.stats


def assert_same(gpu, ora, n):
    route, cost, st = gpu
    rc, oroute, ocost, ost = ora
    assert rc == 0
    assert route.tolist() == oroute.tolist(), "tour differs from the oracle"
    assert np.float32(cost).tobytes() == np.float32(ocost).tobytes(), f"cost {cost!r} != {ocost!r}"
    for k in ("sweeps", "candidates", "moves", "reversed"):
        assert st[k] == ost[k], f"{k}: gpu {st[k]} != oracle {ost[k]}"


def test_device_is_gfx950_and_lds_limit(ctx):
    info = ctx.device_info()
    assert info["arch"].startswith("gfx950") and info["cus"] >= 200
    assert ctx.two_opt_lds_max_n() >= 13509  # configs 3 and 5 sizes fit one CU's LDS


def test_tsp5_reference_unit_tests(ctx):
    # two_opt.rs:100-131
    pts = np.array([[0.0, 0.0], [0.0, 0.5], [0.0, 1.0], [1.0, 1.0], [1.0, 0.0]], dtype=np.float32)
    route, cost, st = gpu_two_opt(ctx, pts, None, 5)
    assert route.tolist() == [0, 1, 2, 3, 4] and cost == np.float32(4.0)
    route, cost, st = gpu_two_opt(ctx, pts, None, 5, init=[0, 1, 2, 3, 4])
    assert route.tolist() == [0, 1, 2, 3, 4]


@pytest.mark.parametrize("name", ["berlin52", "a280", "att532", "att48"])
def test_tsplib_euclid_matches_oracle_and_goldens(ctx, name, tsplib_dir, goldens):
    d = T.parse_tsplib(os.path.join(tsplib_dir, f"{name}.tsp"))
    xy, n, ids = d["xy"], d["n"], d["ids"]
    rc, nn, _ = O.nearest_neighbor(xy, None, n, 3)
    for key, init in (("identity_two_opt", None), ("nn_two_opt", nn)):
        g = gpu_two_opt(ctx, xy, None, n, init)
        assert_same(g, O.two_opt(xy, None, n, init=init), n)
        assert f5(g[1]) == goldens[name][key]["cost"]
        assert ids[g[0]].tolist() == goldens[name][key]["route_ids"]


def test_edge_sizes(ctx):
    import teeline_amd as TA
    xy = O.synth_xy(70, seed=11)
    for n in (3, 4, 5, 6, 7, 63, 64, 65, 66, 67, 70):
        assert_same(gpu_two_opt(ctx, xy[:n], None, n), O.two_opt(xy[:n], None, n), n)
    for n in (1, 2):  # the reference panics (usize underflow, two_opt.rs:17,29)
        with pytest.raises(TA.ReferencePanics):
            gpu_two_opt(ctx, xy[:n], None, n)
    with pytest.raises(TA.TeelineGpuError):  # not a permutation
        prob = TA.TspProblem(np.arange(5), xy[:5])
        out = np.empty(5, np.uint32)
        import ctypes as C
        bad = np.array([0, 1, 1, 3, 4], np.uint32)
        ctx.check(ctx.lib.tl_two_opt(ctx.handle, prob.xy.ctypes.data_as(C.c_void_p), 5, None,
                                     bad.ctypes.data_as(C.c_void_p), 0, out.ctypes.data_as(C.c_void_p), None, None))


def test_duplicate_points_and_ties(ctx):
    # collisions: repeated coordinates and a lattice (many exactly equal distances -> strict `<` matters)
    g = np.stack(np.meshgrid(np.arange(12, dtype=np.float32), np.arange(12, dtype=np.float32)), -1).reshape(-1, 2)
    rng = np.random.default_rng(5)
    lattice = g[rng.permutation(len(g))]
    assert_same(gpu_two_opt(ctx, lattice, None, len(lattice)), O.two_opt(lattice, None, len(lattice)), len(lattice))
    dup = np.concatenate([lattice[:50], lattice[:50], np.zeros((10, 2), np.float32)])
    dup = np.ascontiguousarray(dup[rng.permutation(len(dup))])
    assert_same(gpu_two_opt(ctx, dup, None, len(dup)), O.two_opt(dup, None, len(dup)), len(dup))


@pytest.mark.parametrize("n,seed", [(257, 1), (1002, 0), (2000, 9), (4097, 4)])
def test_synthetic_random_and_greedy_starts(ctx, n, seed):
    xy = O.synth_xy(n, seed=seed)
    rc, nn, _ = O.nearest_neighbor(xy, None, n, 3)
    assert_same(gpu_two_opt(ctx, xy, None, n, nn), O.two_opt(xy, None, n, init=nn), n)
    rp = O.restart_perm(n, 777, 2)
    assert_same(gpu_two_opt(ctx, xy, None, n, rp), O.two_opt(xy, None, n, init=rp), n)


def test_synthetic_10000_full_size(ctx, goldens):
    # BASELINE config 3 at full size against the committed golden (oracle: 12 sweeps, 599 700 036 candidates)
    n = 10000
    g = goldens["synthetic10000"]
    xy = O.synth_xy(n)
    rc, nn, cnn = O.nearest_neighbor(xy, None, n, 3)
    assert f5(cnn) == g["nn_cost"]
    route, cost, st = gpu_two_opt(ctx, xy, None, n, nn)
    w = np.arange(1, n + 1, dtype=np.uint32)
    assert f5(cost) == g["nn_two_opt"]["cost"] == "77647.55469"
    assert {k: st[k] for k in ("sweeps", "candidates", "moves", "reversed")} == g["nn_two_opt"]["stats"]
    assert int(np.bitwise_xor.reduce(route * w)) == g["nn_two_opt"]["route_crc"]
    assert O.validate_tour(route) and route[0] == nn[0] and route[-1] == nn[-1]  # open path endpoints fixed
    # size-independent property: the result is a fixed point of the reference's sweep
    rc, again, c2, st2 = O.two_opt(xy, None, n, init=route, max_candidates=1)
    assert st2["moves"] == 0 and again.tolist() == route.tolist() and c2 == cost
    # random restart start (70 970 moves)
    rp = O.restart_perm(n, 12345, 0)
    route, cost, st = gpu_two_opt(ctx, xy, None, n, rp)
    gg = g["restart0_seed12345_two_opt"]
    assert f5(cost) == gg["cost"] and {k: st[k] for k in ("sweeps", "candidates", "moves", "reversed")} == gg["stats"]
    assert int(np.bitwise_xor.reduce(route * w)) == gg["route_crc"]


def test_no_prune_flag_gives_identical_results(tsplib_dir):
    import teeline_amd as TA
    with TA.Context(0, TA.TL_FLAG_NO_PRUNE) as c2:
        for n, seed in ((500, 3), (1500, 8)):
            xy = O.synth_xy(n, seed=seed)
            rp = O.restart_perm(n, 1, 0)
            assert_same(gpu_two_opt(c2, xy, None, n, rp), O.two_opt(xy, None, n, init=rp), n)


@pytest.mark.parametrize("name", ["gr17", "ring6_explicit", "bays29"])
def test_explicit_matrix_form(ctx, name, tsplib_dir, goldens):
    d = T.parse_tsplib(os.path.join(tsplib_dir, f"{name}.tsp"))
    g = gpu_two_opt(ctx, None, d["packed"], d["n"])
    assert_same(g, O.two_opt(None, d["packed"], d["n"]), d["n"])
    assert g[0].tolist() == goldens[name]["identity_two_opt"]["route_pos"]


def test_matrix_form_equals_on_the_fly_form_pr1002_size(ctx):
    # BASELINE config 2 shape (pr1002 itself is not in the reference tree: synthetic n = 1002, labelled)
    n = 1002
    xy = O.synth_xy(n)
    packed = O.dm_build_packed(xy)
    rp = O.restart_perm(n, 5, 1)
    a = gpu_two_opt(ctx, xy, None, n, rp)
    b = gpu_two_opt(ctx, None, packed, n, rp)
    assert a[0].tolist() == b[0].tolist() and a[1] == b[1]
    assert_same(b, O.two_opt(None, packed, n, init=rp), n)


@pytest.mark.parametrize("n,start", [(1002, "identity"), (1002, "nn"), (1500, "random"), (3000, "random"), (9000, "random")])
def test_matrix_form_deferred_rows(ctx, n, start):
    """The matrix-form kernel on starts with moves every few candidates (identity, random: rows with many chained hits, one reversal
    per hit behind the step's barrier), the NN start (wide 16-row steps) and sizes up to 9 000 (reversals longer than 8 192
    positions).  Tour, cost bits, sweeps, moves, reversal count against the oracle on the same matrix.  (Written in round 4 for a
    form of the kernel that defers a dense row's reversals to the row's end and composes them — commit b6c7f95, parity green, not
    faster: NOTEBOOK.md round 4 — and kept as coverage of the kernel that stayed.)"""
    xy = O.synth_xy(n, seed=31)
    packed = O.dm_build_packed(xy)
    init = None if start == "identity" else (O.nearest_neighbor(xy, None, n, 3)[1] if start == "nn" else O.restart_perm(n, 17, 0))
    assert_same(gpu_two_opt(ctx, None, packed, n, init), O.two_opt(None, packed, n, init=init), n)


def test_multistart_matches_per_restart_oracle(ctx):
    import teeline_amd as TA
    n, R, seed = 600, 12, 4242
    xy = O.synth_xy(n, seed=21)
    prob = TA.TspProblem(np.arange(n), xy)
    sol, costs = TA.two_opt.multistart(prob, R, seed=seed, first=3, ctx=ctx, return_costs=True)
    ocosts, oroutes = [], []
    for r in range(3, 3 + R):
        rc, p, c, st = O.two_opt(xy, None, n, init=O.restart_perm(n, seed, r))
        ocosts.append(c)
        oroutes.append(p)
    assert [np.float32(c).tobytes() for c in costs] == [np.float32(c).tobytes() for c in ocosts]
    keys = [ctx.lib.tl_pack_cost_key(float(c), 3 + i) for i, c in enumerate(ocosts)]
    best = int(np.argmin(keys))
    assert sol.stats["best_restart"] == 3 + best
    assert list(sol.route()) == oroutes[best].tolist() and sol.total == ocosts[best]


def test_multistart_over_several_contexts_is_independent_of_the_split(ctx):
    # tl_two_opt_multistart_devices: one process, one context per device (here: 1, 2, 3 and 5 contexts on the one GPU of
    # the test box); the deal of restarts over the contexts must not show in any output
    import teeline_amd as TA
    n, R, seed = 500, 11, 99
    xy = O.synth_xy(n, seed=17)
    prob = TA.TspProblem(np.arange(n), xy)
    ref, ref_costs = TA.two_opt.multistart(prob, R, seed=seed, first=5, ctx=ctx, return_costs=True)
    ocosts = [O.two_opt(xy, None, n, init=O.restart_perm(n, seed, r))[2] for r in range(5, 5 + R)]
    assert [np.float32(c).tobytes() for c in ref_costs] == [np.float32(c).tobytes() for c in ocosts]
    for k in (1, 2, 3, 5, 12):
        cs = [TA.Context(0) for _ in range(k)]
        try:
            sol, costs = TA.two_opt.multistart_devices(prob, R, cs, seed=seed, first=5, return_costs=True)
        finally:
            [c.close() for c in cs]
        assert costs.tobytes() == ref_costs.tobytes() and list(sol.route()) == list(ref.route())
        assert sol.total == ref.total and sol.stats["best_restart"] == ref.stats["best_restart"]
        for key in ("sweeps", "candidates", "moves", "reversed"):
            assert sol.stats[key] == ref.stats[key]


def test_multistart_collective_through_rccl_inside_the_library(ctx):
    # TL_FLAG_MULTISTART_RCCL (round 5): the library itself min-all-reduces the devices' packed (cost, restart) keys over RCCL and
    # broadcasts the winner's tour from its owner (one process, ncclCommInitAll; librccl dlopen'ed on first use).  The test box has ONE
    # GPU, so this is the one-device communicator — every call of the path (group start / all-reduce / broadcast / the hand-out from
    # device 0's buffer) runs, over a communicator of size 1; the result must be the host-minimum path's.
    import teeline_amd as TA
    n, R, seed = 700, 9, 4
    xy = O.synth_xy(n, seed=23)
    prob = TA.TspProblem(np.arange(n), xy)
    ref, ref_costs = TA.two_opt.multistart(prob, R, seed=seed, first=2, ctx=ctx, return_costs=True)
    with TA.Context(0, TA.TL_FLAG_MULTISTART_RCCL) as cr:
        for _ in range(2):  # (the second call finds the communicator cached)
            sol, costs = TA.two_opt.multistart_devices(prob, R, [cr], seed=seed, first=2, return_costs=True)
            assert costs.tobytes() == ref_costs.tobytes() and list(sol.route()) == list(ref.route())
            assert sol.total == ref.total and sol.stats["best_restart"] == ref.stats["best_restart"]


def test_dm_is_euc2d_tells_coordinate_matrices_from_explicit_ones(ctx, tsplib_dir):
    # the reference's DistanceMatrix keeps no DistanceType (distance_matrix.rs:86-93): the shim asks the library
    import teeline_amd as TA
    xy = O.synth_xy(700, seed=3)
    packed = O.dm_build_packed(xy)
    assert TA.distance_matrix.is_euc2d(xy, packed, ctx=ctx)
    bad = packed.copy()
    bad[len(bad) // 3] = np.nextafter(bad[len(bad) // 3], np.float32(np.inf))
    assert not TA.distance_matrix.is_euc2d(xy, bad, ctx=ctx)
    bad = packed.copy()
    bad[-1] += 1.0
    assert not TA.distance_matrix.is_euc2d(xy, bad, ctx=ctx)
    e = T.parse_tsplib(os.path.join(tsplib_dir, "burma14.tsp"))
    assert not TA.distance_matrix.is_euc2d(e["xy"], O.dm_build_packed(e["xy"], geo=True), ctx=ctx)
    assert TA.distance_matrix.is_euc2d(e["xy"], O.dm_build_packed(e["xy"]), ctx=ctx)
    g = T.parse_tsplib(os.path.join(tsplib_dir, "gr17.tsp"))
    assert not TA.distance_matrix.is_euc2d(g["xy"], g["packed"], ctx=ctx)


def test_device_init_tour_with_out_of_range_entry_is_refused(ctx):
    # tl_two_opt_batch_dev takes device-resident initial tours the host cannot check: an entry >= n must not index xy
    import ctypes as C
    import torch
    from teeline_amd import _capi
    n, R = 300, 3
    xy = O.synth_xy(n, seed=5)
    inits = np.stack([np.arange(n, dtype=np.uint32) for _ in range(R)])
    inits[1, 17] = n + 5
    dev = torch.device("cuda", 0)
    d_xy, d_init = torch.from_numpy(xy).to(dev), torch.from_numpy(inits.astype(np.int32)).to(dev)
    d_pos = torch.zeros((R, n), dtype=torch.int32, device=dev)
    d_cost = torch.zeros(R, dtype=torch.float32, device=dev)
    d_stats = torch.zeros((R, _capi.TL_DEV_STATS_STRIDE), dtype=torch.int64, device=dev)
    s = torch.cuda.current_stream()
    ctx.check(ctx.lib.tl_two_opt_batch_dev(ctx.handle, d_xy.data_ptr(), n, d_init.data_ptr(), 0, 0, R, 0, d_pos.data_ptr(),
                                           d_cost.data_ptr(), d_stats.data_ptr(), C.c_void_p(s.cuda_stream)))
    torch.cuda.synchronize()
    st, cost = d_stats.cpu().numpy(), d_cost.cpu().numpy()
    assert st[:, 3].tolist() == [0, 2, 0] and np.isnan(cost[1]) and not np.isnan(cost[0]) and not np.isnan(cost[2])
    rc, p, c, _ = O.two_opt(xy, None, n)
    assert d_pos[0].cpu().numpy().tolist() == p.tolist() == d_pos[2].cpu().numpy().tolist()


def test_tour_length_and_dm_build(ctx, tsplib_dir):
    import teeline_amd as TA
    b = T.parse_tsplib(os.path.join(tsplib_dir, "berlin52.tsp"))
    dm = TA.distance_matrix.build(b["ids"], b["xy"], ctx=ctx)
    assert np.array_equal(dm.items, O.dm_build_packed(b["xy"]))
    opt = T.parse_opt_tour(os.path.join(tsplib_dir, "berlin52.opt.tour"))
    assert f5(dm.tour_length(opt, ctx=ctx)) == "7544.36572"  # README.md:365
    # n = 10^4: 49 995 000 correctly rounded sqrt, bit-for-bit (also the sqrt_rn check over dense mantissas)
    xy = O.synth_xy(10000)
    dmb, ms = TA.distance_matrix.build(np.arange(10000), xy, ctx=ctx, return_ms=True)
    ref = O.dm_build_packed(xy)
    assert np.array_equal(dmb.items.view(np.uint32), ref.view(np.uint32))
    perm = O.restart_perm(10000, 3, 0)
    assert dmb.tour_length_by_pos(perm, ctx=ctx).tobytes() == O.tour_length(xy, None, perm).tobytes()
    # GEO (distance_matrix.rs:59-75, 457-464)
    d = T.parse_tsplib(os.path.join(tsplib_dir, "burma14.tsp"))
    geo = TA.distance_matrix.build(d["ids"], d["xy"], kind="geo", ctx=ctx)
    assert np.array_equal(geo.items, O.dm_build_packed(d["xy"], geo=True))
    pair = TA.distance_matrix.build([1, 2], np.array([[16.47, 96.10], [23.70, 96.99]], np.float32), kind="geo", ctx=ctx)
    assert pair.items.tolist() == [837.0]


# ---------------------------------------------------------------- TL_MODE_BEST_SWEEP (own mode; oracle = tlo_two_opt_best)
@pytest.mark.parametrize("n,seed,start", [(5, 1, "id"), (52, 0, "berlin"), (64, 2, "rand"), (200, 3, "id"), (700, 4, "rand"),
                                          (1000, 5, "rand"), (1500, 6, "nn")])
def test_best_sweep_matches_its_oracle(ctx, n, seed, start, tsplib_dir):
    if start == "berlin":
        xy = T.parse_tsplib(os.path.join(tsplib_dir, "berlin52.tsp"))["xy"]
        init = None
    else:
        xy = O.synth_xy(n, seed=seed)
        init = None if start == "id" else (O.restart_perm(n, 9, seed) if start == "rand" else O.nearest_neighbor(xy, None, n, 3)[1])
    g = gpu_two_opt(ctx, xy, None, n, init, mode=1)
    rc, oroute, ocost, ost = O.two_opt(xy, None, n, init=init, best=True)
    assert rc == 0 and g[0].tolist() == oroute.tolist()
    assert np.float32(g[1]).tobytes() == np.float32(ocost).tobytes()
    for k in ("sweeps", "candidates", "moves", "reversed"):
        assert g[2][k] == ost[k], k


def test_best_sweep_full_size_is_a_2opt_local_optimum(ctx):
    # n = 10 000 is far beyond what the O(moves * n^2) oracle finishes; check size-independent properties instead:
    # valid tour, open-path endpoints fixed, strictly shorter than the start, and a fixed point of the reference's sweep.
    n = 10000
    xy = O.synth_xy(n)
    rc, nn, cnn = O.nearest_neighbor(xy, None, n, 3)
    route, cost, st = gpu_two_opt(ctx, xy, None, n, nn, mode=1)
    assert O.validate_tour(route) and route[0] == nn[0] and route[-1] == nn[-1] and cost < cnn
    assert st["sweeps"] == st["moves"] + 1 and st["candidates"] == st["sweeps"] * ((n - 3) * (n - 2) // 2)
    rc, again, c2, st2 = O.two_opt(xy, None, n, init=route, max_candidates=1)
    assert st2["moves"] == 0 and c2 == cost
    assert cost == O.tour_length(xy, None, route)


def test_config5_size_and_lds_limit(ctx):
    # n = 13 509 (usa13509 size; the file is not in the reference tree -> synthetic points, labelled) from the NN seed
    import teeline_amd as TA
    n = 13509
    xy = O.synth_xy(n)
    rc, nn, _ = O.nearest_neighbor(xy, None, n, 3)
    assert_same(gpu_two_opt(ctx, xy, None, n, nn), O.two_opt(xy, None, n, init=nn), n)
    nmax = ctx.two_opt_lds_max_n()
    assert 13509 <= nmax < 16384
    # the largest LDS-resident size (last tile / group logic) and the first size beyond it (HBM-resident tour,
    # scan spread over the chip: two_opt_large.hip) — both must be fixed points of the reference's sweep
    big = O.synth_xy(nmax + 1, seed=2)
    for m in (nmax, nmax + 1):
        xym = big[:m]
        rc, nnm, _ = O.nearest_neighbor(xym, None, m, 3)
        route, cost, st = gpu_two_opt(ctx, xym, None, m, nnm)
        rc, again, c2, st2 = O.two_opt(xym, None, m, init=route, max_candidates=1)
        assert st2["moves"] == 0 and c2 == cost and O.validate_tour(route)
    # (round 5: multi-start / population beyond the limit run through the HBM form — tests/test_gpu_limits.py::test_two_opt_size_limits;
    #  the device-resident batch entry stays LDS-only and says so loudly, never a CPU fallback)
    import ctypes as C
    import torch
    dev = torch.device("cuda", 0)
    m = nmax + 1
    d_xy = torch.from_numpy(np.ascontiguousarray(big[:m])).to(dev)
    d_pos = torch.empty((1, m), dtype=torch.int32, device=dev)
    d_cost = torch.empty(1, dtype=torch.float32, device=dev)
    d_st = torch.zeros((1, 16), dtype=torch.int64, device=dev)
    rc = ctx.lib.tl_two_opt_batch_dev(ctx.handle, d_xy.data_ptr(), m, None, 1, 0, 1, 0, d_pos.data_ptr(), d_cost.data_ptr(), d_st.data_ptr(),
                                      C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == -6  # TL_ERR_UNSUPPORTED


def test_lds_limit_random_start_matches_oracle(ctx):
    # the largest LDS-resident size from a random restart: rows with dozens of deferred reversals, composed flushes that
    # span every register slot (15 x 1024 positions) and all four tile groups — tours, costs and counters as the oracle's
    n = ctx.two_opt_lds_max_n()
    xy = O.synth_xy(n, seed=3)
    rp = O.restart_perm(n, 4242, 1)
    assert_same(gpu_two_opt(ctx, xy, None, n, rp), O.two_opt(xy, None, n, init=rp), n)


def test_large_n_path_matches_oracle(ctx):
    # HBM-resident REF_ORDER path, forced (TL_FLAG_2OPT_FORCE_HBM) on sizes the oracle finishes quickly: identical tours /
    # costs / counters
    import teeline_amd as TA
    with TA.Context(0, TA.TL_FLAG_2OPT_FORCE_HBM) as c2:
        for n, seed in ((3, 1), (4, 1), (5, 2), (64, 3), (65, 4), (700, 5), (3000, 6)):
            xy = O.synth_xy(n, seed=seed)
            assert_same(gpu_two_opt(c2, xy, None, n), O.two_opt(xy, None, n), n)
            if n >= 64:
                rp = O.restart_perm(n, 8, seed)
                assert_same(gpu_two_opt(c2, xy, None, n, rp), O.two_opt(xy, None, n, init=rp), n)
    # n = 20 000 from the NN seed (oracle: ~2.4e9 candidates, too slow for the suite): size-independent checks
    n = 20000
    xy = O.synth_xy(n)
    rc, nn, cnn = O.nearest_neighbor(xy, None, n, 3)
    route, cost, st = gpu_two_opt(ctx, xy, None, n, nn)
    assert O.validate_tour(route) and route[0] == nn[0] and route[-1] == nn[-1] and cost < cnn
    assert st["candidates"] == st["sweeps"] * ((n - 3) * (n - 2) // 2)
    rc, again, c2, st2 = O.two_opt(xy, None, n, init=route, max_candidates=1)
    assert st2["moves"] == 0 and c2 == cost


def test_device_resident_batch_entry_with_explicit_starts(ctx):
    # tl_two_opt_batch_dev with caller-provided device buffers (what bench.py uses), explicit initial tours
    import ctypes as C
    import torch
    from teeline_amd import _capi
    n, R = 500, 5
    xy = O.synth_xy(n, seed=13)
    inits = np.stack([O.restart_perm(n, 21, r) for r in range(R)]).astype(np.int32)
    dev = torch.device("cuda", 0)
    d_xy, d_init = torch.from_numpy(xy).to(dev), torch.from_numpy(inits).to(dev)
    d_pos = torch.empty((R, n), dtype=torch.int32, device=dev)
    d_cost = torch.empty(R, dtype=torch.float32, device=dev)
    d_stats = torch.zeros((R, _capi.TL_DEV_STATS_STRIDE), dtype=torch.int64, device=dev)
    s = torch.cuda.current_stream()
    ctx.check(ctx.lib.tl_two_opt_batch_dev(ctx.handle, d_xy.data_ptr(), n, d_init.data_ptr(), 0, 0, R, 0, d_pos.data_ptr(),
                                           d_cost.data_ptr(), d_stats.data_ptr(), C.c_void_p(s.cuda_stream)))
    torch.cuda.synchronize()
    assert ctx.last_kernel_ms() > 0
    for r in range(R):
        rc, p, c, st = O.two_opt(xy, None, n, init=inits[r].astype(np.uint32))
        assert d_pos[r].cpu().numpy().astype(np.uint32).tolist() == p.tolist()
        assert np.float32(d_cost[r].item()).tobytes() == np.float32(c).tobytes()
        assert (int(d_stats[r, 0]), int(d_stats[r, 1]), int(d_stats[r, 2]), int(d_stats[r, 3])) == (st["sweeps"], st["moves"], st["reversed"], 0)


def test_population_equals_individual_descents(ctx):
    """tl_two_opt_population: every tour of a batch gets exactly the descent tl_two_opt / the oracle gives it alone."""
    import teeline_amd as TA
    n, count = 700, 9
    xy = O.synth_xy(n, seed=4242)
    prob = TA.TspProblem(np.arange(10, 10 + n), xy)  # ids != positions
    inits = [O.restart_perm(n, 99, r) for r in range(count - 1)] + [np.arange(n, dtype=np.uint32)]
    sols = TA.two_opt.solve_population(prob, [(p + 10).tolist() for p in inits], ctx=ctx)
    assert len(sols) == count
    for p, sol in zip(inits, sols):
        rc, route, cost, st = O.two_opt(xy, None, n, init=p)
        assert rc == 0
        assert [int(v) - 10 for v in sol.route()] == route.tolist()
        assert np.float32(sol.total).tobytes() == np.float32(cost).tobytes()
    assert sols[0].stats["moves"] == sum(O.two_opt(xy, None, n, init=p)[3]["moves"] for p in inits)


def test_population_explicit_matrix_and_bad_input(ctx, tsplib_dir):
    import teeline_amd as TA
    prob = TA.tsplib.read_from_file(os.path.join(tsplib_dir, "gr17.tsp")).problem()
    n = len(prob)
    ids = [int(v) for v in prob.ids]
    tours = [ids, ids[::-1], ids[5:] + ids[:5]]
    sols = TA.two_opt.solve_population(prob, tours, ctx=ctx)
    for t, sol in zip(tours, sols):
        one = TA.two_opt.solve(prob, None, None, t, ctx=ctx)
        assert list(sol.route()) == list(one.route()) and float(sol.total) == float(one.total)
    with pytest.raises(TA.TeelineGpuError):
        TA.two_opt.solve_population(prob, [ids, ids[:-1] + [ids[0]]], ctx=ctx)  # second tour repeats a city


@pytest.mark.parametrize("n,restarts_per_cu", [(400, 4.1), (4000, 1.6), (3600, 2.1)])  # 4-wave float2, 8-wave float2, 4-wave grid form
def test_batches_with_several_descents_per_cu_match_the_oracle(ctx, n, restarts_per_cu):
    # More descents than CUs: where the LDS holds four (n <= ~3000) or two (n <= ~7100) tours the batch runs the 4- / 8-wave form of
    # a descent, several per CU.  Same tours as one descent per CU, whatever the form: the costs of ALL restarts against a
    # one-per-CU run, a sample of them (first, last, two in the middle) against the oracle; and the forms forced by flag.
    import teeline_amd as TA
    cus = ctx.device_info()["cus"]
    R, seed = int(cus * restarts_per_cu), 777
    xy = O.synth_xy(n, seed=31)
    prob = TA.TspProblem(np.arange(n), xy)
    sol, costs = TA.two_opt.multistart(prob, R, seed=seed, ctx=ctx, return_costs=True)
    ref = np.concatenate([TA.two_opt.multistart(prob, min(cus, R - f), seed=seed, first=f, ctx=ctx, return_costs=True)[1]
                          for f in range(0, R, cus)])
    assert np.asarray(costs, np.float32).tobytes() == np.asarray(ref, np.float32).tobytes()
    for r in (0, R // 3, R // 2, R - 1):
        rc, p, c, st = O.two_opt(xy, None, n, init=O.restart_perm(n, seed, r))
        assert np.float32(costs[r]).tobytes() == np.float32(c).tobytes()
    for flag in (TA.TL_FLAG_2OPT_NT512, TA.TL_FLAG_2OPT_NT256):
        with TA.Context(0, flag) as c2:
            init = O.restart_perm(n, seed, 5)
            assert_same(gpu_two_opt(c2, xy, None, n, init), O.two_opt(xy, None, n, init=init), n)


def test_grid_coordinate_form_matches_the_oracle_where_the_instance_lies_on_a_grid():
    # TL_FLAG_2OPT_FX: tours kept as 2 x 20-bit grid coordinates (decoded exactly) instead of float2.  Decimal grids of 0..4
    # digits and integer coordinates take the form; arbitrary floats, negative or too large coordinates fall back to float2
    # inside the library — the results must be the oracle's either way.
    import teeline_amd as TA
    rng = np.random.default_rng(5)
    with TA.Context(0, TA.TL_FLAG_2OPT_FX) as c2:
        for n, digits in ((700, 3), (1500, 0), (1200, 1), (900, 4), (2500, 2)):
            xy = (rng.integers(0, 1000000, (n, 2)).astype(np.float32) / np.float32(10.0 ** digits)).astype(np.float32)
            for init in (None, O.restart_perm(n, 3, 1)):
                assert_same(gpu_two_opt(c2, xy, None, n, init), O.two_opt(xy, None, n, init=init), n)
        for xy in (rng.random((800, 2)).astype(np.float32) * 1000, (rng.integers(-500, 500, (600, 2))).astype(np.float32),
                   rng.integers(0, 1 << 22, (500, 2)).astype(np.float32)):
            n = len(xy)
            assert_same(gpu_two_opt(c2, xy, None, n, None), O.two_opt(xy, None, n), n)


def test_pr1002_like_lattice_ties(ctx):
    # pr1002 (BASELINE configs[1]; not in the reference tree) is a drilled-board instance: points on a coarse grid, so equal
    # distances — and candidates whose two sums tie exactly, which the reference's strict `<` (two_opt.rs:49) must leave alone —
    # are everywhere.  A 31 x 32 lattice with some sites dropped (n = 962..992) in shuffled, row-major and NN order; coordinate
    # form, matrix form in HBM and the un-pruned form against the oracle.
    import teeline_amd as TA
    rng = np.random.default_rng(1002)
    g = np.stack(np.meshgrid(np.arange(32, dtype=np.float32) * 100.0, np.arange(31, dtype=np.float32) * 100.0), -1).reshape(-1, 2)
    for drop in (0, 30):
        keep = np.sort(rng.permutation(len(g))[: len(g) - drop])
        pts = np.ascontiguousarray(g[keep])
        n = len(pts)
        packed = O.dm_build_packed(pts)
        rc, nn, _ = O.nearest_neighbor(pts, None, n, 3)
        for init in (None, rng.permutation(n).astype(np.uint32), nn):
            want = O.two_opt(pts, None, n, init=init)
            assert_same(gpu_two_opt(ctx, pts, None, n, init), want, n)
            assert_same(gpu_two_opt(ctx, None, packed, n, init), want, n)
        with TA.Context(0, TA.TL_FLAG_NO_PRUNE) as c2:
            assert_same(gpu_two_opt(c2, pts, None, n, None), O.two_opt(pts, None, n), n)


def _expected_messages(xy, ids, init):
    # two_opt.rs:22-65 restated as its message stream (f32 distances exactly as KDPoint::distance computes them)
    n = len(xy)
    path = list(range(n)) if init is None else [int(v) for v in init]

    def D(p, q):
        dx, dy = xy[p, 0] - xy[q, 0], xy[p, 1] - xy[q, 1]
        return np.sqrt(np.float32(dx * dx) + np.float32(dy * dy), dtype=np.float32)

    msgs = [("PathUpdate", ([int(ids[v]) for v in path], 0.0))]
    improved = True
    while improved:
        improved = False
        for i in range(0, n - 3):
            msgs.append(("CityChange", int(ids[path[i]])))
            for j in range(i + 2, n - 1):
                cur = np.float32(D(path[i], path[i + 1]) + D(path[j], path[j + 1]))
                neu = np.float32(D(path[i], path[j]) + D(path[i + 1], path[j + 1]))
                if neu < cur:
                    path[i + 1:j + 1] = path[i + 1:j + 1][::-1]
                    improved = True
                    msgs.append(("PathUpdate", ([int(ids[v]) for v in path], float(neu))))
    msgs.append(("Done", None))
    return msgs, path


def test_progress_channel_replays_the_reference_messages(ctx, tsplib_dir):
    # VERDICT r02 "missing" 4: per-move progress.  With a progress callback two_opt::solve goes through tl_two_opt_trace and replays
    # PathUpdate(start) / CityChange per outer i / PathUpdate(path, new_distance) per move / Done exactly as two_opt.rs:22-65 sends them.
    import teeline_amd as TA
    d = T.parse_tsplib(os.path.join(tsplib_dir, "berlin52.tsp"))
    xy, n, ids = d["xy"], d["n"], d["ids"]
    rc, nn, _ = O.nearest_neighbor(xy, None, n, 3)
    for init in (None, nn, O.restart_perm(n, 3, 0)):
        got = []
        prob = TA.TspProblem(ids, xy)
        sol = TA.two_opt.solve(prob, None, lambda kind, payload: got.append((kind, payload)),
                               None if init is None else [int(ids[v]) for v in init], ctx=ctx)
        want, path = _expected_messages(xy, ids, init)
        assert got == want
        assert list(sol.route()) == [int(ids[v]) for v in path]


def test_move_list_matches_the_oracle(ctx):
    # tl_two_opt_trace: the applied moves (i, j) in the reference's order with a mark where a new sweep begins, against the oracle's
    # record of the same loop; a short log buffer holds the prefix and reports the full count
    import ctypes as C
    import teeline_amd as TA
    from teeline_amd import _capi
    for n, seed in ((5, 1), (64, 2), (700, 3), (3000, 4), (10000, 5)):
        xy = O.synth_xy(n, seed=seed if n < 10000 else 0)
        init = O.restart_perm(n, 12345, 0)
        rc, route, cost, st, ij, dist, sw = O.two_opt_trace(xy, None, n, init=init)
        words, last = O.trace_words(ij, sw)
        words += [0xFFFFFFFF] * (st["sweeps"] - last)   # the final sweeps without a move are marked too
        want = np.asarray(words, dtype=np.uint32)
        out = np.empty(n, dtype=np.uint32)
        c, stt, ln = C.c_float(), _capi.TlStats(), C.c_uint32()
        for cap in (len(want) + 7, max(len(want) // 3, 1)):
            log = np.full(cap, 0x12345678, dtype=np.uint32)
            ctx.check(ctx.lib.tl_two_opt_trace(ctx.handle, xy.ctypes.data_as(C.c_void_p), n, None, init.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p),
                                               C.byref(c), C.byref(stt), log.ctypes.data_as(C.c_void_p), cap, C.byref(ln)))
            assert ln.value == len(want) == stt.moves + stt.sweeps - 1 and out.tolist() == route.tolist()
            m = min(cap, len(want))
            assert log[:m].tolist() == want[:m].tolist()
            assert (log[m:] == 0x12345678).all()
        assert np.float32(c.value).tobytes() == np.float32(cost).tobytes()


def test_move_list_and_progress_of_the_matrix_form(ctx, tsplib_dir):
    # tl_two_opt_trace with dm_packed: the matrix-form kernel writes the same move list (EXPLICIT / GEO problems and a synthetic matrix
    # at pr1002's size), and two_opt::solve replays the reference's messages with new_distance read from problem.distances
    import ctypes as C
    import teeline_amd as TA
    from teeline_amd import _capi
    cases = []
    for name in ("gr17.tsp", "bays29.tsp", "burma14.tsp"):
        d = T.parse_tsplib(os.path.join(tsplib_dir, name))
        packed = d["packed"] if d["packed"] is not None else O.dm_build_packed(d["xy"], geo=True)  # burma14: GEO
        cases.append((name, d["xy"], d["n"], packed, d["ids"]))
    n = 1002
    xy = O.synth_xy(n)
    cases.append(("synthetic1002", xy, n, O.dm_build_packed(xy), np.arange(n)))
    for name, xy, n, packed, ids in cases:
        packed = np.ascontiguousarray(packed, dtype=np.float32)
        for init in (None, O.restart_perm(n, 7, 0)):
            rc, route, cost, st, ij, dist, sw = O.two_opt_trace(xy, packed, n, init=init)
            words, last = O.trace_words(ij, sw)
            words += [0xFFFFFFFF] * (st["sweeps"] - last)
            want = np.asarray(words, dtype=np.uint32)
            out = np.empty(n, dtype=np.uint32)
            c, stt, ln = C.c_float(), _capi.TlStats(), C.c_uint32()
            cap = len(want) + 5
            log = np.full(cap, 0x12345678, dtype=np.uint32)
            xyc = np.ascontiguousarray(xy, dtype=np.float32)
            ctx.check(ctx.lib.tl_two_opt_trace(ctx.handle, xyc.ctypes.data_as(C.c_void_p), n, packed.ctypes.data_as(C.c_void_p),
                                               None if init is None else init.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p),
                                               C.byref(c), C.byref(stt), log.ctypes.data_as(C.c_void_p), cap, C.byref(ln)))
            assert ln.value == len(want) == stt.moves + stt.sweeps - 1, name
            assert out.tolist() == route.tolist() and log[:len(want)].tolist() == want.tolist(), name
            assert np.float32(c.value).tobytes() == np.float32(cost).tobytes()
            if n <= 64:  # the replayed message stream: every PathUpdate's new_distance is the oracle's record of the compared sum
                got = []
                prob = TA.TspProblem(ids, xyc, TA.distance_matrix.DistanceMatrix(n, packed, ids, "explicit"))
                sol = TA.two_opt.solve(prob, None, lambda kind, payload: got.append((kind, payload)),
                                       None if init is None else [int(ids[v]) for v in init], ctx=ctx)
                upd = [m for m in got if m[0] == "PathUpdate"]
                assert got[0][0] == "PathUpdate" and got[-1] == ("Done", None) and len(upd) == 1 + len(ij)
                assert [np.float32(m[1][1]).tobytes() for m in upd[1:]] == [np.float32(v).tobytes() for v in dist]
                assert upd[-1][1][0] == [int(ids[v]) for v in route] if len(ij) else True
                assert sum(1 for m in got if m[0] == "CityChange") == st["sweeps"] * (n - 3)


def test_progress_at_tiny_sizes(ctx):
    # n = 3 ... 6: the move list and the replayed messages where rows are few (n = 3: no row at all, one sweep; n < 4: 3-opt and Or-opt
    # return before their first message), coordinates and matrix form
    import teeline_amd as TA
    for n in (3, 4, 5, 6):
        for seed in (1, 2, 3):
            xy = O.synth_xy(n, seed=seed)
            packed = O.dm_build_packed(xy)
            init = O.restart_perm(n, seed, 0)
            for form in ("xy", "dm"):
                prob = TA.TspProblem(np.arange(n), xy, None if form == "xy" else TA.distance_matrix.DistanceMatrix(n, packed, np.arange(n), "explicit"))
                got = []
                sol = TA.two_opt.solve(prob, None, lambda k, p: got.append((k, p)), [int(v) for v in init], ctx=ctx)
                rc, route, cost, st, ij, dist, sw = O.two_opt_trace(xy, None if form == "xy" else packed, n, init=init)
                assert list(sol.route()) == route.tolist(), (n, seed, form)
                pu = [m for m in got if m[0] == "PathUpdate"]
                assert len(pu) == 1 + len(ij) and got[-1] == ("Done", None)
                assert sum(1 for m in got if m[0] == "CityChange") == st["sweeps"] * max(n - 3, 0), (n, seed, form, st)
                assert [np.float32(m[1][1]).tobytes() for m in pu[1:]] == [np.float32(v).tobytes() for v in dist]
            for solver in (TA.three_opt, TA.or_opt):
                got = []
                sol = solver.solve(TA.TspProblem(np.arange(n), xy), None, lambda k, p: got.append((k, p)), [int(v) for v in init], ctx=ctx)
                assert (got == []) == (n < 4), (solver.__name__, n, got[:2])
                if n >= 4:
                    assert got[0][0] == "PathUpdate" and got[-1] == ("Done", None) and got[-2][1][0] == list(sol.route())
