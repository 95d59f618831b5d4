"""GPU parity tests (-m gpu): the matrix-form 2-opt's sweeps on lists (two_opt_dm.hip, LATE) against the CPU oracle.

A sweep in which few cities have a tour edge beyond their 16th-nearest distance decides a row from a's 16 nearest (as c), b's reverse
list (as e) and the long cities instead of walking the matrix rows.  The tour, the cost bits and the sweep / move / reversal counters
must be the oracle's on every start and every kind of matrix — the lists prune with D[a][c] < D[a][b] or D[b][e] < D[c][e], which
holds for ANY symmetric matrix (no triangle inequality is used), so the cases include random non-metric weights.
"""
import numpy as np
import pytest

import _oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def forms():
    """default thresholds, lists wherever they fit (TL_FLAG_2OPT_NL_ALWAYS), lists off"""
    import teeline_amd as TA
    cs = {"default": TA.Context(0), "always": TA.Context(0, TA.TL_FLAG_2OPT_NL_ALWAYS), "off": TA.Context(0, TA.TL_FLAG_2OPT_NO_NL)}
    yield cs
    for c in cs.values():
        c.close()


def run(ctx, packed, n, init):
    import teeline_amd as TA
    prob = TA.TspProblem(np.arange(n), np.zeros((n, 2), np.float32), TA.distance_matrix.DistanceMatrix(n, packed, np.arange(n), "explicit"))
    sol = TA.two_opt.solve(prob, None, None, None if init is None else [int(v) for v in init], ctx=ctx)
    return (np.asarray(sol.route(), dtype=np.uint32), sol.total, sol.stats), ctx.two_opt_last_counters()


def same(gpu, ora):
    route, cost, st = gpu
    rc, oroute, ocost, ost = ora
    assert rc == 0
    assert route.tolist() == oroute.tolist(), "tour differs from the oracle"
    assert np.float32(cost).tobytes() == np.float32(ocost).tobytes()
    for k in ("sweeps", "candidates", "moves", "reversed"):
        assert st[k] == ost[k], f"{k}: gpu {st[k]} != oracle {ost[k]}"


def random_symmetric_packed(n, seed, kind):
    rng = np.random.default_rng(seed)
    m = n * (n - 1) // 2
    if kind == "uniform":      # non-metric: independent weights
        return rng.random(m, dtype=np.float32) * np.float32(1000.0)
    if kind == "small_ints":   # many exact ties, zeros included
        return rng.integers(0, 12, m).astype(np.float32)
    raise ValueError(kind)


def starts(xy, packed, n):
    nn = O.nearest_neighbor(xy, packed if xy is None else None, n, 3)[1]
    return (("identity", None), ("nn", nn), ("random", O.restart_perm(n, 17, 0)))


@pytest.mark.parametrize("n", [200, 333, 1002])
def test_euclidean_matrix_all_forms_equal_the_oracle(forms, n):
    xy = O.synth_xy(n, seed=31 + n)
    packed = O.dm_build_packed(xy)
    for name, init in starts(xy, packed, n):
        ora = O.two_opt(None, packed, n, init=init)
        used = {}
        for form, ctx in forms.items():
            g, cnt = run(ctx, packed, n, init)
            same(g, ora)
            used[form] = cnt[5:8]
        assert used["off"] == [0, 0, 0]
        assert used["always"][1] >= 1, f"{name}: no sweep ran on the lists"  # (the local optimum's last sweep at the latest)
        if name == "nn" and n >= 500:  # (below n = 500 the library does not build the lists by itself)
            assert used["default"][1] >= 1, "NN start: the default thresholds never took the lists"
        if n < 500:
            assert used["default"] == [0, 0, 0]


@pytest.mark.parametrize("kind,n", [("uniform", 300), ("uniform", 700), ("small_ints", 400)])
def test_non_metric_and_tied_matrices(forms, kind, n):
    packed = random_symmetric_packed(n, 5 + n, kind)
    for name, init in (("identity", None), ("random", O.restart_perm(n, 3, 1))):
        ora = O.two_opt(None, packed, n, init=init)
        for form, ctx in forms.items():
            g, cnt = run(ctx, packed, n, init)
            same(g, ora)


def test_clustered_instance_with_overfull_reverse_lists(forms):
    """A hub: 120 cities within 1e-3 of one point and 400 spread out — every spread city near the hub lists hub cities first, so the hub
    cities' reverse lists exceed their 48 slots and their rows walk the matrix rows; duplicates give zero distances."""
    rng = np.random.default_rng(77)
    hub = np.float32(0.5) + rng.random((120, 2), dtype=np.float32) * np.float32(1e-3)
    hub[10:20] = hub[0]  # exact duplicates
    xy = np.concatenate([hub, rng.random((400, 2), dtype=np.float32)]).astype(np.float32)
    rng.shuffle(xy)
    n = len(xy)
    packed = O.dm_build_packed(xy)
    for name, init in starts(xy, packed, n):
        ora = O.two_opt(None, packed, n, init=init)
        for form, ctx in forms.items():
            g, cnt = run(ctx, packed, n, init)
            same(g, ora)
            if form == "always":
                assert cnt[6] >= 1


def test_lattice_ties_and_nan_rows(forms):
    """Integer lattice (every distance tied many times over) and a matrix with a NaN row: a city whose row holds a NaN has no bound,
    is long for ever, and its own rows walk the matrix — the oracle's comparisons with NaN are all false, like the kernel's."""
    side = 16
    xy = np.array([[x, y] for y in range(side) for x in range(side)], np.float32)
    n = len(xy)
    packed = O.dm_build_packed(xy)
    for name, init in (("identity", None), ("random", O.restart_perm(n, 9, 2))):
        ora = O.two_opt(None, packed, n, init=init)
        for form, ctx in forms.items():
            same(run(ctx, packed, n, init)[0], ora)
    bad = packed.copy()
    r = 37
    for c in range(n):
        if c != r:
            hi, lo = max(r, c), min(r, c)
            bad[hi * (hi - 1) // 2 + lo] = np.float32("nan")
    init = O.restart_perm(n, 9, 3)
    ora = O.two_opt(None, bad, n, init=init)
    for form, ctx in forms.items():
        (route, cost, st), cnt = run(ctx, bad, n, init)
        assert route.tolist() == ora[1].tolist()
        assert st["moves"] == ora[3]["moves"] and st["sweeps"] == ora[3]["sweeps"]


def test_population_of_tours_on_one_matrix(forms):
    """tl_two_opt_population: 24 descents share one matrix and one set of lists"""
    import teeline_amd as TA
    n = 600
    xy = O.synth_xy(n, seed=3)
    packed = O.dm_build_packed(xy)
    prob = TA.TspProblem(np.arange(n), xy, TA.distance_matrix.DistanceMatrix(n, packed, np.arange(n), "explicit"))
    pop = [[int(v) for v in O.restart_perm(n, 11, r)] for r in range(23)] + [[int(v) for v in O.nearest_neighbor(xy, None, n, 3)[1]]]
    for form in ("default", "always"):
        sols = TA.two_opt.solve_population(prob, pop, ctx=forms[form])
        for init, s in zip(pop, sols):
            rc, p, c, st = O.two_opt(None, packed, n, init=np.asarray(init, np.uint32))
            assert list(s.route()) == p.tolist() and np.float32(s.total).tobytes() == np.float32(c).tobytes()


def test_last_counters_before_any_call_and_after_a_coordinate_call():
    """tl_two_opt_last_counters: an error on a context that has not run a 2-opt call yet; after a coordinate-form call the first five
    words are that descent's sweeps, moves, reversed elements, status and steps."""
    import teeline_amd as TA
    with TA.Context(0) as c:
        with pytest.raises(TA.TeelineGpuError):
            c.two_opt_last_counters()
        n = 300
        xy = O.synth_xy(n, seed=2)
        sol = TA.two_opt.solve(TA.TspProblem(np.arange(n), xy), None, None, None, ctx=c)
        cnt = c.two_opt_last_counters()
        assert cnt[0] == sol.stats["sweeps"] and cnt[1] == sol.stats["moves"] and cnt[3] == 0 and cnt[4] > 0
