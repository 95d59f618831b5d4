"""The LDS-resident LK of small instances (-m gpu; round 5): k_lk_ils — one workgroup, every array in LDS, the chain search level by level
with depth-first-order keys — and the SPECULATIVE EPOCHS on top of it (one workgroup per epoch, a batch of consecutive epochs at once,
taken in order up to the first accepted one).  Three forms of the same search: the default for these sizes ("spec"), the epochs one
after the other in one workgroup ("seq": TL_FLAG_LK_NO_SPECULATION) and the chip-wide scans every other size runs ("chip":
TL_FLAG_LK_CHIP_WIDE).  All three against the oracle (lin_kernighan.rs:35-499): tour, cost bits, scans / searches / moves /
exchanged edges, progress messages."""
import ctypes as C
import os

import numpy as np
import pytest

import _oracle as O
import _tsplib as T

pytestmark = pytest.mark.gpu

FORMS = ["spec", "seq", "chip"]


@pytest.fixture(scope="module")
def forms():
    import teeline_amd as TA
    TA._capi.load()
    c = {"spec": TA.Context(0, TA.TL_FLAG_LK_ILS_LDS), "seq": TA.Context(0, TA.TL_FLAG_LK_ILS_LDS | TA.TL_FLAG_LK_NO_SPECULATION),
         "chip": TA.Context(0, TA.TL_FLAG_LK_CHIP_WIDE)}
    yield c
    for v in c.values():
        v.close()


def prob(xy):
    import teeline_amd as TA
    return TA.TspProblem(np.arange(len(xy)), xy)


def gpu_lk(ctx, xy, init=None, seed=1, tx=None, **kw):
    import teeline_amd as TA
    h = TA.HeuristicOptions(epochs=kw.get("epochs", 100), platoo_epochs=kw.get("platoo_epochs", 10), n_nearest=kw.get("n_nearest", 5))
    sol = TA.lin_kernighan.solve(prob(xy), TA.LKOptions(h, kw.get("max_depth", 5)), tx, None if init is None else [int(v) for v in init], ctx=ctx, seed=seed)
    return np.asarray(sol.route(), dtype=np.uint32), sol.total, sol.stats


def assert_same(g, o, what=""):
    route, cost, st = g
    rc, oroute, ocost, ost = o[:4]
    assert rc == 0 and route.tolist() == oroute.tolist(), f"{what}: tour differs from the oracle"
    assert np.float32(cost).tobytes() == np.float32(ocost).tobytes(), what
    assert (st["sweeps"], st["candidates"], st["moves"], st["reversed"]) == (ost["sweeps"], ost["candidates"], ost["moves"], ost["reversed"]), what


def lattice(m, seed):
    g = np.stack(np.meshgrid(np.arange(m, dtype=np.float32), np.arange(m, dtype=np.float32)), -1).reshape(-1, 2)
    return np.ascontiguousarray(g[np.random.default_rng(seed).permutation(len(g))])


def olk(xy, **kw):
    return O.lin_kernighan_trace(xy, cap=4096, **kw)


@pytest.mark.parametrize("form", FORMS)
def test_published_instances_with_the_cli_options(forms, form, tsplib_dir):
    # bench/baseline-solvers.tsv:17-26 (berlin52 0.10-0.22 s, a280 1.1-3.1 s on the reference's laptop); CLI defaults mod.rs:596-613,1321-1325.
    # a280 is also the instance on which a SPECULATIVE epoch meets a kick whose lk_pass cycles (the oracle does not return from it
    # either): that epoch runs out of its scan budget and is never needed, because an earlier epoch of its batch is accepted.
    kw = dict(epochs=10000, platoo_epochs=500, n_nearest=3, max_depth=5, seed=1)
    for name, want in (("berlin52", "7544.36572"), ("a280", None)):
        xy = T.parse_tsplib(os.path.join(tsplib_dir, f"{name}.tsp"))["xy"]
        g = gpu_lk(forms[form], xy, **kw)
        assert_same(g, olk(xy, **kw), f"{name} {form}")
        if want:
            assert f"{float(g[1]):.5f}" == want


@pytest.mark.parametrize("form", FORMS)
def test_shapes_of_the_search(forms, form):
    # candidate-list widths that change the key's digit width (k = 1, 3 -> 2 bits, 5, 7 -> 3, 8, 12 -> 4), depths 1..6, a lattice (ties in
    # every candidate list), a random start (a long first pass), sizes around the 4-wave / 16-wave switch (n = 200 / 201), tiny tours
    ctx = forms[form]
    xy = O.synth_xy(200, seed=4)
    for depth, k in ((1, 5), (2, 3), (3, 8), (6, 5), (5, 1), (4, 7), (3, 12)):
        kw = dict(seed=1, epochs=20, max_depth=depth, n_nearest=k)
        assert_same(gpu_lk(ctx, xy, **kw), olk(xy, **kw), f"depth {depth} k {k} {form}")
    xy = lattice(10, 2)
    assert_same(gpu_lk(ctx, xy, seed=11, epochs=60, platoo_epochs=30), olk(xy, seed=11, epochs=60, platoo_epochs=30), f"lattice {form}")
    xy = O.synth_xy(201, seed=6)
    init = O.restart_perm(201, 5, 0)
    assert_same(gpu_lk(ctx, xy, init=init, seed=2, epochs=15), olk(xy, init=init, seed=2, epochs=15), f"random start {form}")
    for n in (4, 5, 7, 8, 9, 16):  # n < 8: double_bridge returns the tour as it is (:487-489)
        xy = O.synth_xy(n, seed=n)
        for k in (3, 5):
            assert_same(gpu_lk(ctx, xy, seed=3, epochs=12, platoo_epochs=4, n_nearest=k), olk(xy, seed=3, epochs=12, platoo_epochs=4, n_nearest=k), f"n {n} k {k} {form}")
    xy = O.synth_xy(64, seed=2)
    assert_same(gpu_lk(ctx, xy, seed=5, epochs=0), olk(xy, seed=5, epochs=0), f"no epochs {form}")
    assert_same(gpu_lk(ctx, xy, seed=5, epochs=1, platoo_epochs=1), olk(xy, seed=5, epochs=1, platoo_epochs=1), f"one epoch {form}")


@pytest.mark.parametrize("form", ["spec", "seq"])
def test_sizes_beyond_the_default_cutover_and_deep_chains(forms, form):
    # TL_FLAG_LK_ILS_LDS takes the form wherever it fits: n = 1500 (16 waves, chunks of 1024 pairs, queue overflows halve a chunk);
    # max_depth 8 runs the same kernel from the deep build (chains of up to 16 exchanges)
    ctx = forms[form]
    xy = O.synth_xy(1500, seed=9)
    assert_same(gpu_lk(ctx, xy, seed=4, epochs=6), olk(xy, seed=4, epochs=6), f"n 1500 {form}")
    xy = O.synth_xy(150, seed=3)
    assert_same(gpu_lk(ctx, xy, seed=2, epochs=15, n_nearest=2, max_depth=8), olk(xy, seed=2, epochs=15, n_nearest=2, max_depth=8), f"depth 8 {form}")


@pytest.mark.parametrize("form", FORMS)
def test_progress_messages_of_every_form(forms, form):
    # lin_kernighan.rs:71,90: one PathUpdate(best_tour, best_dist) after the first pass and one per accepted epoch, in order — also when
    # the accepted epoch was one of a speculative batch
    import teeline_amd as TA
    xy = O.synth_xy(300, seed=21)
    got = []
    h = TA.HeuristicOptions(epochs=400, platoo_epochs=120, n_nearest=5)
    sol = TA.lin_kernighan.solve(prob(xy), TA.LKOptions(h, 5), lambda k, p: got.append((k, p)), None, ctx=forms[form], seed=6)
    rc, oroute, ocost, ost, osnaps = O.lin_kernighan_trace(xy, epochs=400, platoo_epochs=120, seed=6, cap=1024)
    assert rc == 0 and len(osnaps) > 8
    assert [k for k, _ in got] == ["PathUpdate"] * len(osnaps)
    for (_, (route, dist)), (spos, sdist) in zip(got, osnaps):
        assert route == spos.tolist() and np.float32(dist).tobytes() == np.float32(sdist).tobytes()
    assert list(sol.route()) == oroute.tolist()


def _cycling_start(tsplib_dir):
    """a280: the best tour after 10 epochs (CLI options, seed 1), kicked with the draws of epoch 170 — a tour from which the reference's
    lk_pass does not terminate (found by the speculative epochs; the oracle does not return from it within minutes)."""
    xy = T.parse_tsplib(os.path.join(tsplib_dir, "a280.tsp"))["xy"]
    n = len(xy)
    rc, best, cost, st, snaps = O.lin_kernighan_trace(xy, epochs=10, platoo_epochs=500, n_nearest=3, max_depth=5, seed=1, cap=64)
    assert rc == 0 and f"{float(cost):.5f}" == "2887.68677"
    M = (1 << 64) - 1

    def splitmix_at(seed, k):
        z = (seed + (k + 1) * 0x9E3779B97F4A7C15) & M
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
        return z ^ (z >> 31)

    r = [splitmix_at(1, 3 * 170 + q) % (n // 4) for q in range(3)]
    # (splitmix64 is counter-based: the stream of seed + 510 * gamma IS the stream of seed 1 from draw 510 = 3 * 170 on)
    return xy, O.double_bridge(best, r[0], r[1], r[2]), best, (1 + 510 * 0x9E3779B97F4A7C15) & M


@pytest.mark.parametrize("form", FORMS)
def test_a_cycling_lk_pass_is_reported_not_waited_for(forms, form, tsplib_dir):
    # The reference's `loop { find_lk_move ... apply }` (lin_kernighan.rs:468-478) never ends on this start.  Every form counts the
    # moves of a pass and gives up beyond 64 n + 4096 with TL_ERR_NO_CONVERGE; the context stays usable.
    import teeline_amd as TA
    xy, start, best10, seed170 = _cycling_start(tsplib_dir)
    ctx = forms[form]
    with pytest.raises(TA.TeelineGpuError) as e:   # the FIRST pass cycles
        gpu_lk(ctx, xy, init=start, seed=1, epochs=0, n_nearest=3)
    assert e.value.code == TA._capi.TL_ERR_NO_CONVERGE
    # ... and an EPOCH's pass: from the locally optimal tour the first pass moves nothing, epoch 0's kick (the seed whose first three
    # draws are those of epoch 170) starts the cycle.  In the speculative form this is the case of an unfinished epoch that IS the next
    # one: it is run again with 8 x the scan budget until the budget is the cap.
    with pytest.raises(TA.TeelineGpuError) as e:
        gpu_lk(ctx, xy, init=best10, seed=seed170, epochs=50, platoo_epochs=50, n_nearest=3)
    assert e.value.code == TA._capi.TL_ERR_NO_CONVERGE
    small = O.synth_xy(60, seed=1)
    assert_same(gpu_lk(ctx, small, seed=1, epochs=5), olk(small, seed=1, epochs=5), "after the error")


def test_the_default_context_takes_the_lds_form_for_small_instances(ctx, forms):
    # n <= 700 by default; up to 2000 when epochs and plateau are >= 64 (the speculative epochs fill the chip).  Read off the kernel time:
    # berlin52 with the CLI's options is ~3 ms in the LDS form and ~50 ms chip-wide.
    xy = O.synth_xy(52, seed=77)
    kw = dict(epochs=3000, platoo_epochs=300, n_nearest=3, seed=1)
    d = gpu_lk(ctx, xy, **kw)
    c = gpu_lk(forms["chip"], xy, **kw)
    assert d[0].tolist() == c[0].tolist() and d[2]["sweeps"] == c[2]["sweeps"]
    assert d[2]["kernel_ms"] < 0.5 * c[2]["kernel_ms"], (d[2]["kernel_ms"], c[2]["kernel_ms"])
