"""The late phase of the LDS 2-opt descent (neighbour-list rows, teeline_amd/csrc/two_opt_nl.hip): the lists against a numpy
restatement, and the descent with the lists in use from the second sweep on (TL_FLAG_2OPT_NL_ALWAYS) against the oracle.
Reference: src/tsp/two_opt.rs:26-61 — the lists only change WHICH candidates a row looks at, never a decision."""
import ctypes as C

import numpy as np
import pytest

import _oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nl_ctx():
    import teeline_amd as TA
    c = TA.Context(0, TA.TL_FLAG_2OPT_NL_ALWAYS)
    yield c
    c.close()


def sq_bits(xy, u):
    """bits of fl(fl(dx*dx) + fl(dy*dy)) from city u to every city, the kernels' sqdist (tl_device.h)"""
    d = xy[u][None, :] - xy
    d = (d * d).astype(np.float32)
    return (d[:, 0] + d[:, 1]).astype(np.float32).view(np.uint32).astype(np.int64)


def lists(ctx, xy, form=0):
    n = len(xy)
    ka, kb, rb = C.c_uint32(), C.c_uint32(), C.c_uint32()
    rec = np.empty((n, 64), np.uint16)
    dkb2 = np.empty(n, np.uint32)
    knn_b = np.empty((n, 24), np.uint16)
    rcnt = np.empty(n, np.uint32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    ctx.check(ctx.lib.tl_two_opt_neighbour_lists(ctx.handle, p(xy), n, form, p(rec), p(dkb2), p(knn_b), p(rcnt), C.byref(ka), C.byref(kb), C.byref(rb)))
    assert (ka.value, kb.value, rb.value) == (16, 24, 36)
    return rec, dkb2, knn_b, rcnt


@pytest.mark.parametrize("form", [0, 1])
@pytest.mark.parametrize("kind,n", [("uniform", 3000), ("lattice", 900), ("clusters", 1500), ("duplicates", 400), ("two_points", 700), ("small", 27)])
def test_neighbour_lists_match_numpy(ctx, kind, n, form):
    rng = np.random.default_rng(5)
    if kind == "two_points":  # hundreds of cities at one distance: the wave kernel's buffer overflows, its bisection path runs
        xy = np.array([[0.0, 0.0], [3.0, 4.0]])[rng.integers(0, 2, n)]
    elif kind == "small":
        xy = rng.random((n, 2)) * 10
    elif kind == "uniform":
        xy = rng.random((n, 2)) * 1000
    elif kind == "lattice":  # ties at every distance
        xy = np.stack(np.meshgrid(np.arange(30), np.arange(30)), -1).reshape(-1, 2)[rng.permutation(900)]
    elif kind == "clusters":  # reverse lists far beyond their 36 slots
        c = rng.random((5, 2)) * 1000
        xy = c[rng.integers(0, 5, n)] + rng.normal(0, 1.0, (n, 2))
    else:  # many cities at the same point
        xy = rng.integers(0, 6, (n, 2))
    xy = np.ascontiguousarray(xy, dtype=np.float32)
    rec, dkb2, knn_b, rcnt = lists(ctx, xy, form)
    want_cnt = np.zeros(n, np.int64)
    for u in range(n):
        d = sq_bits(xy, u)
        d[u] = 1 << 40
        srt = np.sort(d)
        # the KB nearest: everything strictly closer than the KB-th distance, the rest of the list at exactly that distance
        assert int(dkb2[u]) == srt[23]
        lb = knn_b[u].astype(np.int64)
        assert len(set(lb.tolist())) == 24 and u not in lb
        assert set(np.nonzero(d < srt[23])[0].tolist()) <= set(lb.tolist()) and (d[lb] <= srt[23]).all()
        np.add.at(want_cnt, lb, 1)
        la = rec[u, 4:20].astype(np.int64)
        assert len(set(la.tolist())) == 16 and u not in la
        assert set(np.nonzero(d < srt[15])[0].tolist()) <= set(la.tolist()) and (d[la] <= srt[15]).all()
        assert int(rec[u, 0]) == srt[15] >> 16
    assert rcnt.astype(np.int64).tolist() == want_cnt.tolist()
    rev = [set() for _ in range(n)]
    for u in range(n):
        for v in knn_b[u]:
            rev[int(v)].add(u)
    for v in range(n):
        got = [int(x) for x in rec[v, 20:56] if x != 0xFFFF]
        assert len(got) == min(len(rev[v]), 36) and len(set(got)) == len(got) and set(got) <= rev[v]
        assert int(rec[v, 2]) == (1 if len(rev[v]) > 36 else 0)
    if kind == "clusters":
        assert (rec[:, 2] == 1).any()


def solve(ctx, xy, init):
    import teeline_amd as TA
    n = len(xy)
    sol = TA.two_opt.solve(TA.TspProblem(np.arange(n), xy), None, None, None if init is None else [int(v) for v in init], ctx=ctx)
    return np.asarray(sol.route(), dtype=np.uint32), sol.total, sol.stats


def same(gpu, ora):
    route, cost, st = gpu
    rc, oroute, ocost, ost = ora
    assert rc == 0 and route.tolist() == oroute.tolist(), "tour differs from the oracle"
    assert np.float32(cost).tobytes() == np.float32(ocost).tobytes()
    for k in ("sweeps", "candidates", "moves", "reversed"):
        assert st[k] == ost[k], f"{k}: gpu {st[k]} != oracle {ost[k]}"


@pytest.mark.parametrize("n,seed", [(27, 3), (64, 1), (257, 2), (1002, 0), (2500, 7), (4097, 4)])
def test_late_phase_from_the_second_sweep_on(nl_ctx, n, seed):
    xy = O.synth_xy(n, seed=seed)
    rp = O.restart_perm(n, 777, seed)
    same(solve(nl_ctx, xy, rp), O.two_opt(xy, None, n, init=rp))
    rc, nn, _ = O.nearest_neighbor(xy, None, n, 3)
    same(solve(nl_ctx, xy, nn), O.two_opt(xy, None, n, init=nn))
    same(solve(nl_ctx, xy, None), O.two_opt(xy, None, n, init=None))


def test_late_phase_on_ties_duplicates_and_clusters(nl_ctx):
    rng = np.random.default_rng(11)
    # a lattice (every decision a near-tie), a handful of distinct points (most rows are zero-length edges), tight clusters far
    # apart (reverse lists overflow, long edges between the clusters: rows fall back to tiles, the long list overflows)
    lat = np.stack(np.meshgrid(np.arange(33), np.arange(31)), -1).reshape(-1, 2).astype(np.float32)
    lat = np.ascontiguousarray(lat[rng.permutation(len(lat))])
    dup = np.ascontiguousarray(rng.integers(0, 7, (600, 2)).astype(np.float32))
    c = rng.random((6, 2)) * 10000
    clu = np.ascontiguousarray((c[rng.integers(0, 6, 2000)] + rng.normal(0, 1.0, (2000, 2))).astype(np.float32))
    line = np.sort(rng.random(800)).astype(np.float32) * 1000
    col = np.ascontiguousarray(np.stack([line, 0.25 * line], 1).astype(np.float32))
    for xy in (lat, dup, clu, col):
        n = len(xy)
        for init in (None, O.restart_perm(n, 5, 1)):
            same(solve(nl_ctx, xy, init), O.two_opt(xy, None, n, init=init))


def test_late_phase_move_list(nl_ctx):
    # tl_two_opt_trace through the late phase: every move and every sweep mark in the reference's order
    import teeline_amd as TA
    from teeline_amd import _capi
    for n, seed in ((700, 3), (3000, 4)):
        xy = O.synth_xy(n, seed=seed)
        init = O.restart_perm(n, 12345, 0)
        rc, route, cost, st, ij, dist, sw = O.two_opt_trace(xy, None, n, init=init)
        words, last = O.trace_words(ij, sw)
        words += [0xFFFFFFFF] * (st["sweeps"] - last)
        want = np.asarray(words, dtype=np.uint32)
        out = np.empty(n, dtype=np.uint32)
        c, stt, ln = C.c_float(), _capi.TlStats(), C.c_uint32()
        log = np.zeros(len(want) + 3, dtype=np.uint32)
        nl_ctx.check(nl_ctx.lib.tl_two_opt_trace(nl_ctx.handle, xy.ctypes.data_as(C.c_void_p), n, None, init.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p),
                                                 C.byref(c), C.byref(stt), log.ctypes.data_as(C.c_void_p), len(log), C.byref(ln)))
        assert ln.value == len(want) and out.tolist() == route.tolist() and log[:len(want)].tolist() == want.tolist()


def test_default_late_phase_equals_the_tile_only_kernel_on_a_batch(ctx):
    # n = 10^4, eight seeded restarts: the default context (late phase from the fifth sweep on) against TL_FLAG_2OPT_NO_NL
    import torch
    import teeline_amd as TA
    n, R = 10000, 8
    xy = O.synth_xy(n)
    dev = torch.device("cuda", 0)
    d_xy = torch.from_numpy(xy).to(dev)
    s = torch.cuda.current_stream()
    res = []
    with TA.Context(0, TA.TL_FLAG_2OPT_NO_NL) as plain:
        for c in (ctx, plain):
            d_pos = torch.empty((R, n), dtype=torch.int32, device=dev)
            d_cost = torch.empty(R, dtype=torch.float32, device=dev)
            d_st = torch.zeros((R, 16), dtype=torch.int64, device=dev)
            c.check(c.lib.tl_two_opt_batch_dev(c.handle, d_xy.data_ptr(), n, None, 12345, 100, R, 0, d_pos.data_ptr(), d_cost.data_ptr(), d_st.data_ptr(), C.c_void_p(s.cuda_stream)))
            torch.cuda.synchronize()
            res.append((d_pos.cpu().numpy(), d_cost.cpu().numpy().view(np.uint32), d_st.cpu().numpy()))
    (pa, ca, sa), (pb, cb, sb) = res
    assert (pa == pb).all() and (ca == cb).all() and (sa[:, :4] == sb[:, :4]).all()
    assert (sa[:, 14] >> 32).min() > 0 and (sb[:, 14] >> 32).max() == 0  # the default really went through the late phase


def test_largest_instance_with_a_late_phase_and_the_first_without(ctx):
    # the late phase needs 2 more bytes per city (+ 4 KB) beside the tour: n = 12 416 is the last size it fits one CU's LDS at;
    # one city more runs the tile-only kernel.  NN start (a few thousand moves), against the oracle.
    import torch
    dev = torch.device("cuda", 0)
    s = torch.cuda.current_stream()
    for n, late in ((12416, True), (12417, False)):
        xy = O.synth_xy(n, seed=3)
        rc, nn, _ = O.nearest_neighbor(xy, None, n, 3)
        d_xy = torch.from_numpy(xy).to(dev)
        d_init = torch.from_numpy(nn.astype(np.int32)).to(dev)
        d_pos = torch.empty((1, n), dtype=torch.int32, device=dev)
        d_cost = torch.empty(1, dtype=torch.float32, device=dev)
        d_st = torch.zeros((1, 16), dtype=torch.int64, device=dev)
        ctx.check(ctx.lib.tl_two_opt_batch_dev(ctx.handle, d_xy.data_ptr(), n, C.c_void_p(d_init.data_ptr()), 0, 0, 1, 0, d_pos.data_ptr(), d_cost.data_ptr(), d_st.data_ptr(),
                                               C.c_void_p(s.cuda_stream)))
        torch.cuda.synchronize()
        st = d_st.cpu().numpy()[0]
        rc, route, cost, ost = O.two_opt(xy, None, n, init=nn)
        assert d_pos.cpu().numpy()[0].astype(np.uint32).tolist() == route.tolist()
        assert d_cost.cpu().numpy().view(np.uint32)[0] == np.float32(cost).view(np.uint32)
        assert (int(st[0]), int(st[1]), int(st[2])) == (ost["sweeps"], ost["moves"], ost["reversed"])
        assert ((int(st[14]) >> 32) > 0) == late


def test_async_batches_on_two_streams_of_one_context_hand_the_lists_over(ctx):
    """the neighbour lists live in ONE per-context buffer, tl_two_opt_batch_dev is asynchronous and takes the caller's
    stream.  Two instances, alternately, on two streams of ONE context, nothing synchronised in between: every call rebuilds the lists
    for its coordinates while the previous call's descents may still be reading them — unless the library orders the rebuild behind
    them on the device (ev_ws).  Every descent of every call must equal the oracle's."""
    import torch
    dev = torch.device("cuda", 0)
    n, R, rounds = 3000, 6, 4
    inst = [O.synth_xy(n, seed=21), O.synth_xy(n, seed=22)]
    d_xy = [torch.from_numpy(x).to(dev) for x in inst]
    streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
    torch.cuda.synchronize()
    outs = []
    for k in range(rounds):
        for w in (0, 1):
            d_pos = torch.empty((R, n), dtype=torch.int32, device=dev)
            d_cost = torch.empty(R, dtype=torch.float32, device=dev)
            d_st = torch.zeros((R, 16), dtype=torch.int64, device=dev)
            with torch.cuda.stream(streams[w]):
                ctx.check(ctx.lib.tl_two_opt_batch_dev(ctx.handle, d_xy[w].data_ptr(), n, None, 777, 10 * k, R, 0, d_pos.data_ptr(), d_cost.data_ptr(),
                                                       d_st.data_ptr(), C.c_void_p(streams[w].cuda_stream)))
            outs.append((w, k, d_pos, d_cost, d_st))
    torch.cuda.synchronize()
    want = {}
    for w, k, d_pos, d_cost, d_st in outs:
        pos, cost, st = d_pos.cpu().numpy(), d_cost.cpu().numpy(), d_st.cpu().numpy()
        assert (st[:, 14] >> 32).min() > 0, "the descents should have gone through the late phase (the lists were read)"
        for r in range(R):
            key = (w, 10 * k + r)
            if key not in want:
                rc, route, c, ost = O.two_opt(inst[w], None, n, init=O.restart_perm(n, 777, 10 * k + r))
                want[key] = (route.tolist(), np.float32(c).tobytes(), ost["sweeps"], ost["moves"])
            route, cbytes, sw, mv = want[key]
            assert pos[r].astype(np.uint32).tolist() == route, (w, k, r)
            assert np.float32(cost[r]).tobytes() == cbytes and (int(st[r, 0]), int(st[r, 1])) == (sw, mv), (w, k, r)
