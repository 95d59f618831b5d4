"""Full-size parity of BASELINE configs[3] and configs[4] (-m gpu), against tests/golden/goldens_large.json (made by
tests/golden/make_goldens_large.py with the CPU oracle; the oracle needs minutes for these sizes, the GPU milliseconds).

  configs[3]  synthetic n = 10 000, seeded restarts 0..7 through the device-resident batch entry bench.py times
              (tl_two_opt_batch_dev, Fisher-Yates drawn on the device), and through tl_two_opt_multistart
  configs[4]  synthetic n = 13 509: k-NN lists (k = 5), NN seed, nn -> 2-opt, and Lin-Kernighan (n_nearest 5, depth 5)
Reference: two_opt.rs:26-61, lin_kernighan.rs:12-27,35-100, nearest_neighbor.rs:8-76."""
import json
import os
import zlib

import numpy as np
import pytest

import _oracle as O

pytestmark = pytest.mark.gpu


def f5(x):
    return f"{float(x):.5f}"


def crc(a):
    return int(zlib.crc32(np.ascontiguousarray(a, dtype="<u4").tobytes()))


@pytest.fixture(scope="module")
def large(golden_dir):
    with open(os.path.join(golden_dir, "goldens_large.json")) as fh:
        return json.load(fh)


def test_config3_seeded_batch_n10000_device_entry(ctx, large):
    import torch
    from teeline_amd import _capi
    g = large["synthetic10000_seed12345"]
    n, R = g["n"], len(g["restarts"])
    xy = O.synth_xy(n)
    dev = torch.device("cuda", 0)
    d_xy = torch.from_numpy(xy).to(dev)
    d_pos = torch.empty((R, n), dtype=torch.int32, device=dev)
    d_cost = torch.empty(R, dtype=torch.float32, device=dev)
    d_stats = torch.zeros((R, _capi.TL_DEV_STATS_STRIDE), dtype=torch.int64, device=dev)
    s = torch.cuda.current_stream()
    import ctypes as C
    ctx.check(ctx.lib.tl_two_opt_batch_dev(ctx.handle, d_xy.data_ptr(), n, None, 12345, 0, R, _capi.TL_MODE_REF_ORDER,
                                           d_pos.data_ptr(), d_cost.data_ptr(), d_stats.data_ptr(), C.c_void_p(s.cuda_stream)))
    torch.cuda.synchronize()
    pos, cost, st = d_pos.cpu().numpy().astype(np.uint32), d_cost.cpu().numpy(), d_stats.cpu().numpy()
    per_sweep = (n - 3) * (n - 2) // 2
    for r in range(R):
        e = g["restarts"][str(r)]
        assert f5(cost[r]) == e["cost"], (r, f5(cost[r]), e["cost"])
        assert crc(pos[r]) == e["route_crc32"], r
        assert int(st[r, 0]) == e["stats"]["sweeps"] and int(st[r, 1]) == e["stats"]["moves"]
        assert int(st[r, 2]) == e["stats"]["reversed"] and int(st[r, 3]) == 0
        assert int(st[r, 0]) * per_sweep == e["stats"]["candidates"]
        assert sorted(pos[r].tolist()) == list(range(n))
    # the first restart's start permutation is the specified Fisher-Yates stream (oracle tlo_restart_perm)
    assert crc(O.restart_perm(n, 12345, 0)) == g["restarts"]["0"]["init_crc32"]


def test_config3_two_descents_per_cu_grid_form_n10000(ctx, large):
    # More restarts than CUs at n = 10^4: the batch runs the grid-coordinate form (2 x 20-bit coordinates per city, decoded
    # exactly — checked by the library for every coordinate of the instance), two descents per CU.  Restarts 0..7 against the
    # committed goldens, every restart against a one-descent-per-CU run of the float2 form.
    import ctypes as C
    import torch
    from teeline_amd import _capi
    g = large["synthetic10000_seed12345"]
    n = g["n"]
    cus = ctx.device_info()["cus"]
    R = 2 * cus + 8
    xy = O.synth_xy(n)
    dev = torch.device("cuda", 0)
    d_xy = torch.from_numpy(xy).to(dev)
    s = torch.cuda.current_stream()

    def run(first, count):
        d_pos = torch.empty((count, n), dtype=torch.int32, device=dev)
        d_cost = torch.empty(count, dtype=torch.float32, device=dev)
        d_stats = torch.zeros((count, _capi.TL_DEV_STATS_STRIDE), dtype=torch.int64, device=dev)
        ctx.check(ctx.lib.tl_two_opt_batch_dev(ctx.handle, d_xy.data_ptr(), n, None, 12345, first, count, _capi.TL_MODE_REF_ORDER,
                                               d_pos.data_ptr(), d_cost.data_ptr(), d_stats.data_ptr(), C.c_void_p(s.cuda_stream)))
        torch.cuda.synchronize()
        return d_pos.cpu().numpy().astype(np.uint32), d_cost.cpu().numpy(), d_stats.cpu().numpy(), ctx.last_kernel_ms()

    pos, cost, st, ms_two = run(0, R)
    for r in range(8):
        e = g["restarts"][str(r)]
        assert f5(cost[r]) == e["cost"] and crc(pos[r]) == e["route_crc32"]
        assert int(st[r, 0]) == e["stats"]["sweeps"] and int(st[r, 1]) == e["stats"]["moves"] and int(st[r, 2]) == e["stats"]["reversed"]
    ms_one = 0.0
    for first in range(0, R, cus):
        p1, c1, s1, ms = run(first, min(cus, R - first))
        ms_one += ms
        k = len(c1)
        assert c1.tobytes() == cost[first:first + k].tobytes()
        assert np.array_equal(p1, pos[first:first + k]) and np.array_equal(s1[:, :4], st[first:first + k, :4])
    assert ms_two < ms_one, (ms_two, ms_one)  # what the form is for


def test_config3_multistart_entry_n10000(ctx, large):
    import teeline_amd as TA
    g = large["synthetic10000_seed12345"]
    n = g["n"]
    xy = O.synth_xy(n)
    prob = TA.TspProblem(np.arange(n), xy)
    sol, costs = TA.two_opt.multistart(prob, 8, seed=12345, first=0, ctx=ctx, return_costs=True)
    want = [g["restarts"][str(r)] for r in range(8)]
    assert [f5(c) for c in costs] == [e["cost"] for e in want]
    best = min(range(8), key=lambda r: (np.float32(want[r]["cost"]), r))
    assert sol.stats["best_restart"] == best and f5(sol.total) == want[best]["cost"]
    assert crc(np.asarray(sol.route(), dtype=np.uint32)) == want[best]["route_crc32"]
    assert sol.stats["candidates"] == sum(e["stats"]["candidates"] for e in want)
    # a shard that does not start at restart 0 (what rank 1 of a 2-GPU strong run owns)
    sol2, costs2 = TA.two_opt.multistart(prob, 4, seed=12345, first=4, ctx=ctx, return_costs=True)
    assert [f5(c) for c in costs2] == [e["cost"] for e in want[4:]]


def test_config4_knn_nn_and_two_opt_n13509(ctx, large):
    import teeline_amd as TA
    g = large["synthetic13509"]
    n = g["n"]
    xy = O.synth_xy(n)
    assert int(zlib.crc32(xy.tobytes())) == g["xy_crc32"]
    prob = TA.TspProblem(np.arange(n), xy)
    cand = TA.lin_kernighan.build_candidates(prob, 5, ctx=ctx)
    assert cand.shape == (n, 5) and crc(cand) == g["knn_k5"]["crc32"] and cand[:3].tolist() == g["knn_k5"]["head"]
    nn = TA.nearest_neighbor.solve(prob, ctx=ctx)
    assert f5(nn.total) == g["nn"]["cost"] and crc(np.asarray(nn.route(), dtype=np.uint32)) == g["nn"]["route_crc32"]
    sol = TA.two_opt.solve(prob, None, None, nn.route(), ctx=ctx)
    e = g["nn_two_opt"]
    assert f5(sol.total) == e["cost"] and crc(np.asarray(sol.route(), dtype=np.uint32)) == e["route_crc32"]
    assert {k: sol.stats[k] for k in ("sweeps", "candidates", "moves", "reversed")} == e["stats"]


def test_config4_lin_kernighan_n13509(ctx, large):
    import teeline_amd as TA
    g = large["synthetic13509"]
    n = g["n"]
    xy = O.synth_xy(n)
    prob = TA.TspProblem(np.arange(n), xy)
    e = g["lk_epochs2_seed7_from_nn"]
    o = e["opts"]
    nn = TA.nearest_neighbor.solve(prob, ctx=ctx)
    h = TA.HeuristicOptions(epochs=o["epochs"], platoo_epochs=o["platoo_epochs"], n_nearest=o["n_nearest"])
    sol = TA.lin_kernighan.solve(prob, TA.LKOptions(h, o["max_depth"]), None, nn.route(), ctx=ctx, seed=o["seed"])
    route = np.asarray(sol.route(), dtype=np.uint32)
    assert sorted(route.tolist()) == list(range(n))
    assert f5(sol.total) == e["cost"], (f5(sol.total), e["cost"])
    assert crc(route) == e["route_crc32"]
    assert {k: sol.stats[k] for k in ("sweeps", "candidates", "moves", "reversed")} == e["stats"]
    # and the default seeding path (init_tour = None -> NN seed inside tl_lk, lin_kernighan.rs:47-55) gives the same run
    sol2 = TA.lin_kernighan.solve(prob, TA.LKOptions(h, o["max_depth"]), None, None, ctx=ctx, seed=o["seed"])
    assert f5(sol2.total) == e["cost"] and crc(np.asarray(sol2.route(), dtype=np.uint32)) == e["route_crc32"]
