"""N > 1 path on CPU: two processes, gloo backend, 127.0.0.1 rendezvous.  Covers the host logic bench.py runs
between ranks — restart sharding, key packing, the min-all-reduce of the best (cost, restart) key and the
whole-job throughput aggregation — with per-shard costs produced by the CPU oracle standing in for the GPU
descents (the GPU kernels themselves are covered by the -m gpu tests)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

WORLD = 2
R_PER_RANK = 3
N = 120
SEED = 77


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, port, outq, total=0):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    sys.path.insert(0, os.path.dirname(here))
    import _oracle as O
    from teeline_amd.host import multistart as ms
    from teeline_amd.host import synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    try:
        xy = synth.synth_xy(N)
        first, count = ms.shard_total(rank, WORLD, total) if total else ms.shard(rank, R_PER_RANK)
        costs, cands, tours = [], 0, []
        for r in range(first, first + count):
            init = synth.restart_perm(N, SEED, r)
            rc, p, c, st = O.two_opt(xy, None, N, init=init)
            costs.append(c)
            tours.append(np.asarray(p, dtype=np.int32))
            cands += st["candidates"]
        keys = ms.pack_keys(torch.tensor(np.asarray(costs, dtype=np.float32)), first)
        best = ms.allreduce_best(keys, dist)
        shared = ms.share_best_tour(keys, torch.tensor(np.stack(tours)), best, dist)
        total, tmax = ms.aggregate_throughput(cands, 1.0 + rank, torch.device("cpu"), dist)
        dist.barrier()
        outq.put((rank, int(best.item()), total, tmax, [float(c) for c in costs], cands, shared.tolist(), [t.tolist() for t in tours]))
    finally:
        dist.destroy_process_group()


def test_two_rank_min_allreduce_of_best_tour_key():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, port, q)) for r in range(WORLD)]
    [p.start() for p in procs]
    res = sorted(q.get(timeout=120) for _ in range(WORLD))
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    from teeline_amd.host import multistart as ms
    all_costs = res[0][4] + res[1][4]
    assert len(all_costs) == WORLD * R_PER_RANK
    # every rank ends with the same key = global (min cost, lowest restart id on ties)
    assert res[0][1] == res[1][1]
    cost, restart = ms.unpack_key(res[0][1])
    want = min(range(len(all_costs)), key=lambda i: (np.float32(all_costs[i]), i))
    assert restart == want and np.float32(cost) == np.float32(all_costs[want])
    # ... and with the winner's tour (one SUM-all-reduce, only the owner contributes)
    all_tours = res[0][7] + res[1][7]
    assert res[0][6] == res[1][6] == all_tours[want]
    # whole-job accounting: candidates are summed, time is the slowest rank's
    assert res[0][2] == res[1][2] == res[0][5] + res[1][5]
    assert res[0][3] == res[1][3] == 2.0


def test_two_rank_strong_shard_map():
    # BASELINE configs[3] as worded: a fixed number of restarts in all (here 5: an uneven split, 3 + 2), dealt in contiguous
    # blocks; the winner is the one a single process finds over restarts 0..4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, port, q, 5)) for r in range(WORLD)]
    [p.start() for p in procs]
    res = sorted(q.get(timeout=120) for _ in range(WORLD))
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    from teeline_amd.host import multistart as ms
    assert [len(r[4]) for r in res] == [3, 2]
    all_costs = res[0][4] + res[1][4]
    want = min(range(5), key=lambda i: (np.float32(all_costs[i]), i))
    assert res[0][1] == res[1][1] and ms.unpack_key(res[0][1])[1] == want
    assert res[0][6] == res[1][6] == (res[0][7] + res[1][7])[want]


def _shard_worker(rank, world, port, outq):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.dirname(here))
    from teeline_amd.host import multistart as ms
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rows = [ms.shard_total(rank, world, total) for total in (256, 255, 10, 7, 1)]
        t = torch.tensor(rows, dtype=torch.int64)
        got = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(got, t)
        if rank == 0:
            outq.put([g.tolist() for g in got])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [1, 2, 3, 4, 8])
def test_library_shard_map_is_the_ranks_shard_map(world):
    # tl_two_opt_multistart_devices (ONE process, n_ctxs devices) deals restarts with tl_multistart_shard —
    # contiguous blocks, the first count % n take one more; the ranks of a multi-process job use multistart.shard_total.  The two
    # maps must be the same map, or a run's winner would depend on how it was spread: compared here for world 1..8, the ranks'
    # side gathered over a real gloo process group of that size (world 1: no group).
    import ctypes as C
    from teeline_amd import _capi, build
    from teeline_amd.host import multistart as ms
    build.build()
    lib = _capi.load()
    totals = (256, 255, 10, 7, 1)

    def lib_map(total, first=0):
        out = []
        for d in range(world):
            f, c = C.c_uint32(), C.c_uint32()
            assert lib.tl_multistart_shard(first, total, world, d, C.byref(f), C.byref(c)) == 0
            out.append([f.value, c.value])
        return out

    if world == 1:
        ranks = [[list(ms.shard_total(0, 1, t)) for t in totals]]
    else:
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_shard_worker, args=(r, world, port, q)) for r in range(world)]
        [p.start() for p in procs]
        ranks = q.get(timeout=180)
        [p.join(timeout=60) for p in procs]
        assert all(p.exitcode == 0 for p in procs)
    for k, total in enumerate(totals):
        assert [ranks[r][k] for r in range(world)] == lib_map(total), (world, total)
    # a shifted first restart shifts every block; bad arguments are refused
    assert lib_map(10, first=100) == [[100 + f, c] for f, c in lib_map(10)]
    f, c = C.c_uint32(), C.c_uint32()
    assert lib.tl_multistart_shard(0, 10, 0, 0, C.byref(f), C.byref(c)) == _capi.TL_ERR_BADARG
    assert lib.tl_multistart_shard(0, 10, 4, 4, C.byref(f), C.byref(c)) == _capi.TL_ERR_BADARG


def test_key_packing_matches_c_abi_definition():
    from teeline_amd.host import multistart as ms
    costs = torch.tensor([3.5, 1.25, 1.25, 77647.55469], dtype=torch.float32)
    keys = ms.pack_keys(costs, 10)
    assert keys.dtype == torch.int64 and int(keys.argmin()) == 1            # tie -> lowest restart id
    assert ms.unpack_key(int(keys[3])) == (float(np.float32(77647.55469)), 13)
    from teeline_amd import _capi, build
    build.build()
    lib = _capi.load()
    for i, c in enumerate(costs.tolist()):
        assert lib.tl_pack_cost_key(c, 10 + i) == int(keys[i])
    assert ms.shard(3, 256) == (768, 256)
    # strong scaling (BASELINE configs[3]: 256 restarts in all over 1/2/4/8 GPUs): contiguous blocks that tile [0, total)
    for world in (1, 2, 3, 4, 8):
        for total in (256, 10, 7):
            blocks = [ms.shard_total(r, world, total) for r in range(world)]
            assert blocks[0][0] == 0 and sum(c for _, c in blocks) == total
            assert all(blocks[r][0] + blocks[r][1] == blocks[r + 1][0] for r in range(world - 1))
            assert max(c for _, c in blocks) - min(c for _, c in blocks) <= 1
    assert ms.shard_total(3, 8, 256) == (96, 32)
    # one process, no process group: the "shared" tour is the local winner's
    tours = torch.arange(4 * 6, dtype=torch.int32).reshape(4, 6)
    best = ms.allreduce_best(keys, None)
    assert int(best) == int(keys[1]) and ms.share_best_tour(keys, tours, best, None).tolist() == tours[1].tolist()
