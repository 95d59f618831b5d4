"""Re-entrancy campaign: T host threads, each with its OWN tl_ctx, run a mix of the library's entry points concurrently for a
given time — 2-opt at n = 52 / 1 002 / 4 097 / 10 000 (so that launches with different LDS sizes of the same kernel overlap:
csrc/tl_kernels.h allow_max_lds), the matrix form, 3-opt, Or-opt, Lin-Kernighan (hipGraph capture on one stream while the others
allocate and launch), the NN seed, candidate lists, tl_dm_build, multi-start — and every result is compared bit for bit with what
the CPU oracle gave for the same input before the threads started.  Then two threads share ONE context: every call must come back
either right or TL_ERR_BUSY.

The reference calls its solvers from arbitrary threads (teeline-api/src/services/tsp_service.rs:295,328 spawn_blocking;
teeline-qt/src/solver_engine.rs:412-432 worker thread); SURVEY.md §8(b): "contexts are independent".

    python tests/probes/thread_campaign.py [seconds] [threads]          (TEELINE_GPU_LIB selects the library, e.g. the jitter build)
"""
import json
import os
import sys
import threading
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import _oracle as O  # noqa: E402
import _tsplib as T  # noqa: E402
import teeline_amd as TA  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
nthreads = int(sys.argv[2]) if len(sys.argv) > 2 else 8
TSPLIB = os.path.join(ROOT, "tests", "golden", "tsplib")


def bits(x):
    return np.float32(x).tobytes()


def crc(a):
    return int(zlib.crc32(np.ascontiguousarray(a, dtype="<u4").tobytes()))


def prob(xy, packed=None):
    n = len(xy)
    if packed is None:
        return TA.TspProblem(np.arange(n), xy)
    return TA.TspProblem(np.arange(n), xy, TA.distance_matrix.DistanceMatrix(n, packed, np.arange(n), "explicit"))


def sol_key(sol):
    return (crc(np.asarray(sol.route(), dtype=np.uint32)), bits(sol.total), sol.stats["sweeps"], sol.stats["moves"])


def okey(o):
    rc, route, cost, st = o
    assert rc == 0
    return (crc(route), bits(cost), st["sweeps"], st["moves"])


# ---- the jobs: (name, callable(ctx) -> comparable, expected) — expectations from the oracle (the n = 10^4 one from the committed golden)
jobs = []
berlin = T.parse_tsplib(os.path.join(TSPLIB, "berlin52.tsp"))["xy"]
xy1k, xy4k, xy10k, xy2k, xy5k = O.synth_xy(1002), O.synth_xy(4097, seed=3), O.synth_xy(10000), O.synth_xy(2000, seed=6), O.synth_xy(5000, seed=2)
t_or = time.time()

jobs.append(("2opt n=52", lambda c: sol_key(TA.two_opt.solve(prob(berlin), ctx=c)), okey(O.two_opt(berlin, None, 52))))
nn1k = O.nearest_neighbor(xy1k, None, 1002, 3)[1]
jobs.append(("2opt n=1002 nn", lambda c: sol_key(TA.two_opt.solve(prob(xy1k), None, None, [int(v) for v in nn1k], ctx=c)),
             okey(O.two_opt(xy1k, None, 1002, init=nn1k))))
p4k = O.restart_perm(4097, 77, 0)
jobs.append(("2opt n=4097 random", lambda c: sol_key(TA.two_opt.solve(prob(xy4k), None, None, [int(v) for v in p4k], ctx=c)),
             okey(O.two_opt(xy4k, None, 4097, init=p4k))))
gl = json.load(open(os.path.join(ROOT, "tests", "golden", "goldens_large.json")))["synthetic10000_seed12345"]["restarts"]["1"]
p10k = O.restart_perm(10000, 12345, 1)


def two_opt_10k(c):
    s = TA.two_opt.solve(prob(xy10k), None, None, [int(v) for v in p10k], ctx=c)
    return (crc(np.asarray(s.route(), dtype=np.uint32)), f"{float(s.total):.5f}", s.stats["sweeps"], s.stats["moves"])


jobs.append(("2opt n=10000 restart 1 (golden)", two_opt_10k, (gl["route_crc32"], gl["cost"], gl["stats"]["sweeps"], gl["stats"]["moves"])))
dm1k = O.dm_build_packed(xy1k)
jobs.append(("2opt matrix n=1002", lambda c: sol_key(TA.two_opt.solve(prob(xy1k, dm1k), ctx=c)), okey(O.two_opt(None, dm1k, 1002))))
nnb = O.nearest_neighbor(berlin, None, 52, 3)[1]
jobs.append(("3opt berlin52 nn", lambda c: sol_key(TA.three_opt.solve(prob(berlin), None, None, [int(v) for v in nnb], ctx=c)),
             okey(O.three_opt(berlin, None, 52, init=nnb))))
xy150 = O.synth_xy(150, seed=9)
jobs.append(("oropt n=150", lambda c: sol_key(TA.or_opt.solve(prob(xy150), ctx=c)), okey(O.or_opt(xy150, None, 150))))


def lk(c, xy, seed, epochs):
    s = TA.lin_kernighan.solve(prob(xy), TA.LKOptions(TA.HeuristicOptions(epochs=epochs, platoo_epochs=10, n_nearest=5), 5), ctx=c, seed=seed)
    return (crc(np.asarray(s.route(), dtype=np.uint32)), bits(s.total), s.stats["sweeps"], s.stats["moves"])


jobs.append(("lk berlin52", lambda c: lk(c, berlin, 3, 100), okey(O.lin_kernighan(berlin, seed=3))))
jobs.append(("lk n=2000 (chip-wide step, hipGraph)", lambda c: lk(c, xy2k, 5, 4), okey(O.lin_kernighan(xy2k, seed=5, epochs=4))))
jobs.append(("dm_build n=2000", lambda c: crc(TA.distance_matrix.build(np.arange(2000), xy2k, ctx=c).items.view(np.uint32)),
             crc(O.dm_build_packed(xy2k).view(np.uint32))))
onn = O.nearest_neighbor(xy5k, None, 5000, 3)
jobs.append(("nn seed n=5000", lambda c: (lambda s: (crc(np.asarray(s.route(), dtype=np.uint32)), bits(s.total)))(TA.nearest_neighbor.solve(prob(xy5k), ctx=c)),
             (crc(onn[1]), bits(onn[2]))))
jobs.append(("candidates n=5000 k=5", lambda c: crc(TA.lin_kernighan.build_candidates(prob(xy5k), 5, ctx=c)), crc(O.build_candidates_kdtree(xy5k, 5)[0])))
oms = [O.two_opt(xy1k, None, 1002, init=O.restart_perm(1002, 99, r)) for r in range(6)]


def multistart(c):
    s, costs = TA.two_opt.multistart(prob(xy1k), 6, seed=99, ctx=c, return_costs=True)
    return (crc(np.asarray(s.route(), dtype=np.uint32)), costs.tobytes())


best = min(range(6), key=lambda r: (oms[r][2], r))
jobs.append(("multistart 6 x n=1002", multistart, (crc(oms[best][1]), np.asarray([o[2] for o in oms], dtype=np.float32).tobytes())))
xy500 = O.synth_xy(500, seed=4)
jobs.append(("2opt best-sweep n=500", lambda c: sol_key(TA.two_opt.solve(prob(xy500), ctx=c, mode=TA.TL_MODE_BEST_SWEEP)),
             okey(O.two_opt(xy500, None, 500, best=True))))
print(f"{len(jobs)} jobs, oracle expectations in {time.time() - t_or:.1f} s; library: {TA._capi.LIB_PATH}", flush=True)

runs = [0] * nthreads
fails = []
lock = threading.Lock()
t_end = time.time() + budget


def worker(k):
    with TA.Context(0) as c:
        q = (k * 5) % len(jobs)  # every thread starts somewhere else: different kernels and LDS sizes overlap
        first = True
        while first or time.time() < t_end:
            for r in range(len(jobs)):
                name, fn, want = jobs[(q + r) % len(jobs)]
                try:
                    got = fn(c)
                except Exception as exc:  # an error code from the library is a failure of the contract too
                    got = repr(exc)
                runs[k] += 1
                if got != want:
                    with lock:
                        fails.append((k, name, got, want))
                if not first and time.time() >= t_end:
                    break
            first = False  # every thread runs every job at least once


t0 = time.time()
th = [threading.Thread(target=worker, args=(k,)) for k in range(nthreads)]
[t.start() for t in th]
[t.join() for t in th]
for f in fails[:10]:
    print("MISMATCH thread %d %s: got %r want %r" % f, flush=True)
print(f"own contexts: {nthreads} threads, {sum(runs)} runs ({min(runs)}..{max(runs)} per thread), {len(fails)} mismatches, {time.time() - t0:.0f} s", flush=True)

# ---- two threads on ONE context: right, or TL_ERR_BUSY — never a wrong tour, never a crash
shared = {"ok": 0, "busy": 0, "bad": []}
want1k = jobs[1][2]
with TA.Context(0) as c:
    stop = time.time() + min(4.0, budget / 4 + 1.0)

    def hammer(k):
        while time.time() < stop:
            try:
                got = jobs[1][1](c) if k == 0 else jobs[0][1](c)
                with lock:
                    if got == (want1k if k == 0 else jobs[0][2]):
                        shared["ok"] += 1
                    else:
                        shared["bad"].append((k, got))
            except TA._capi.TeelineGpuError as exc:
                with lock:
                    if exc.code == TA._capi.TL_ERR_BUSY:
                        shared["busy"] += 1
                    else:
                        shared["bad"].append((k, repr(exc)))

    th = [threading.Thread(target=hammer, args=(k,)) for k in range(2)]
    [t.start() for t in th]
    [t.join() for t in th]
    # the context is still good afterwards
    after = jobs[0][1](c) == jobs[0][2]
print(f"shared context: {shared['ok']} right, {shared['busy']} TL_ERR_BUSY, {len(shared['bad'])} wrong, usable afterwards: {after}", flush=True)
for b in shared["bad"][:5]:
    print("SHARED-CONTEXT FAILURE", b, flush=True)
bad = len(fails) + len(shared["bad"]) + (0 if after else 1) + (0 if shared["ok"] > 0 else 1)
print(f"thread campaign: {sum(runs) + shared['ok'] + shared['busy']} runs, {bad} mismatches")
sys.exit(1 if bad else 0)
