"""Developer probe: randomized parity campaign of LK (three-level split scan, kept chains, prefix window) and the NN seed
against the oracle.   python tests/probes/fuzz_campaign_lk.py [seconds]      FUZZ_DEEP=1: max_depth 7..12 (the lk_deep build), n <= 400
FUZZ_ILS=1: the LDS form's speculative epochs at depth — n <= 300, up to 400 epochs with a plateau of up to 120 (round 5).  A run on which the
library reports a cycling lk_pass (TL_ERR_NO_CONVERGE) is counted and skipped: the oracle would not return from it."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, _oracle as O, teeline_amd as TA

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
t0 = time.time(); runs = fails = 0
with TA.Context(0) as ctx:
    seed = 0
    while time.time() - t0 < budget:
        seed += 1
        rng = np.random.default_rng(1000 + seed)
        deep = bool(os.environ.get("FUZZ_DEEP"))
        n = int(rng.integers(5, 400)) if deep else (int(rng.integers(5, 700)) if seed % 5 else int(rng.integers(700, 2500)))
        kind = seed % 4
        if kind == 0: xy = rng.random((n, 2)) * 1000
        elif kind == 1: xy = rng.integers(0, int(rng.integers(3, 30)), (n, 2))
        elif kind == 2:
            c = rng.random((int(rng.integers(2, 9)), 2)) * 1000; xy = c[rng.integers(0, len(c), n)] + rng.normal(0, 1.0, (n, 2))
        else:
            a = rng.random(n) * 2 * np.pi; xy = np.stack([np.cos(a), np.sin(a)], 1) * 300 + 300
        xy = np.ascontiguousarray(xy, dtype=np.float32)
        k = int(rng.integers(1, 6 if deep else 9)); depth = int(rng.integers(7, 13) if deep else rng.integers(1, 7)); epochs = int(rng.integers(0, 12)); s = int(rng.integers(1, 1 << 30))
        platoo = 4
        if os.environ.get("FUZZ_ILS"):
            n = min(n, int(rng.integers(8, 300))); xy = xy[:n]
            epochs = int(rng.integers(20, 400)); platoo = int(rng.integers(5, 120)); k = int(rng.integers(2, 6)); depth = int(rng.integers(2, 6))
        h = TA.HeuristicOptions(epochs=epochs, platoo_epochs=platoo, n_nearest=k)
        try:
            sol = TA.lin_kernighan.solve(TA.TspProblem(np.arange(n), xy), TA.LKOptions(h, depth), ctx=ctx, seed=s)
        except TA.TeelineGpuError as exc:
            if exc.code != TA._capi.TL_ERR_NO_CONVERGE:
                raise
            cycling = globals().get("cycling", 0) + 1
            globals()["cycling"] = cycling
            continue
        rc, route, cost, st = O.lin_kernighan(xy, seed=s, epochs=epochs, platoo_epochs=platoo, n_nearest=k, max_depth=depth)
        ok = list(sol.route()) == route.tolist() and np.float32(sol.total).tobytes() == np.float32(cost).tobytes() and \
            (sol.stats["sweeps"], sol.stats["candidates"], sol.stats["moves"], sol.stats["reversed"]) == (st["sweeps"], st["candidates"], st["moves"], st["reversed"])
        runs += 1
        if not ok:
            fails += 1
            print(f"LK MISMATCH seed={seed} n={n} kind={kind} k={k} depth={depth} epochs={epochs}: gpu {float(sol.total)!r} {sol.stats} vs oracle {float(cost)!r} {st}", flush=True)
        # candidate lists: the kd-tree walk against the oracle's restated kd-tree (lattices and clusters hold distance ties)
        kc = int(rng.integers(1, 17))
        got = TA.lin_kernighan.build_candidates(TA.TspProblem(np.arange(n), xy), kc, ctx=ctx)
        want, _ = O.build_candidates_kdtree(xy, kc)
        runs += 1
        if not np.array_equal(got, want):
            fails += 1
            print(f"KNN MISMATCH seed={seed} n={n} kind={kind} k={kc}: {int((got != want).any(axis=1).sum())} rows", flush=True)
        if n <= 400 and seed % 3 == 0:  # NN seed over a matrix (GEO / EXPLICIT problems)
            packed = O.dm_build_packed(xy)
            pm = TA.TspProblem(np.arange(n), xy, TA.distance_matrix.DistanceMatrix(n, packed, np.arange(n), "explicit"))
            nnm = TA.nearest_neighbor.solve(pm, TA.HeuristicOptions(n_nearest=3), ctx=ctx)
            rc, r3, c3 = O.nearest_neighbor(None, packed, n, 3)
            runs += 1
            if list(nnm.route()) != r3.tolist() or np.float32(nnm.total).tobytes() != np.float32(c3).tobytes():
                fails += 1
                print(f"NN-DM MISMATCH seed={seed} n={n} kind={kind}", flush=True)
        kk = int(rng.integers(0, 9))
        nn = TA.nearest_neighbor.solve(TA.TspProblem(np.arange(n), xy), TA.HeuristicOptions(n_nearest=max(kk, 1)), ctx=ctx)
        rc, r2, c2 = O.nearest_neighbor(xy, None, n, max(kk, 1))
        runs += 1
        if list(nn.route()) != r2.tolist() or np.float32(nn.total).tobytes() != np.float32(c2).tobytes():
            fails += 1
            print(f"NN MISMATCH seed={seed} n={n} kind={kind} k={kk}", flush=True)
print(f"LK/NN fuzz campaign: {runs} runs, {fails} mismatches, {time.time() - t0:.0f} s" + (f" ({globals().get('cycling', 0)} runs reported a cycling lk_pass)" if os.environ.get("FUZZ_ILS") else ""))
