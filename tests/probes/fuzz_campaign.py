"""Developer probe: a long randomized parity campaign of REF_ORDER 2-opt (coordinate and matrix form) against the oracle.
   python tests/probes/fuzz_campaign.py [seconds]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, _oracle as O, teeline_amd as TA

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
t0 = time.time(); runs = fails = 0
with TA.Context(0) as ctx, TA.Context(0, TA.TL_FLAG_NO_PRUNE) as ctx2, TA.Context(0, TA.TL_FLAG_2OPT_NT512) as ctx3, \
        TA.Context(0, TA.TL_FLAG_2OPT_NT256) as ctx4, TA.Context(0, TA.TL_FLAG_2OPT_FX) as ctx5, TA.Context(0, TA.TL_FLAG_2OPT_NL_ALWAYS) as ctx6, \
        TA.Context(0, TA.TL_FLAG_2OPT_NO_NL) as ctx7, TA.Context(0, TA.TL_FLAG_2OPT_NL_ALWAYS | TA.TL_FLAG_2OPT_NT512) as ctx8, \
        TA.Context(0, TA.TL_FLAG_2OPT_NL_ALWAYS | TA.TL_FLAG_2OPT_NT256) as ctx9:
    seed = 0
    while time.time() - t0 < budget:
        seed += 1
        rng = np.random.default_rng(seed)
        n = int(rng.integers(4, 3200)) if seed % 4 else int(rng.integers(4, 200))
        if os.environ.get("FUZZ_BIG"):  # long rows: deferred reversals over many register slots, all tile groups
            n = int(rng.integers(3200, 14500))
        kind = seed % 8
        if kind == 0: xy = rng.random((n, 2)) * 1000
        elif kind == 1: xy = rng.integers(0, int(rng.integers(2, 40)), (n, 2))
        elif kind == 2:
            c = rng.random((int(rng.integers(2, 12)), 2)) * 1000; xy = c[rng.integers(0, len(c), n)] + rng.normal(0, 0.5, (n, 2))
        elif kind == 3:
            t = np.sort(rng.random(n)) * 1000; xy = np.stack([t, 0.25 * t], 1)          # sorted collinear: long chains per row
        elif kind == 4:
            a = np.sort(rng.random(n)) * 2 * np.pi; xy = np.stack([np.cos(a), np.sin(a)], 1) * 300 + 300
        elif kind == 6: xy = (rng.integers(0, 1000000, (n, 2)).astype(np.float32) / np.float32(10.0 ** int(rng.integers(0, 5)))).astype(np.float32)  # decimal grids
        elif kind == 7: xy = rng.integers(0, 1 << int(rng.integers(8, 21)), (n, 2))                       # integer coordinates up to 2^20
        else: xy = rng.normal(0, 1, (n, 2)) * 10.0 ** rng.integers(-3, 4, (n, 1))
        xy = np.ascontiguousarray(xy, dtype=np.float32)
        start = seed % 3
        init = None if start == 0 else (O.restart_perm(n, seed, 0) if start == 1 else np.arange(n, dtype=np.uint32)[::-1].copy())
        rc, route, cost, st = O.two_opt(xy, None, n, init=init)
        prob = TA.TspProblem(np.arange(n), xy)
        cases = [("coord", prob, ctx)]
        if seed % 5 == 0: cases.append(("noprune", prob, ctx2))
        # the 8- and 4-wave forms of a descent (what a batch runs when two or four descents share a CU); a flush holds 15 elements
        # per thread, so they take n <= 7680 / 3840 (beyond that the library falls back to the wider form by itself)
        if seed % 2 == 0: cases.append(("nt512", prob, ctx3))
        if seed % 3 == 0: cases.append(("nt256", prob, ctx4))
        # grid-coordinate form of the tour (used where the instance lies on a decimal grid: kinds 0 / 2 / 4 / 5 mostly do not, kind 1
        # — integer lattices — does; the library falls back to float2 by itself)
        if seed % 2 == 1: cases.append(("fx", prob, ctx5))
        # the late phase (neighbour-list rows, csrc/two_opt_nl.hip) from the second sweep on at every n it takes (n >= 26; by default
        # it starts later in a descent and only from n = 3000), and the kernel without it
        cases.append(("nl", prob, ctx6))
        if seed % 4 == 0: cases.append(("no_nl", prob, ctx7))
        if seed % 3 == 1: cases.append(("nl_nt512", prob, ctx8))  # the 8-wave form with its late phase (what a batch of two descents per CU runs)
        if seed % 3 == 2: cases.append(("nl_nt256", prob, ctx9))  # ... and the 4-wave form (four per CU)
        if n <= 1500 and seed % 3 == 0:
            dm = TA.distance_matrix.build(np.arange(n), xy, ctx=ctx)
            pm = TA.TspProblem(np.arange(n), xy, TA.distance_matrix.DistanceMatrix(n, dm.items, np.arange(n), "explicit"))
            cases.append(("matrix", pm, ctx))
            cases.append(("matrix_nl", pm, ctx6))  # its late sweeps on lists wherever they fit (two_opt_dm.hip; by default from n = 500)
        if os.environ.get("FUZZ_DM"):  # the matrix form only, every seed: the Euclidean matrix and a non-metric one (independent weights, ties)
            n = int(rng.integers(8, 1400))
            xy = np.ascontiguousarray(xy[:n] if len(xy) >= n else rng.random((n, 2)) * 1000, dtype=np.float32)
            init = None if start == 0 else (O.restart_perm(n, seed, 0) if start == 1 else O.nearest_neighbor(xy, None, n, 3)[1])
            packed = O.dm_build_packed(xy)
            if seed % 2:
                m = n * (n - 1) // 2
                packed = (rng.integers(0, int(rng.integers(3, 2000)), m).astype(np.float32) if seed % 4 == 1 else rng.random(m, dtype=np.float32) * np.float32(100.0))
            rc, route, cost, st = O.two_opt(None, packed, n, init=init)
            pm = TA.TspProblem(np.arange(n), xy, TA.distance_matrix.DistanceMatrix(n, packed, np.arange(n), "explicit"))
            cases = [("matrix", pm, ctx), ("matrix_nl", pm, ctx6)] + ([("matrix_no_nl", pm, ctx7)] if seed % 4 == 0 else [])
        for name, p, c in cases:
            sol = TA.two_opt.solve(p, None, None, None if init is None else [int(v) for v in init], ctx=c)
            a, b = np.float32(sol.total), np.float32(cost)
            ok = list(sol.route()) == route.tolist() and (a.tobytes() == b.tobytes() or (np.isnan(a) and np.isnan(b))) and \
                (sol.stats["sweeps"], sol.stats["moves"], sol.stats["reversed"]) == (st["sweeps"], st["moves"], st["reversed"])
            runs += 1
            if not ok:
                fails += 1
                print(f"MISMATCH seed={seed} n={n} kind={kind} start={start} form={name}: gpu {float(a)!r} {sol.stats['moves']} vs oracle {float(b)!r} {st['moves']}", flush=True)
print(f"fuzz campaign: {runs} runs, {fails} mismatches, {time.time() - t0:.0f} s")
