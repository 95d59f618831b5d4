"""Developer probe: randomized parity campaign of Or-opt (single scans and full solves, coordinate and matrix form) and
3-opt scans against the oracle.   python tests/probes/fuzz_campaign_oropt.py [seconds]"""
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
        rng = np.random.default_rng(5000 + seed)
        n = int(rng.integers(4, 500)) if seed % 6 else int(rng.integers(500, 2600))
        kind = seed % 4
        if kind == 0: xy = rng.random((n, 2)) * 1000
        elif kind == 1: xy = rng.integers(0, int(rng.integers(2, 25)), (n, 2))
        elif kind == 2:
            c = rng.random((int(rng.integers(2, 9)), 2)) * 1000; xy = c[rng.integers(0, len(c), n)] + rng.normal(0, 1.0, (n, 2))
        else:
            t = np.sort(rng.random(n)) * 1000; xy = np.stack([t, 0.3 * t + rng.normal(0, 0.01, n)], 1)
        xy = np.ascontiguousarray(xy, dtype=np.float32)
        path = O.restart_perm(n, seed, 0) if seed % 2 else np.arange(n, dtype=np.uint32)
        prob = TA.TspProblem(np.arange(n), xy)
        # one scan
        g = TA.or_opt.find_best_move(prob, path, ctx=ctx)
        o = O.or_opt_find_best_move(xy, None, path)
        runs += 1
        same = (g is None and o is None) or (g is not None and o is not None and np.float32(g[0]).tobytes() == np.float32(o[0]).tobytes() and tuple(g[1:]) == tuple(o[1:]))
        if not same:
            fails += 1
            print(f"OR-OPT SCAN MISMATCH seed={seed} n={n} kind={kind}: gpu {g} oracle {o}", flush=True)
        if n <= 400:  # full solve (coordinate form; matrix form every third)
            forms = [("coord", prob, None)]
            if seed % 3 == 0:
                dm = TA.distance_matrix.build(np.arange(n), xy, ctx=ctx)
                forms.append(("matrix", TA.TspProblem(np.arange(n), xy, TA.distance_matrix.DistanceMatrix(n, dm.items, np.arange(n), "explicit")), dm.items))
            for name, p, packed in forms:
                sol = TA.or_opt.solve(p, None, None, [int(v) for v in path], ctx=ctx)
                rc, route, cost, st = O.or_opt(xy if packed is None else None, packed, n, init=path)
                runs += 1
                if list(sol.route()) != route.tolist() or np.float32(sol.total).tobytes() != np.float32(cost).tobytes() or sol.stats["moves"] != st["moves"]:
                    fails += 1
                    print(f"OR-OPT SOLVE MISMATCH seed={seed} n={n} kind={kind} form={name}: gpu {float(sol.total)!r}/{sol.stats['moves']} oracle {float(cost)!r}/{st['moves']}", flush=True)
        if n <= 300:
            g3 = TA.three_opt.find_best_move(prob, path, ctx=ctx)
            o3 = O.three_opt_find_best_move(xy, None, path)
            runs += 1
            if repr(g3) != repr(o3) and not (g3 is not None and o3 is not None and tuple(map(float, g3)) == tuple(map(float, o3))):
                fails += 1
                print(f"3-OPT SCAN MISMATCH seed={seed} n={n} kind={kind}: gpu {g3} oracle {o3}", flush=True)
print(f"Or-opt / 3-opt fuzz campaign: {runs} runs, {fails} mismatches, {time.time() - t0:.0f} s")
