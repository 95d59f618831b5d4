"""Developer probe: a randomized campaign of the progress channel (ABI v4 *_trace entries through the Python mirrors) against the
oracle: 2-opt move lists (coordinates and matrix form), the 3-opt and Or-opt message streams restated with the oracle's pieces,
LK's best-tour messages (tlo_lin_kernighan_trace).
   python tests/probes/fuzz_campaign_trace.py [seconds]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, _oracle as O, teeline_amd as TA

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
t0 = time.time(); runs = fails = 0; last = t0


def bits(v):
    return np.float32(v).tobytes()


with TA.Context(0) as ctx:
    seed = 0
    while time.time() - t0 < budget:
        seed += 1
        rng = np.random.default_rng(seed)
        kind = seed % 4
        n = int(rng.integers(4, 700)) if kind != 3 else int(rng.integers(4, 120))
        if seed % 3 == 0: xy = rng.integers(0, int(rng.integers(2, 30)), (n, 2))       # lattices: ties everywhere
        elif seed % 3 == 1: xy = rng.random((n, 2)) * 1000
        else: xy = rng.normal(0, 1, (n, 2)) * 10.0 ** rng.integers(-2, 3, (n, 1))
        xy = np.ascontiguousarray(xy, dtype=np.float32)
        ids = np.arange(n) + int(rng.integers(0, 50))
        init = None if seed % 5 == 0 else O.restart_perm(n, seed, 0)
        packed = O.dm_build_packed(xy) if seed % 2 else None
        prob = TA.TspProblem(ids, xy, None if packed is None else TA.distance_matrix.DistanceMatrix(n, packed, ids, "explicit"))
        init_ids = None if init is None else [int(ids[v]) for v in init]
        got = []
        tx = lambda k, p: got.append((k, p))  # noqa: E731
        ok = True
        if kind == 0:      # 2-opt: every PathUpdate's new_distance = the oracle's record; CityChange count; final route
            sol = TA.two_opt.solve(prob, None, tx, init_ids, ctx=ctx)
            rc, route, cost, st, ij, dist, sw = O.two_opt_trace(xy, packed, n, init=init)
            pu = [m for m in got if m[0] == "PathUpdate"]
            ok = (list(sol.route()) == [int(ids[v]) for v in route] and len(pu) == 1 + len(ij) and got[-1] == ("Done", None) and
                  [bits(m[1][1]) for m in pu[1:]] == [bits(v) for v in dist] and
                  sum(1 for m in got if m[0] == "CityChange") == st["sweeps"] * max(n - 3, 0) and (not len(ij) or pu[-1][1][0] == list(sol.route())))
        elif kind == 1:    # Or-opt: the path and its f32 tour length after every relocation
            sol = TA.or_opt.solve(prob, None, tx, init_ids, ctx=ctx)
            tour = np.arange(n, dtype=np.uint32) if init is None else init.copy()
            want = [("PathUpdate", ([int(ids[v]) for v in tour], 0.0))]
            while True:
                mv = O.or_opt_find_best_move(xy, packed, tour)
                if mv is None:
                    break
                rc, tour = O.apply_relocation(tour, mv[1], mv[3], mv[2], mv[4])
                want.append(("PathUpdate", ([int(ids[v]) for v in tour], float(O.tour_length(xy, packed, tour)))))
            want.append(("Done", None))
            ok = len(got) == len(want) and all(a[0] == b[0] and (a[0] == "Done" or (a[1][0] == b[1][0] and bits(a[1][1]) == bits(b[1][1]))) for a, b in zip(got, want))
        elif kind == 2:    # LK: best tours and best_dist in order
            ep = int(rng.integers(1, 25))
            h = TA.HeuristicOptions(epochs=ep, platoo_epochs=10, n_nearest=int(rng.integers(2, 8)))
            md = int(rng.integers(2, 6))
            sol = TA.lin_kernighan.solve(prob, TA.LKOptions(h, md), tx, init_ids, ctx=ctx, seed=seed)
            rc, oroute, ocost, ost, snaps = O.lin_kernighan_trace(xy, init=init, epochs=ep, platoo_epochs=10, n_nearest=h.n_nearest, max_depth=md, seed=seed, packed=packed)
            ok = (len(got) == len(snaps) and list(sol.route()) == [int(ids[v]) for v in oroute] and bits(sol.total) == bits(ocost) and
                  all(g[0] == "PathUpdate" and g[1][0] == [int(ids[v]) for v in s[0]] and bits(g[1][1]) == bits(s[1]) for g, s in zip(got, snaps)))
        else:              # 3-opt: the path after every apply_3opt
            cap3 = 64 * n + 1024
            if O.three_opt(xy, packed, n, init=init, max_moves=cap3)[3]["moves"] >= cap3:
                # the reference's loop does not terminate here (savings > 0.0 in f32 on a cycle of neutral moves, three_opt.rs:61,121):
                # the library gives up after the same number of passes
                try:
                    TA.three_opt.solve(prob, None, tx, init_ids, ctx=ctx)
                    ok = False
                except TA._capi.TeelineGpuError as e:
                    ok = e.code == TA._capi.TL_ERR_NO_CONVERGE
                runs += 1
                fails += 0 if ok else 1
                continue
            sol = TA.three_opt.solve(prob, None, tx, init_ids, ctx=ctx)
            path = np.arange(n, dtype=np.uint32) if init is None else init.copy()
            want = [[int(ids[v]) for v in path]]
            while True:
                mv = O.three_opt_find_best_move(xy, packed, path)
                if mv is None:
                    break
                rc, path = O.apply_3opt(path, *mv[:4])
                want.append([int(ids[v]) for v in path])
            ok = got[-1] == ("Done", None) and [m[1][0] for m in got[:-1]] == want and all(m[1][1] == 0.0 for m in got[:-1])
        runs += 1
        if not ok:
            fails += 1
            print(f"MISMATCH seed={seed} n={n} kind={kind} matrix={packed is not None}", flush=True)
        if time.time() - last > 60:
            last = time.time()
            print(f"... {runs} runs, {fails} mismatches, {last - t0:.0f} s", flush=True)
print(f"progress-channel fuzz campaign: {runs} runs, {fails} mismatches, {time.time() - t0:.0f} s")
