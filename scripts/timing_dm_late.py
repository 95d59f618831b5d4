"""Matrix-form 2-opt at n = 1002 (BASELINE configs[1] shape): NN / identity / random start and the population of 256, with the late
sweeps on lists (default), wherever they fit (TL_FLAG_2OPT_NL_ALWAYS) and off (TL_FLAG_2OPT_NO_NL); counters 5-7 of descent 0."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, teeline_amd as TA
ns = [int(v) for v in sys.argv[1:]] or [1002]
for n in ns:
    xy = TA.synth.synth_xy(n)
    res = {}
    for form, flags in (("off", TA.TL_FLAG_2OPT_NO_NL), ("default", 0), ("always", TA.TL_FLAG_2OPT_NL_ALWAYS)):
        with TA.Context(0, flags) as ctx:
            dm = TA.distance_matrix.build(np.arange(n), xy, ctx=ctx)
            pm = TA.TspProblem(np.arange(n), xy, TA.distance_matrix.DistanceMatrix(n, dm.items, np.arange(n), "explicit"))
            nn = [int(v) for v in TA.nearest_neighbor.solve(TA.TspProblem(np.arange(n), xy), ctx=ctx).route()]
            rnd = [int(v) for v in TA.synth.restart_perm(n, 1, 0)]
            for name, init in (("nn", nn), ("identity", None), ("random", rnd)):
                for _ in range(3):
                    s = TA.two_opt.solve(pm, None, None, init, ctx=ctx)
                cnt = ctx.two_opt_last_counters()
                key = (name, tuple(s.route()), float(s.total))
                res.setdefault(name, []).append(key)
                print(f"n={n} {form:8s} {name:9s}: kernel {s.stats['kernel_ms']:7.3f} ms  sweeps {s.stats['sweeps']:3d} moves {s.stats['moves']:6d} steps {cnt[4]:6d}"
                      f" | late steps {cnt[5]:5d} late sweeps {cnt[6]:3d} matrix-row rows {cnt[7]:4d}", flush=True)
            pop = [[int(v) for v in TA.synth.restart_perm(n, 1, r)] for r in range(256)]
            for _ in range(2):
                sols = TA.two_opt.solve_population(pm, pop, ctx=ctx)
            st = sols[0].stats
            print(f"n={n} {form:8s} population-256: kernel {st['kernel_ms']:7.3f} ms  {st['candidates'] / st['kernel_ms'] / 1e6:7.2f} G candidates/s"
                  f"  best {min(float(s_.total) for s_ in sols):.3f}", flush=True)
            res.setdefault("pop", []).append(tuple(float(s_.total) for s_ in sols))
    for k, v in res.items():
        assert all(x == v[0] for x in v), f"{k}: the forms disagree"
    print(f"n={n}: all forms agree")
