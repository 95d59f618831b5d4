"""Tuning build (TEELINE_GPU_LIB=…/libteeline_gpu_tune.so): the matrix form's late-sweep thresholds TL_DM_LONG_MAX / TL_DM_MOVES_MAX
over the three starts at n = 1002 (and other sizes given on the command line)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import numpy as np, teeline_amd as TA
n = int(sys.argv[1])
xy = TA.synth.synth_xy(n)
with TA.Context(0) as ctx:
    dm = TA.distance_matrix.build(np.arange(n), xy, ctx=ctx)
    pm = TA.TspProblem(np.arange(n), xy, TA.distance_matrix.DistanceMatrix(n, dm.items, np.arange(n), "explicit"))
    nn = [int(v) for v in TA.nearest_neighbor.solve(TA.TspProblem(np.arange(n), xy), ctx=ctx).route()]
    rnd = [int(v) for v in TA.synth.restart_perm(n, 1, 0)]
    out = []
    for name, init in (("nn", nn), ("identity", None), ("random", rnd)):
        for _ in range(3):
            s = TA.two_opt.solve(pm, None, None, init, ctx=ctx)
        cnt = ctx.two_opt_last_counters()
        out.append(f"{name} {s.stats['kernel_ms']:.3f} ms ({cnt[6]} late sweeps, {cnt[5]} steps)")
    pop = [[int(v) for v in TA.synth.restart_perm(n, 1, r)] for r in range(256)]
    for _ in range(2):
        sols = TA.two_opt.solve_population(pm, pop, ctx=ctx)
    out.append(f"pop256 {sols[0].stats['kernel_ms']:.3f} ms")
    print(" | ".join(out))
''' % (ROOT, ROOT)
ns = [int(v) for v in sys.argv[1:]] or [1002]
env0 = dict(os.environ, TEELINE_GPU_LIB=os.path.join(ROOT, "teeline_amd", "libteeline_gpu_tune.so"))
for n in ns:
    for lm, mm in ((0, 0), (256, 10**9), (256, n), (256, n // 2), (256, n // 4), (256, n // 8), (256, n // 16), (128, 10**9), (128, n // 4)):
        env = dict(env0, TL_DM_LONG_MAX=str(lm), TL_DM_MOVES_MAX=str(mm))
        r = subprocess.run([sys.executable, "-c", CHILD, str(n)], env=env, capture_output=True, text=True)
        print(f"n={n} long_max {lm:4d} moves_max {mm:10d}: {r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-400:]}", flush=True)
