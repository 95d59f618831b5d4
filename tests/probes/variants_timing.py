"""Developer probe: time build variants (build_variants/*.so) on the n=10^4 descents; each in a subprocess."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
probe = r'''
import os, sys
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import numpy as np, _oracle as O, teeline_amd as TA
n = 10000
xy = O.synth_xy(n); prob = TA.TspProblem(np.arange(n), xy)
with TA.Context(0) as ctx:
    rc, nn, _ = O.nearest_neighbor(xy, None, n, 3)
    for name, init in (("greedy", nn), ("random", O.restart_perm(n, 12345, 0))):
        for rep in range(2):
            sol = TA.two_opt.solve(prob, None, None, [int(v) for v in init], ctx=ctx)
        s = sol.stats
        print(f"  {name}: cost={float(sol.total):.5f} moves={s['moves']} kernel_ms={s['kernel_ms']:.2f} -> {s['candidates']/s['kernel_ms']/1e6:.2f} Gcand/s")
    sol = TA.two_opt.multistart(prob, 256, seed=12345, ctx=ctx)
    s = sol.stats
    print(f"  multistart256: kernel_ms={s['kernel_ms']:.2f} -> {s['candidates']/s['kernel_ms']/1e6:.2f} Gcand/s")
''' % (ROOT, ROOT)
for v in sys.argv[1:]:
    env = dict(os.environ)
    if v != "default":
        env["TEELINE_GPU_LIB"] = os.path.join(ROOT, "build_variants", v + ".so")
    if v == "prof":
        env["TL_DUMP_STATS"] = "1"
    print(f"== {v}", flush=True)
    subprocess.run([sys.executable, "-c", probe], env=env)
