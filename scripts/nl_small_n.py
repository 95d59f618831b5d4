"""Late phase at small n: kernel time of single descents (NN start, identity start, random start) for the library in TEELINE_GPU_LIB."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, teeline_amd as TA
for flags in (TA.TL_FLAG_2OPT_NO_NL, 0, TA.TL_FLAG_2OPT_NL_ALWAYS):
    with TA.Context(0, flags) as ctx:
        out = []
        for n in (500, 1002, 2000, 3000, 5000):
            xy = TA.synth.synth_xy(n)
            prob = TA.TspProblem(np.arange(n), xy)
            nn = TA.nearest_neighbor.solve(prob, ctx=ctx).route()
            rp = [int(v) for v in TA.synth.restart_perm(n, 12345, 0)]
            t = []
            for init in (nn, None, rp):
                t.append(min(TA.two_opt.solve(prob, None, None, init, ctx=ctx).stats["kernel_ms"] for _ in range(4)))
            out.append(f"n={n}: nn {t[0]:.3f} id {t[1]:.3f} rnd {t[2]:.3f}")
        print(f"{os.path.basename(os.environ.get('TEELINE_GPU_LIB', 'default'))} flags {flags:#x} | " + " | ".join(out))
