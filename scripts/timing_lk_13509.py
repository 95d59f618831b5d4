import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, teeline_amd as TA
n = 13509
xy = TA.synth.synth_xy(n)
p = TA.TspProblem(np.arange(n), xy)
fl = int(sys.argv[1], 0) if len(sys.argv) > 1 else 0
with TA.Context(0, fl) as ctx:
    best = 1e9
    for _ in range(3):
        s = TA.lin_kernighan.solve(p, TA.LKOptions(TA.HeuristicOptions(epochs=20, platoo_epochs=10, n_nearest=5), 5), ctx=ctx, seed=1)
        best = min(best, s.stats["kernel_ms"])
print(f"{os.environ.get('TEELINE_GPU_LIB', 'product')}: n={n} kernel {best:.2f} ms rounds {s.stats['sweeps']} {best * 1e3 / s.stats['sweeps']:.2f} us/round cost {float(s.total):.5f}", flush=True)
