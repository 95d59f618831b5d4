import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, teeline_amd as TA
n = int(os.environ.get("N", 13509))
xy = TA.synth.synth_xy(n)
with TA.Context(0) as ctx:
    s = TA.lin_kernighan.solve(TA.TspProblem(np.arange(n), xy), TA.LKOptions(TA.HeuristicOptions(epochs=5, platoo_epochs=10, n_nearest=5), 5), ctx=ctx, seed=1)
    print(s.stats, float(s.total))
