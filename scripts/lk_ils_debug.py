import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, teeline_amd as TA
import _tsplib as T
name, epochs = sys.argv[1], int(sys.argv[2])
xy = T.parse_tsplib(os.path.join(ROOT, "tests", "golden", "tsplib", f"{name}.tsp"))["xy"]
with TA.Context(0, TA.TL_FLAG_LK_ILS_LDS) as ctx:
    sol = TA.lin_kernighan.solve(TA.TspProblem(np.arange(len(xy)), xy), TA.LKOptions(TA.HeuristicOptions(epochs=epochs, platoo_epochs=500, n_nearest=3), 5), None, None, ctx=ctx, seed=1)
    print(name, sol.stats, float(sol.total), flush=True)
