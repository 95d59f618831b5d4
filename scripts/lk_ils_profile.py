"""Per-phase cycles of k_lk_ils (a -DTL_PROFILE_ILS build: TEELINE_GPU_LIB=build_variants/libtl_profile_ils.so) on berlin52 / a280 / synthetic n."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, teeline_amd as TA
import _tsplib as T
cases = [(name, T.parse_tsplib(os.path.join(ROOT, "tests", "golden", "tsplib", f"{name}.tsp"))["xy"], dict(epochs=10000, platoo_epochs=500, n_nearest=3)) for name in ("berlin52", "a280")]
cases.append(("synth1000", TA.synth.synth_xy(1000), dict(epochs=100, platoo_epochs=10, n_nearest=5)))
for name, xy, h in cases:
    with TA.Context(0, TA.TL_FLAG_LK_ILS_LDS) as ctx:
        sol = TA.lin_kernighan.solve(TA.TspProblem(np.arange(len(xy)), xy), TA.LKOptions(TA.HeuristicOptions(**h), 5), None, None, ctx=ctx, seed=1)
        print(name, sol.stats["sweeps"], "rounds", sol.stats["kernel_ms"], "ms", sol.stats["kernel_ms"] * 1e3 / sol.stats["sweeps"], "us/round", flush=True)
