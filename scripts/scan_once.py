"""The launches bench.py's extras time, once each, for a rocprofv3 --pmc pass (scripts/pmc_scans.sh): the 3-opt scan at n = 1002 and
the Or-opt scan at n = 5000 over the NN tour (find_best_move, twice: the second is the one whose counters are kept) and the LK run at
n = 13 509 / 20 epochs; `best`: the BEST_SWEEP descent at n = 10^4 from the NN seed; `noprune`: the 256-restart batch with TL_FLAG_NO_PRUNE."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import teeline_amd as TA
ctx = TA.Context(0)
what = sys.argv[1] if len(sys.argv) > 1 else "scans"
if what == "scans":
    p3 = TA.TspProblem(np.arange(1002), TA.synth.synth_xy(1002))
    nn3 = [int(v) for v in TA.nearest_neighbor.solve(p3, ctx=ctx).route()]
    for _ in range(2):
        TA.three_opt.find_best_move(p3, nn3, ctx=ctx)
    print("3opt kernel_ms", ctx.last_kernel_ms())
    p5 = TA.TspProblem(np.arange(5000), TA.synth.synth_xy(5000))
    nn5 = [int(v) for v in TA.nearest_neighbor.solve(p5, ctx=ctx).route()]
    for _ in range(2):
        TA.or_opt.find_best_move(p5, nn5, ctx=ctx)
    print("oropt kernel_ms", ctx.last_kernel_ms())
elif what == "best":
    p = TA.TspProblem(np.arange(10000), TA.synth.synth_xy(10000))
    nn = TA.nearest_neighbor.solve(p, ctx=ctx).route()
    s = TA.two_opt.solve(p, None, None, nn, ctx=ctx, mode=TA.TL_MODE_BEST_SWEEP)
    print("best-sweep kernel_ms", s.stats["kernel_ms"], "sweeps", s.stats["sweeps"])
elif what == "noprune":
    p = TA.TspProblem(np.arange(10000), TA.synth.synth_xy(10000))
    with TA.Context(0, TA.TL_FLAG_NO_PRUNE) as c2:
        s = TA.two_opt.multistart(p, 256, seed=12345, first=0, ctx=c2)
    print("no-prune kernel_ms", s.stats["kernel_ms"], "cost", float(s.total))
else:
    p13 = TA.TspProblem(np.arange(13509), TA.synth.synth_xy(13509))
    s = TA.lin_kernighan.solve(p13, TA.LKOptions(TA.HeuristicOptions(epochs=20, platoo_epochs=10, n_nearest=5), 5), ctx=ctx, seed=1)
    print("lk kernel_ms", s.stats["kernel_ms"], "scans", s.stats["sweeps"], "moves", s.stats["moves"])
