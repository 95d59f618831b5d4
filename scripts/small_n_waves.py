"""One descent at small n on 16 / 8 / 4 waves (coordinate form) and the matrix form beside it: which form a lone descent should take.
python scripts/small_n_waves.py [n ...]"""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, _oracle as O, teeline_amd as TA
for n in [int(v) for v in sys.argv[1:]] or [1002]:
    xy = O.synth_xy(n)
    rc, nn, _ = O.nearest_neighbor(xy, None, n, 3)
    starts = (("nn", nn), ("identity", None), ("random", O.restart_perm(n, 5, 0)))
    for tag, flags in (("16 waves", 0), ("8 waves", TA.TL_FLAG_2OPT_NT512), ("4 waves", TA.TL_FLAG_2OPT_NT256)):
        with TA.Context(0, flags=flags) as ctx:
            pc = TA.TspProblem(np.arange(n), xy)
            for name, init in starts:
                for rep in range(2):
                    s = TA.two_opt.solve(pc, None, None, None if init is None else [int(v) for v in init], ctx=ctx)
                print(f"n {n} coordinates {tag:9s} {name:8s}: moves {s.stats['moves']:6d} cost {s.total:.3f} kernel {s.stats['kernel_ms']:.3f} ms", flush=True)
    with TA.Context(0) as ctx:
        dm = TA.distance_matrix.build(np.arange(n), xy, ctx=ctx)
        pm = TA.TspProblem(np.arange(n), xy, TA.distance_matrix.DistanceMatrix(n, dm.items, np.arange(n), "explicit"))
        for name, init in starts:
            for rep in range(2):
                s = TA.two_opt.solve(pm, None, None, None if init is None else [int(v) for v in init], ctx=ctx)
            print(f"n {n} matrix form          {name:8s}: moves {s.stats['moves']:6d} cost {s.total:.3f} kernel {s.stats['kernel_ms']:.3f} ms", flush=True)
