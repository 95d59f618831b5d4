import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, _oracle as O, teeline_amd as TA
n = 1002
xy = O.synth_xy(n)
with TA.Context(0) as ctx:
    dm = TA.distance_matrix.build(np.arange(n), xy, ctx=ctx)
    pm = TA.TspProblem(np.arange(n), xy, TA.distance_matrix.DistanceMatrix(n, dm.items, np.arange(n), "explicit"))
    rc, nn, _ = O.nearest_neighbor(xy, None, n, 3)
    for name, init in (("nn", nn), ("identity", None), ("random", O.restart_perm(n, 5, 0))):
        for rep in range(2):
            s = TA.two_opt.solve(pm, None, None, None if init is None else [int(v) for v in init], ctx=ctx)
        st = s.stats
        print(f"{os.path.basename(os.environ.get('TEELINE_GPU_LIB','default'))} {name:8s}: moves {st['moves']} steps {st['reversed']} kernel {st['kernel_ms']:.3f} ms")
