"""Where a step of the matrix-form 2-opt kernel spends its cycles (a -DTL_DM_PROFILE build, wave 0's s_memtime stamps):
python scripts/dm_profile.py [-Dflag ...]   (builds teeline_amd/libtl_dmprof.so and runs n = 1002 from the NN tour, the identity and a random permutation)"""
import ctypes, os, subprocess, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
lib = os.path.join(os.getcwd(), "teeline_amd", "libtl_dmprof.so")
if "TEELINE_GPU_LIB" not in os.environ:  # build the stamped library, then run this file again with it (a child; nothing here has touched the GPU)
    subprocess.check_call([sys.executable, "-c", "import sys; from teeline_amd import build as B; B.build(extra_flags=['-DTL_DM_PROFILE'] + sys.argv[1:], out=%r)" % lib] + sys.argv[1:])
    sys.exit(subprocess.call([sys.executable, __file__], env=dict(os.environ, TEELINE_GPU_LIB=lib)))
import numpy as np, _oracle as O, teeline_amd as TA
n = 1002
xy = O.synth_xy(n)
with TA.Context(0) as ctx:
    dm = TA.distance_matrix.build(np.arange(n), xy, ctx=ctx)
    rc, nn, _ = O.nearest_neighbor(xy, None, n, 3)
    pm = TA.TspProblem(np.arange(n), xy, TA.distance_matrix.DistanceMatrix(n, dm.items, np.arange(n), "explicit"))
    for name, init in (("nn", nn), ("identity", None), ("random", O.restart_perm(n, 5, 0))):
        print(f"--- {name}: (device line: wave 0's shader cycles per phase, summed over the steps of either block shape)", flush=True)
        s = TA.two_opt.solve(pm, None, None, None if init is None else [int(v) for v in init], ctx=ctx)
        print(f"    moves {s.stats['moves']} kernel {s.stats['kernel_ms']:.3f} ms", flush=True)
