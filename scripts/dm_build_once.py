"""One packed distance-matrix build at n = 10^4 (plus one warm-up) — the target of the rocprofv3 PMC passes for k_dm_build_packed_blocked."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, teeline_amd as TA
n = int(os.environ.get("N", 10000))
xy = TA.synth.synth_xy(n)
with TA.Context(0) as ctx:
    for _ in range(2):
        dm, ms = TA.distance_matrix.build(np.arange(n), xy, ctx=ctx, return_ms=True)
    print(f"dm_build n={n}: {ms*1e3:.1f} us")
