import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, teeline_amd as TA
with TA.Context(0) as ctx:
    for n in (1002, 10000, 13509, 30000):
        xy = TA.synth.synth_xy(n)
        best, tot = 1e9, []
        for rep in range(13):
            dm, ms = TA.distance_matrix.build(np.arange(n), xy, ctx=ctx, return_ms=True)
            if rep >= 3:
                best = min(best, ms); tot.append(ms)
        gb = n*(n-1)/2*4/1e9
        mean = sum(tot) / len(tot)
        print(f"dm_build n={n}: best {best*1e3:.1f} us  mean of 10 {mean*1e3:.1f} us  {gb/(mean*1e-3):.0f} GB/s = {gb/(mean*1e-3)/80:.1f}% of 8 TB/s (mean)")
