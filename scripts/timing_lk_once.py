"""LK at n = 13 509 / 20 epochs once per process (tuning builds read TL_LK_* from the environment): kernel ms, rounds, us per round."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import teeline_amd as TA
n = int(sys.argv[1]) if len(sys.argv) > 1 else 13509
p = TA.TspProblem(np.arange(n), TA.synth.synth_xy(n))
opts = TA.LKOptions(TA.HeuristicOptions(epochs=20, platoo_epochs=10, n_nearest=5), 5)
flag = int(os.environ.get("TL_CREATE_FLAGS", "0"))
with TA.Context(0, flag) as ctx:
    best = min((TA.lin_kernighan.solve(p, opts, ctx=ctx, seed=1) for _ in range(3)), key=lambda s: s.stats["kernel_ms"])
print(f"persist_blocks={os.environ.get('TL_LK_PERSIST_BLOCKS', 'default')} flags={flag} n={n}: kernel {best.stats['kernel_ms']:.2f} ms, {best.stats['sweeps']} rounds, "
      f"{best.stats['kernel_ms'] * 1e3 / best.stats['sweeps']:.2f} us/round, cost {float(best.total):.5f}", flush=True)
