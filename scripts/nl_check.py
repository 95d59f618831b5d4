"""NL rows (neighbour lists in the late sweeps, csrc/two_opt_nl.hip) against the tile-only kernel: same tours, costs and counters for a
batch of seeded restarts, and the time of both.  R restarts from FIRST at n = N; FLAGS_B = extra create flags of the NL context."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, teeline_amd as TA
n, R, first = int(os.environ.get("N", 10000)), int(os.environ.get("R", 256)), int(os.environ.get("FIRST", 0))
reps = int(os.environ.get("REPS", 3))
dev = torch.device("cuda", 0)
xy = TA.synth.synth_xy(n)
d_xy = torch.from_numpy(xy).to(dev)
s = torch.cuda.current_stream()


def run(flags):
    ctx = TA.Context(0, flags)
    d_pos = torch.empty((R, n), dtype=torch.int32, device=dev); d_cost = torch.empty(R, dtype=torch.float32, device=dev)
    d_st = torch.zeros((R, 16), dtype=torch.int64, device=dev)
    ms = []
    for _ in range(reps):
        ctx.check(ctx.lib.tl_two_opt_batch_dev(ctx.handle, d_xy.data_ptr(), n, None, 12345, first, R, 0, d_pos.data_ptr(), d_cost.data_ptr(), d_st.data_ptr(), C.c_void_p(s.cuda_stream)))
        torch.cuda.synchronize()
        ms.append(ctx.last_kernel_ms())
    return d_pos.cpu().numpy(), d_cost.cpu().numpy(), d_st.cpu().numpy(), ms


extra = int(os.environ.get("FLAGS_B", 0))
pa, ca, sa, ma = run(TA.TL_FLAG_2OPT_NO_NL)
pb, cb, sb, mb = run(extra)
print(f"n={n} R={R} first={first}: tiles only {min(ma):.2f} ms {['%.2f' % m for m in ma]} | NL {min(mb):.2f} ms {['%.2f' % m for m in mb]}")
same = (pa == pb).all(axis=1) & (ca.view(np.uint32) == cb.view(np.uint32)) & (sa[:, :4] == sb[:, :4]).all(axis=1)
print(f"identical descents: {int(same.sum())} of {R}; sweeps {sa[:, 0].min()}..{sa[:, 0].max()}; steps mean {sa[:, 4].mean():.0f} vs {sb[:, 4].mean():.0f}")
cyc_a, cyc_b = sa[:, 9].astype(float), sb[:, 9].astype(float)
print(f"descent cycles mean/max: tiles {cyc_a.mean()/1e6:.1f} / {cyc_a.max()/1e6:.1f} M, NL {cyc_b.mean()/1e6:.1f} / {cyc_b.max()/1e6:.1f} M")
if not same.all():
    bad = np.nonzero(~same)[0][:8]
    for r in bad:
        print(f"  restart {first + r}: cost {ca[r]} vs {cb[r]}, stats {sa[r, :5]} vs {sb[r, :5]}")
    sys.exit(1)
