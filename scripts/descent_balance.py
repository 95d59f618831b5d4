"""Per-descent shader-clock counts of one 256-restart launch (d_out_stats word 9): how uneven are the descents the kernel waits for?"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, teeline_amd as TA
from teeline_amd import _capi
n, R = 10000, int(os.environ.get("R", 256))
dev = torch.device("cuda", 0)
xy = TA.synth.synth_xy(n)
ctx = TA.Context(0, int(os.environ.get("FLAGS", 0)))
d_xy = torch.from_numpy(xy).to(dev)
d_pos = torch.empty((R, n), dtype=torch.int32, device=dev); d_cost = torch.empty(R, dtype=torch.float32, device=dev)
d_st = torch.zeros((R, 16), dtype=torch.int64, device=dev)
s = torch.cuda.current_stream()
for _ in range(2):
    ctx.check(ctx.lib.tl_two_opt_batch_dev(ctx.handle, d_xy.data_ptr(), n, None, 12345, 0, R, 0, d_pos.data_ptr(), d_cost.data_ptr(), d_st.data_ptr(), C.c_void_p(s.cuda_stream)))
    torch.cuda.synchronize()
st = d_st.cpu().numpy()
cyc = st[:, 9].astype(np.float64); sw = st[:, 0]; mv = st[:, 1]; steps = st[:, 4]
print(f"kernel {ctx.last_kernel_ms():.2f} ms; descent cycles: mean {cyc.mean()/1e6:.1f} M  max {cyc.max()/1e6:.1f} M  min {cyc.min()/1e6:.1f} M  max/mean {cyc.max()/cyc.mean():.3f}")
print("sweeps histogram:", {int(k): int((sw == k).sum()) for k in np.unique(sw)})
print(f"cycles per sweep: {np.mean(cyc / sw)/1e6:.2f} M; corr(cycles, sweeps) = {np.corrcoef(cyc, sw)[0,1]:.3f}; steps mean {steps.mean():.0f}; moves mean {mv.mean():.0f}")
order = np.argsort(-cyc)[:5]
print("slowest:", [(int(r), int(sw[r]), round(cyc[r] / 1e6, 1)) for r in order])
if st[:, 13].max() > 0:  # role-split kernel diagnostics
    nd, npr, nfd = st[:, 13] & 0xFFFFFF, st[:, 14] & 0xFFFFFFFF, st[:, 15]
    late_clk, late_steps = (st[:, 13] >> 24).astype(np.float64), st[:, 14] >> 32
    if late_steps.max() > 0:
        print(f"late phase: cycles mean {late_clk.mean()/1e6:.1f} M max {late_clk.max()/1e6:.1f} M; steps mean {late_steps.mean():.0f}; cycles per late sweep {np.mean(late_clk / np.maximum(sw - 4, 1))/1e6:.2f} M (sweeps from 5 on)")
        for r in list(order):
            print(f"  restart {int(r)}: late {late_clk[r]/1e6:.1f} M cycles in {int(late_steps[r])} steps, {int(sw[r])} sweeps")
    print(f"descriptors mean {nd.mean():.0f} max {nd.max()}; pruned steps mean {npr.mean():.0f}; dense steps mean {(steps - npr).mean():.0f}; move-log words offered mean {nfd.mean():.1f}")
    for r in list(order) + list(np.argsort(cyc)[:3]):
        print(f"  restart {int(r)}: {cyc[r]/1e6:.1f} M cycles, sweeps {int(sw[r])}, steps {int(steps[r])} (pruned {int(npr[r])}), descriptors {int(nd[r])}, moves {int(mv[r])}")
