"""How much does a second descent per CU buy?  At n = 7000 two descent workgroups fit one CU's LDS (2 x 79.5 KB): time of a
256-restart batch (one per CU) against a 512-restart batch (two per CU)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, teeline_amd as TA
dev = torch.device("cuda", 0)
ctx = TA.Context(0)
s = torch.cuda.current_stream()
for n in (7000, 7400, 10000):
    xy = TA.synth.synth_xy(n)
    d_xy = torch.from_numpy(xy).to(dev)
    for R in (256, 512):
        d_pos = torch.empty((R, n), dtype=torch.int32, device=dev); d_cost = torch.empty(R, dtype=torch.float32, device=dev)
        d_st = torch.zeros((R, 16), dtype=torch.int64, device=dev)
        ms = []
        for _ in range(3):
            ctx.check(ctx.lib.tl_two_opt_batch_dev(ctx.handle, d_xy.data_ptr(), n, None, 12345, 0, R, 0, d_pos.data_ptr(), d_cost.data_ptr(), d_st.data_ptr(), C.c_void_p(s.cuda_stream)))
            torch.cuda.synchronize(); ms.append(ctx.last_kernel_ms())
        sweeps = int(d_st[:, 0].sum().item())
        cands = sweeps * (n - 3) * (n - 2) // 2
        print(f"n={n} restarts={R}: {min(ms[1:]):8.2f} ms  {cands / (min(ms[1:]) * 1e-3):.3e} candidates/s")
