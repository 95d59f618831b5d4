"""Two descents per CU (8 waves each) with and without the late phase: batches of 512 / 1024 seeded restarts at n = 1002 ... 5400."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, teeline_amd as TA
dev = torch.device("cuda", 0)
s = torch.cuda.current_stream()
for n in (500, 1002, 2000, 3000, 5000, 5400, 6000):
    xy = TA.synth.synth_xy(n)
    d_xy = torch.from_numpy(xy).to(dev)
    for R in (512, 1024):
        res = []
        for flags in (TA.TL_FLAG_2OPT_NO_NL, 0):
            with TA.Context(0, flags) as ctx:
                d_pos = torch.empty((R, n), dtype=torch.int32, device=dev); d_cost = torch.empty(R, dtype=torch.float32, device=dev)
                d_st = torch.zeros((R, 16), dtype=torch.int64, device=dev)
                ms = []
                for _ in range(3):
                    ctx.check(ctx.lib.tl_two_opt_batch_dev(ctx.handle, d_xy.data_ptr(), n, None, 12345, 0, R, 0, d_pos.data_ptr(), d_cost.data_ptr(), d_st.data_ptr(), C.c_void_p(s.cuda_stream)))
                    torch.cuda.synchronize(); ms.append(ctx.last_kernel_ms())
                res.append((min(ms[1:]), d_pos.cpu().numpy(), d_cost.cpu().numpy().view(np.uint32), d_st.cpu().numpy()))
        (ta, pa, ca, sa), (tb, pb, cb, sb) = res
        same = (pa == pb).all() and (ca == cb).all() and (sa[:, :4] == sb[:, :4]).all()
        print(f"n={n} restarts={R}: tile-only {ta:8.2f} ms | with the late phase {tb:8.2f} ms ({(sb[:, 14] >> 32).mean():.0f} late steps per descent) | same results: {same}")
