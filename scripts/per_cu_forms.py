"""Restarts per second of the LDS 2-opt batch with 1, 2 or 4 descents per CU (16-, 8-, 4-wave forms): which form for which n?"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, teeline_amd as TA
dev = torch.device("cuda", 0)
s = torch.cuda.current_stream()
R = 1024
for n in (500, 1002, 2000, 3000, 5000, 7000):
    xy = TA.synth.synth_xy(n)
    d_xy = torch.from_numpy(xy).to(dev)
    d_pos = torch.empty((R, n), dtype=torch.int32, device=dev); d_cost = torch.empty(R, dtype=torch.float32, device=dev)
    d_st = torch.zeros((R, 16), dtype=torch.int64, device=dev)
    row = []
    ref = None
    for name, flag in (("auto", 0), ("8 waves", TA.TL_FLAG_2OPT_NT512), ("4 waves", TA.TL_FLAG_2OPT_NT256)):
        with TA.Context(0, flag) as ctx:
            ms = []
            for _ in range(3):
                ctx.check(ctx.lib.tl_two_opt_batch_dev(ctx.handle, d_xy.data_ptr(), n, None, 12345, 0, R, 0, d_pos.data_ptr(), d_cost.data_ptr(), d_st.data_ptr(), C.c_void_p(s.cuda_stream)))
                torch.cuda.synchronize(); ms.append(ctx.last_kernel_ms())
            chk = int(d_pos.to(torch.int64).sum().item()) ^ int(d_cost.view(torch.int32).to(torch.int64).sum().item())
            ref = chk if ref is None else ref
            row.append(f"{name} {min(ms[1:]):7.2f} ms{'' if chk == ref else ' DIFFERENT TOURS'}")
    # one descent per CU: four batches of 256
    with TA.Context(0) as ctx:
        ms = []
        for _ in range(3):
            ctx.check(ctx.lib.tl_two_opt_batch_dev(ctx.handle, d_xy.data_ptr(), n, None, 12345, 0, 256, 0, d_pos.data_ptr(), d_cost.data_ptr(), d_st.data_ptr(), C.c_void_p(s.cuda_stream)))
            torch.cuda.synchronize(); ms.append(ctx.last_kernel_ms())
        row.append(f"16 waves (4 x 256) ~{4 * min(ms[1:]):7.2f} ms")
    print(f"n={n:5d}, {R} restarts: " + " | ".join(row))
