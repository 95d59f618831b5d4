import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch, teeline_amd as TA
dev = torch.device("cuda", 0)
s = torch.cuda.current_stream()
for n in (1002, 2500):
    xy = TA.synth.synth_xy(n); d_xy = torch.from_numpy(xy).to(dev)
    for R in (384, 512, 560, 640, 768, 900):
        d_pos = torch.empty((R, n), dtype=torch.int32, device=dev); d_cost = torch.empty(R, dtype=torch.float32, device=dev)
        d_st = torch.zeros((R, 16), dtype=torch.int64, device=dev)
        row=[]
        for name, flag in (("auto",0),("8w",TA.TL_FLAG_2OPT_NT512),("4w",TA.TL_FLAG_2OPT_NT256)):
            with TA.Context(0, flag) as ctx:
                ms=[]
                for _ in range(3):
                    ctx.check(ctx.lib.tl_two_opt_batch_dev(ctx.handle, d_xy.data_ptr(), n, None, 12345, 0, R, 0, d_pos.data_ptr(), d_cost.data_ptr(), d_st.data_ptr(), C.c_void_p(s.cuda_stream)))
                    torch.cuda.synchronize(); ms.append(ctx.last_kernel_ms())
                row.append(f"{name} {min(ms[1:]):6.2f}")
        print(f"n={n} R={R}: " + " | ".join(row))
