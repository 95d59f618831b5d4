import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch, teeline_amd as TA
dev = torch.device("cuda", 0)
s = torch.cuda.current_stream()
for n in (3500, 4200):
    xy = TA.synth.synth_xy(n); d_xy = torch.from_numpy(xy).to(dev)
    R = 1024
    d_pos = torch.empty((R, n), dtype=torch.int32, device=dev); d_cost = torch.empty(R, dtype=torch.float32, device=dev)
    d_st = torch.zeros((R, 16), dtype=torch.int64, device=dev)
    row=[]; ref=None
    for name, flag in (("auto",0),("8w plain",TA.TL_FLAG_2OPT_NT512)):
        with TA.Context(0, flag) as ctx:
            ms=[]
            for _ in range(3):
                ctx.check(ctx.lib.tl_two_opt_batch_dev(ctx.handle, d_xy.data_ptr(), n, None, 12345, 0, R, 0, d_pos.data_ptr(), d_cost.data_ptr(), d_st.data_ptr(), C.c_void_p(s.cuda_stream)))
                torch.cuda.synchronize(); ms.append(ctx.last_kernel_ms())
            chk = int(d_pos.to(torch.int64).sum().item()) ^ int(d_cost.view(torch.int32).to(torch.int64).sum().item())
            ref = chk if ref is None else ref
            row.append(f"{name} {min(ms[1:]):6.2f}{'' if chk==ref else ' DIFFERENT'}")
    print(f"n={n} R={R}: " + " | ".join(row))
