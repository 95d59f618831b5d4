import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch, teeline_amd as TA
n, R = 10000, 8
dev = torch.device("cuda", 0)
xy = TA.synth.synth_xy(n)
ctx = TA.Context(0)
d_xy = torch.from_numpy(xy).to(dev)
d_pos = torch.empty((R, n), dtype=torch.int32, device=dev); d_cost = torch.empty(R, dtype=torch.float32, device=dev)
d_st = torch.zeros((R, 16), dtype=torch.int64, device=dev)
s = torch.cuda.current_stream()
for rep in range(3):
    ctx.check(ctx.lib.tl_two_opt_batch_dev(ctx.handle, d_xy.data_ptr(), n, None, 12345, 0, R, 0, d_pos.data_ptr(), d_cost.data_ptr(), d_st.data_ptr(), C.c_void_p(s.cuda_stream)))
    torch.cuda.synchronize()
    print(rep, "ms", round(ctx.last_kernel_ms(),2), "sweeps", d_st[:,0].tolist(), "moves", d_st[:,1].tolist(), "steps", d_st[:,4].tolist(), "cost", [round(float(c),1) for c in d_cost.tolist()])
