"""Cycle stamps of one random-restart descent (n = 10^4, restart FIRST) from the -DTL_PROFILE2 (control wave: per step type) / -DTL_PROFILE3 (a lead and a non-lead worker: per segment of a dense step) / -DTL_PROFILE4 (per sweep, printed by the device) builds: TEELINE_GPU_LIB=<that build> R=1 FIRST=152 python scripts/descent_profile.py"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, teeline_amd as TA
n, R = 10000, int(os.environ.get("R", 4))
FIRST = int(os.environ.get("FIRST", 0))
dev = torch.device("cuda", 0)
xy = TA.synth.synth_xy(n)
ctx = TA.Context(0)
d_xy = torch.from_numpy(xy).to(dev)
d_pos = torch.empty((R, n), dtype=torch.int32, device=dev); d_cost = torch.empty(R, dtype=torch.float32, device=dev)
d_st = torch.zeros((R, 16), dtype=torch.int64, device=dev)
s = torch.cuda.current_stream()
for _ in range(2):
    ctx.check(ctx.lib.tl_two_opt_batch_dev(ctx.handle, d_xy.data_ptr(), n, None, 12345, FIRST, R, 0, d_pos.data_ptr(), d_cost.data_ptr(), d_st.data_ptr(), C.c_void_p(s.cuda_stream)))
    torch.cuda.synchronize()
print("kernel ms", ctx.last_kernel_ms())
for r in range(R):
    print(FIRST + r, d_st[r].cpu().numpy().tolist())
