"""Kernel time of the 256-restart batch (n = 10^4) and of the two single descents for the library in TEELINE_GPU_LIB."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, teeline_amd as TA
n, R = 10000, 256
dev = torch.device("cuda", 0)
xy = TA.synth.synth_xy(n)
ctx = TA.Context(0, int(os.environ.get("FLAGS", 0)))
d_xy = torch.from_numpy(xy).to(dev)
d_pos = torch.empty((R, n), dtype=torch.int32, device=dev); d_cost = torch.empty(R, dtype=torch.float32, device=dev)
d_st = torch.zeros((R, 16), dtype=torch.int64, device=dev)
s = torch.cuda.current_stream()
ms = []
for _ in range(4):
    ctx.check(ctx.lib.tl_two_opt_batch_dev(ctx.handle, d_xy.data_ptr(), n, None, 12345, 0, R, 0, d_pos.data_ptr(), d_cost.data_ptr(), d_st.data_ptr(), C.c_void_p(s.cuda_stream)))
    torch.cuda.synchronize(); ms.append(ctx.last_kernel_ms())
crc = int(d_pos.to(torch.int64).sum().item()) ^ int(d_cost.view(torch.int32).to(torch.int64).sum().item())
if crc != 298053045692 or int(d_st[:, 3].max().item()) != 0:  # the committed kernel's tours (sweep-bounded run: a wrong variant cannot hang)
    print(f"{os.path.basename(os.environ.get('TEELINE_GPU_LIB','default')):20s} WRONG RESULTS check {crc} status {int(d_st[:, 3].max().item())}")
    sys.exit(0)
prob = TA.TspProblem(np.arange(n), xy)
nn = TA.nearest_neighbor.solve(prob, ctx=ctx)
a = min(TA.two_opt.solve(prob, None, None, nn.route(), ctx=ctx).stats["kernel_ms"] for _ in range(3))
init = [int(v) for v in TA.synth.restart_perm(n, 12345, 0)]
b = min(TA.two_opt.solve(prob, None, None, init, ctx=ctx).stats["kernel_ms"] for _ in range(3))
crc = int(d_pos.to(torch.int64).sum().item()) ^ int(d_cost.view(torch.int32).to(torch.int64).sum().item())
print(f"{os.path.basename(os.environ.get('TEELINE_GPU_LIB','default')):20s} flags {os.environ.get('FLAGS', 0)} batch256 {min(ms[1:]):8.2f} ms  nn-start {a:6.2f} ms  random-start {b:7.2f} ms  steps/descent {d_st[:,4].float().mean().item():.0f}  check {crc}")
