"""Cascade work of one descent (restart FIRST, n = 10^4) with and without the late phase: rows, tile passes, L0 / L1 / L2 / L3 counts."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, teeline_amd as TA
n, first = int(os.environ.get("N", 10000)), int(os.environ.get("FIRST", 105))
dev = torch.device("cuda", 0)
xy = TA.synth.synth_xy(n)
d_xy = torch.from_numpy(xy).to(dev)
s = torch.cuda.current_stream()
for flags in (TA.TL_FLAG_2OPT_NO_NL | TA.TL_FLAG_COUNT_WORK, TA.TL_FLAG_COUNT_WORK):
    ctx = TA.Context(0, flags)
    d_pos = torch.empty((1, n), dtype=torch.int32, device=dev); d_cost = torch.empty(1, dtype=torch.float32, device=dev)
    d_st = torch.zeros((1, 16), dtype=torch.int64, device=dev)
    ctx.check(ctx.lib.tl_two_opt_batch_dev(ctx.handle, d_xy.data_ptr(), n, None, 12345, first, 1, 0, d_pos.data_ptr(), d_cost.data_ptr(), d_st.data_ptr(), C.c_void_p(s.cuda_stream)))
    torch.cuda.synchronize()
    st = d_st.cpu().numpy()[0]
    print(f"flags {flags:#x}: sweeps {st[0]} moves {st[1]} steps {st[4]} | L0 {st[5]:.3e} L1 {st[6]:.3e} L2 {st[7]:.3e} L3 {st[8]} | pruned rows {st[11]} tile passes {st[12]} | cycles {st[9]/1e6:.1f} M late {(st[13] >> 24)/1e6:.1f} M late steps {st[14] >> 32}")
