import os, sys, time, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import teeline_amd as TA
from teeline_amd import _capi
n, R = 10000, 256
dev = torch.device("cuda", 0)
xy = TA.synth.synth_xy(n); ctx = TA.Context(0); lib, h = ctx.lib, ctx.handle
d_xy = torch.from_numpy(xy).to(dev)
d_pos = torch.empty((R, n), dtype=torch.int32, device=dev); d_cost = torch.empty(R, dtype=torch.float32, device=dev)
d_stats = torch.zeros((R, _capi.TL_DEV_STATS_STRIDE), dtype=torch.int64, device=dev)
stream = torch.cuda.current_stream()
def launch():
    ctx.check(lib.tl_two_opt_batch_dev(h, d_xy.data_ptr(), n, None, 12345, 0, R, _capi.TL_MODE_REF_ORDER, d_pos.data_ptr(), d_cost.data_ptr(), d_stats.data_ptr(), C.c_void_p(stream.cuda_stream)))
launch(); torch.cuda.synchronize()
for rep in range(2):
    t0 = time.perf_counter(); launch(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    ms = ctx.last_kernel_ms(); t3 = time.perf_counter()
    s = int(d_stats[:, 0].sum().item()); t4 = time.perf_counter()
    k = TA.multistart.allreduce_best(TA.multistart.pack_keys(d_cost, 0), None); kk = k.item(); t5 = time.perf_counter()
    print(f"launch {1e3*(t1-t0):.2f} ms, sync {1e3*(t2-t1):.2f}, kernel_ms {ms:.2f}, last_kernel_ms call {1e3*(t3-t2):.2f}, stats sum {1e3*(t4-t3):.2f}, keys {1e3*(t5-t4):.2f}")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter(); e0.record(stream)
for _ in range(3): launch()
e1.record(stream); torch.cuda.synchronize(); t1 = time.perf_counter()
print(f"3 launches back-to-back: wall {1e3*(t1-t0)/3:.2f} ms/step, events {e0.elapsed_time(e1)/3:.2f} ms/step")
