"""Same batch through several instantiations of the LDS 2-opt kernel; prints which restarts differ (a race / logic hole finder).
TEELINE_GPU_LIB selects the library.  python scripts/ab_forms.py [n] [R]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, teeline_amd as TA
from teeline_amd import _capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
R = int(sys.argv[2]) if len(sys.argv) > 2 else 256
xy = TA.synth.synth_xy(n)
dev = torch.device("cuda", 0)
d_xy = torch.from_numpy(xy).to(dev)
s = torch.cuda.current_stream()
def run(flags, count=R):
    with TA.Context(0, flags) as c:
        d_pos = torch.empty((count, n), dtype=torch.int32, device=dev); d_cost = torch.empty(count, dtype=torch.float32, device=dev)
        d_st = torch.zeros((count, 16), dtype=torch.int64, device=dev)
        c.check(c.lib.tl_two_opt_batch_dev(c.handle, d_xy.data_ptr(), n, None, 12345, 0, count, 0, d_pos.data_ptr(), d_cost.data_ptr(), d_st.data_ptr(), C.c_void_p(s.cuda_stream)))
        torch.cuda.synchronize()
        return d_pos.cpu().numpy(), d_cost.cpu().numpy().view(np.uint32), d_st.cpu().numpy()[:, :3], c.last_kernel_ms()
forms = [("nl", 0), ("nl again", 0), ("count", TA.TL_FLAG_COUNT_WORK), ("no_nl", TA.TL_FLAG_2OPT_NO_NL), ("nt512", TA.TL_FLAG_2OPT_NT512), ("no_prune", TA.TL_FLAG_NO_PRUNE)]
ref = None
for name, fl in forms:
    p, c, st, ms = run(fl)
    if ref is None:
        ref = (p, c, st)
    bad = [r for r in range(R) if not (np.array_equal(p[r], ref[0][r]) and c[r] == ref[1][r] and np.array_equal(st[r], ref[2][r]))]
    print(f"{name:10s} {ms:8.2f} ms  differing restarts vs the first form: {bad[:12]}{' ...' if len(bad) > 12 else ''} ({len(bad)})", flush=True)
