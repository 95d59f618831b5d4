"""Re-entrancy of the C ABI (-m gpu): SURVEY.md §8(b) "solver must be re-entrant ... contexts are independent".

The reference's callers run solvers on arbitrary threads (teeline-api/src/services/tsp_service.rs:295,328 `spawn_blocking`;
teeline-qt/src/solver_engine.rs:412-432 worker thread).  tests/probes/thread_campaign.py drives 8 host threads, each with its
own tl_ctx, through 2-opt at four sizes (different LDS sizes of one kernel in flight at once: csrc/tl_kernels.h
allow_max_lds), the matrix form, BEST_SWEEP, 3-opt, Or-opt, LK (hipGraph capture beside other threads' allocations), the NN
seed, candidate lists, tl_dm_build and multi-start for >= 20 s and compares every result bit for bit with the oracle's; then
two threads share ONE context (right or TL_ERR_BUSY, never wrong).  Once on the product library, once on the race-stress build
(-DTL_JITTER).  Each in a child process, because a process binds one library."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROBE = os.path.join(ROOT, "tests", "probes", "thread_campaign.py")


@pytest.mark.parametrize("lib,seconds", [("libteeline_gpu.so", 10), ("libteeline_gpu_jitter.so", 10)])
def test_eight_threads_own_contexts_mixed_solvers(lib, seconds):
    path = os.path.join(ROOT, "teeline_amd", lib)
    assert os.path.exists(path), "built by __graft_entry__.build()"
    env = dict(os.environ, TEELINE_GPU_LIB=path)
    r = subprocess.run([sys.executable, PROBE, str(seconds), "8"], env=env, capture_output=True, text=True, timeout=seconds * 10 + 300)
    out = r.stdout[-4000:] + r.stderr[-3000:]
    assert r.returncode == 0, out
    m = re.search(r"own contexts: 8 threads, (\d+) runs \((\d+)\.\.(\d+) per thread\), (\d+) mismatches, (\d+) s", r.stdout)
    assert m, out
    assert int(m.group(4)) == 0, out
    assert int(m.group(2)) >= 14, "every thread ran every job at least once"
    assert int(m.group(5)) >= seconds, out
    s = re.search(r"shared context: (\d+) right, (\d+) TL_ERR_BUSY, (\d+) wrong, usable afterwards: True", r.stdout)
    assert s and int(s.group(1)) > 0 and int(s.group(3)) == 0, out


def test_a_busy_context_refuses_a_second_thread(ctx):
    """Deterministic form of the shared-context rule: while one thread is inside tl_two_opt (an n = 10^4 descent, ~100 ms) a
    second thread's call on the same context returns TL_ERR_BUSY without touching it; afterwards the context works."""
    import threading
    import time

    import numpy as np

    import _oracle as O
    import teeline_amd as TA
    n = 10000
    xy = O.synth_xy(n)
    prob = TA.TspProblem(np.arange(n), xy)
    init = [int(v) for v in O.restart_perm(n, 12345, 2)]
    res = {}

    def long_call():
        res["sol"] = TA.two_opt.solve(prob, None, None, init, ctx=ctx)

    t = threading.Thread(target=long_call)
    small = TA.TspProblem(np.arange(60), O.synth_xy(60, seed=1))
    codes = []
    t.start()
    t0 = time.time()
    while t.is_alive() and time.time() - t0 < 30:
        try:
            TA.two_opt.solve(small, ctx=ctx)
            codes.append(0)
        except TA._capi.TeelineGpuError as exc:
            codes.append(exc.code)
    t.join()
    assert TA._capi.TL_ERR_BUSY in codes, codes[:20]
    assert set(codes) <= {0, TA._capi.TL_ERR_BUSY}
    # neither call was disturbed: the long one is the golden descent, and the context is usable
    import json
    import zlib
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "goldens_large.json")))["synthetic10000_seed12345"]["restarts"]["2"]
    assert f"{float(res['sol'].total):.5f}" == g["cost"]
    assert int(zlib.crc32(np.asarray(res["sol"].route(), dtype="<u4").tobytes())) == g["route_crc32"]
    rc, route, cost, st = O.two_opt(small.xy, None, 60)
    s2 = TA.two_opt.solve(small, ctx=ctx)
    assert list(s2.route()) == route.tolist() and np.float32(s2.total).tobytes() == np.float32(cost).tobytes()
