"""lin_kernighan::solve — mirror of src/tsp/lin_kernighan.rs:35-100 over tl_lk."""
import ctypes as C

import numpy as np


def solve(problem, opts=None, progress_tx=None, init_tour=None, *, ctx=None, seed=1):
    from . import LKOptions, Solution, default_context
    from .. import _capi
    ctx = ctx or default_context()
    opts = opts or LKOptions()
    opts.validate()
    n = len(problem)
    init_pos = problem.positions_of(init_tour) if init_tour is not None else None
    o = _capi.TlLkOpts(opts.heuristic.epochs, opts.heuristic.platoo_epochs, opts.heuristic.n_nearest, opts.max_depth)
    out = np.empty(n, dtype=np.uint32)
    cost = C.c_float()
    st = _capi.TlStats()
    # GEO / EXPLICIT problems: the search is Euclidean over the city coordinates (lin_kernighan.rs:41 rebuilds its own
    # matrix), but the NN seed (:47-55) and the reported total (:99) go through problem.distances
    packed = problem.explicit_packed()
    ctx.check(ctx.lib.tl_lk(ctx.handle, problem.xy.ctypes.data_as(C.c_void_p), n,
                            None if packed is None else packed.ctypes.data_as(C.c_void_p),
                            None if init_pos is None else init_pos.ctypes.data_as(C.c_void_p), C.byref(o), int(seed),
                            out.ctypes.data_as(C.c_void_p), C.byref(cost), C.byref(st)))
    route = problem.ids[out]
    if progress_tx is not None:
        progress_tx("PathUpdate", ([int(v) for v in route], float(cost.value)))
    return Solution(cost.value, route, problem, st.as_dict())


def build_candidates(problem, k, *, ctx=None):
    """lin_kernighan::build_candidates (lin_kernighan.rs:12-27) in POSITIONS: n x min(k, n-1) array."""
    from . import default_context
    ctx = ctx or default_context()
    n = len(problem)
    kk = min(int(k), n - 1)
    out = np.empty((n, max(kk, 1)), dtype=np.uint32)
    ctx.check(ctx.lib.tl_build_candidates(ctx.handle, problem.xy.ctypes.data_as(C.c_void_p), n, int(k),
                                          out.ctypes.data_as(C.c_void_p)))
    return out[:, :kk]
