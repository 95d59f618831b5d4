"""two_opt::solve — mirror of src/tsp/two_opt.rs:7-67 over tl_two_opt."""
import ctypes as C

import numpy as np

from .. import _capi


def solve(problem, opts=None, progress_tx=None, init_tour=None, *, ctx=None, mode=_capi.TL_MODE_REF_ORDER):
    """problem: TspProblem; opts ignored like the reference's `_opts` (two_opt.rs:9); init_tour: city ids.

    progress_tx: optional callable(kind, payload).  The reference sends PathUpdate at start, CityChange per
    outer i, PathUpdate per move and Done (two_opt.rs:22-24,30-32,53-56,63-65); the GPU path coarsens this to
    the initial PathUpdate, one final PathUpdate and Done (documented difference, only visible in the Qt UI).
    """
    from . import Solution, default_context
    ctx = ctx or default_context()
    n = len(problem)
    init_pos = problem.positions_of(init_tour) if init_tour is not None else None
    if progress_tx is not None:
        start = [int(v) for v in (init_tour if init_tour is not None else problem.ids)]
        progress_tx("PathUpdate", (start, 0.0))
    packed = problem.explicit_packed()
    out = np.empty(n, dtype=np.uint32)
    cost = C.c_float()
    st = _capi.TlStats()
    ctx.check(ctx.lib.tl_two_opt(ctx.handle, problem.xy.ctypes.data_as(C.c_void_p), n,
                                 None if packed is None else packed.ctypes.data_as(C.c_void_p),
                                 None if init_pos is None else init_pos.ctypes.data_as(C.c_void_p), int(mode),
                                 out.ctypes.data_as(C.c_void_p), C.byref(cost), C.byref(st)))
    route = problem.ids[out]
    if progress_tx is not None:
        progress_tx("PathUpdate", ([int(v) for v in route], float(cost.value)))
        progress_tx("Done", None)
    return Solution(cost.value, route, problem, st.as_dict())


def multistart(problem, restarts, seed=0, first=0, *, ctx=None, mode=_capi.TL_MODE_REF_ORDER, return_costs=False):
    """Multi-start 2-opt (north-star config 4; no counterpart in the reference): restarts
    [first, first+restarts) from seeded Fisher–Yates permutations, one descent per CU."""
    from . import Solution, default_context
    ctx = ctx or default_context()
    n = len(problem)
    out = np.empty(n, dtype=np.uint32)
    costs = np.empty(restarts, dtype=np.float32)
    cost, best = C.c_float(), C.c_uint32()
    st = _capi.TlStats()
    ctx.check(ctx.lib.tl_two_opt_multistart(ctx.handle, problem.xy.ctypes.data_as(C.c_void_p), n, int(seed), int(first),
                                            int(restarts), int(mode), out.ctypes.data_as(C.c_void_p), C.byref(cost),
                                            C.byref(best), costs.ctypes.data_as(C.c_void_p), C.byref(st)))
    stats = st.as_dict()
    stats["best_restart"] = best.value
    sol = Solution(cost.value, problem.ids[out], problem, stats)
    return (sol, costs) if return_costs else sol


def multistart_devices(problem, restarts, contexts, seed=0, first=0, *, mode=_capi.TL_MODE_REF_ORDER, return_costs=False):
    """tl_two_opt_multistart_devices: the same multi-start job dealt over several contexts (one per GPU of the node) from
    one process; the result does not depend on how many contexts share the restarts."""
    from . import Solution
    n = len(problem)
    out = np.empty(n, dtype=np.uint32)
    costs = np.empty(restarts, dtype=np.float32)
    cost, best = C.c_float(), C.c_uint32()
    st = _capi.TlStats()
    arr = (C.c_void_p * len(contexts))(*[c.handle for c in contexts])
    c0 = contexts[0]
    c0.check(c0.lib.tl_two_opt_multistart_devices(arr, len(contexts), problem.xy.ctypes.data_as(C.c_void_p), n, int(seed), int(first),
                                                  int(restarts), int(mode), out.ctypes.data_as(C.c_void_p), C.byref(cost),
                                                  C.byref(best), costs.ctypes.data_as(C.c_void_p), C.byref(st)))
    stats = st.as_dict()
    stats["best_restart"] = best.value
    sol = Solution(cost.value, problem.ids[out], problem, stats)
    return (sol, costs) if return_costs else sol


def solve_population(problem, init_tours, *, ctx=None):
    """Refine a population of tours (lists of city ids), each by its own two_opt::solve descent, all concurrently
    (tl_two_opt_population).  Returns one Solution per tour; Solution k equals solve(problem, None, None, init_tours[k])."""
    from . import Solution, default_context
    ctx = ctx or default_context()
    n = len(problem)
    count = len(init_tours)
    init = np.empty((count, n), dtype=np.uint32)
    for k, tour in enumerate(init_tours):
        init[k] = problem.positions_of(tour)
    packed = problem.explicit_packed()
    out = np.empty((count, n), dtype=np.uint32)
    costs = np.empty(count, dtype=np.float32)
    st = _capi.TlStats()
    ctx.check(ctx.lib.tl_two_opt_population(ctx.handle, problem.xy.ctypes.data_as(C.c_void_p), n,
                                            None if packed is None else packed.ctypes.data_as(C.c_void_p),
                                            init.ctypes.data_as(C.c_void_p), count, out.ctypes.data_as(C.c_void_p),
                                            costs.ctypes.data_as(C.c_void_p), C.byref(st)))
    stats = st.as_dict()
    return [Solution(float(costs[k]), problem.ids[out[k]], problem, stats) for k in range(count)]
