"""nearest_neighbor::solve — mirror of src/tsp/nearest_neighbor.rs:8-76 over tl_nearest_neighbor."""
import ctypes as C

import numpy as np


def solve(problem, opts=None, progress_tx=None, init_tour=None, *, ctx=None):
    """init_tour is ignored like the reference's `_init_tour` (nearest_neighbor.rs:12)."""
    from . import HeuristicOptions, Solution, default_context
    from .. import _capi
    ctx = ctx or default_context()
    opts = opts or HeuristicOptions()
    opts.validate()
    packed = problem.explicit_packed()  # GEO / EXPLICIT: the walk reads problem.distances (distance_matrix.rs:259-297)
    n = len(problem)
    out = np.empty(n, dtype=np.uint32)
    cost = C.c_float()
    ctx.check(ctx.lib.tl_nearest_neighbor(ctx.handle, problem.xy.ctypes.data_as(C.c_void_p),
                                          None if packed is None else packed.ctypes.data_as(C.c_void_p), n, int(opts.n_nearest),
                                          out.ctypes.data_as(C.c_void_p), C.byref(cost)))
    return Solution(cost.value, problem.ids[out], problem, {"kernel_ms": ctx.last_kernel_ms()})
