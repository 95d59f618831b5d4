"""nearest_neighbor::solve — mirror of src/tsp/nearest_neighbor.rs:8-76 over tl_nearest_neighbor."""
import ctypes as C

import numpy as np


def solve(problem, opts=None, progress_tx=None, init_tour=None, *, ctx=None):
    """init_tour is ignored like the reference's `_init_tour` (nearest_neighbor.rs:12).  progress_tx: optional
    callable(kind, payload); the reference's messages are replayed from the finished walk (replay_progress)."""
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
    route = problem.ids[out]
    if progress_tx is not None:
        replay_progress(route, progress_tx)
    return Solution(cost.value, route, problem, {"kernel_ms": ctx.last_kernel_ms()})


def replay_progress(route, progress_tx):
    """The reference's message stream (nearest_neighbor.rs:32-34,40-42,67-69,72-74) — it follows from the finished walk:
    PathUpdate([start], 0.0), then per step CityChange(current city) and PathUpdate(path so far, 0.0), then Done."""
    path = [int(v) for v in route]
    if not path:
        return
    progress_tx("PathUpdate", (path[:1], 0.0))
    for t in range(1, len(path)):
        progress_tx("CityChange", path[t - 1])
        progress_tx("PathUpdate", (path[:t + 1], 0.0))
    progress_tx("Done", None)
