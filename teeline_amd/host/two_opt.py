"""two_opt::solve — mirror of src/tsp/two_opt.rs:7-67 over tl_two_opt."""
import ctypes as C

import numpy as np

from .. import _capi


def solve(problem, opts=None, progress_tx=None, init_tour=None, *, ctx=None, mode=_capi.TL_MODE_REF_ORDER):
    """problem: TspProblem; opts ignored like the reference's `_opts` (two_opt.rs:9); init_tour: city ids.

    progress_tx: optional callable(kind, payload) — the reference's mpsc::Sender<ProgressMessage> (only teeline-qt passes one).
    The reference sends PathUpdate(start, 0.0), CityChange(path[i]) per outer i of every sweep, PathUpdate(path, new_distance)
    per improving move and Done (two_opt.rs:22-24,30-32,53-56,63-65).  The descent is one kernel launch, so nothing can be sent
    while it runs; with a channel the descent goes through tl_two_opt_trace, which also returns the applied moves in the
    reference's order, and the SAME message sequence — every path, every new_distance — is replayed from it afterwards.
    Matrix problems (EXPLICIT / GEO) go through the same entry with problem.distances.  Where no move list is available
    (coordinates with n beyond the LDS-resident descent, BEST_SWEEP) the messages are the initial PathUpdate, one final
    PathUpdate and Done.
    """
    from . import Solution, default_context
    ctx = ctx or default_context()
    n = len(problem)
    init_pos = problem.positions_of(init_tour) if init_tour is not None else None
    packed = problem.explicit_packed()
    out = np.empty(n, dtype=np.uint32)
    cost = C.c_float()
    st = _capi.TlStats()
    start_pos = np.arange(n, dtype=np.uint32) if init_pos is None else np.asarray(init_pos, dtype=np.uint32)
    if progress_tx is not None:
        progress_tx("PathUpdate", ([int(v) for v in problem.ids[start_pos]], 0.0))
    traced = (progress_tx is not None and mode == _capi.TL_MODE_REF_ORDER and
              3 <= n <= (65535 if packed is not None else min(ctx.two_opt_lds_max_n(), 65535)))
    if traced:
        cap = max(64, 16 * n)
        while True:
            log = np.empty(cap, dtype=np.uint32)
            ln = C.c_uint32()
            ctx.check(ctx.lib.tl_two_opt_trace(ctx.handle, problem.xy.ctypes.data_as(C.c_void_p), n,
                                               None if packed is None else packed.ctypes.data_as(C.c_void_p),
                                               None if init_pos is None else init_pos.ctypes.data_as(C.c_void_p),
                                               out.ctypes.data_as(C.c_void_p), C.byref(cost), C.byref(st),
                                               log.ctypes.data_as(C.c_void_p), cap, C.byref(ln)))
            if ln.value <= cap:
                break
            cap = int(ln.value)  # the descent is deterministic: once more with room for every move
        replay_progress(problem, start_pos, log[:ln.value], st.sweeps, progress_tx)
    else:
        ctx.check(ctx.lib.tl_two_opt(ctx.handle, problem.xy.ctypes.data_as(C.c_void_p), n,
                                     None if packed is None else packed.ctypes.data_as(C.c_void_p),
                                     None if init_pos is None else init_pos.ctypes.data_as(C.c_void_p), int(mode),
                                     out.ctypes.data_as(C.c_void_p), C.byref(cost), C.byref(st)))
    route = problem.ids[out]
    if progress_tx is not None:
        if not traced:
            progress_tx("PathUpdate", ([int(v) for v in route], float(cost.value)))
        progress_tx("Done", None)
    return Solution(cost.value, route, problem, st.as_dict())


def replay_progress(problem, start_pos, move_log, sweeps, progress_tx):
    """The reference's messages between the initial PathUpdate and Done (two_opt.rs:26-61), rebuilt from the move list of
    tl_two_opt_trace ((i << 16) | j per move, 0xFFFFFFFF where a new sweep begins): per sweep and outer i a CityChange(path[i]);
    per move (i, j) the reversal of path[i+1..=j] and a
    PathUpdate(path, new_distance) with new_distance = d(p[i], p[j]) + d(p[i+1], p[j+1]) on the path BEFORE the move — the f32 sum
    the reference has just compared (:42-49), through problem.distances like the reference."""
    n = len(start_pos)
    path = np.array(start_pos, dtype=np.uint32)
    ids = problem.ids
    dm = problem.distances
    xy = problem.xy

    def dist(p, q):  # problem.distances, or KDPoint::distance (kdtree.rs:291-295) in f32: separate roundings, correctly rounded sqrt
        if dm is not None:
            return np.float32(dm.distance_by_pos(int(p), int(q)))
        dx, dy = xy[p, 0] - xy[q, 0], xy[p, 1] - xy[q, 1]
        return np.sqrt(np.float32(dx * dx) + np.float32(dy * dy), dtype=np.float32)

    MARK = 0xFFFFFFFF  # TL_TRACE_SWEEP: a new sweep begins
    moves = [int(w) for w in move_log]
    k = 0
    for _ in range(int(sweeps)):
        for i in range(n - 3):
            progress_tx("CityChange", int(ids[path[i]]))
            while k < len(moves) and moves[k] != MARK and (moves[k] >> 16) == i:
                j = moves[k] & 0xFFFF
                new_distance = np.float32(dist(path[i], path[j]) + dist(path[i + 1], path[j + 1]))
                path[i + 1:j + 1] = path[i + 1:j + 1][::-1].copy()
                progress_tx("PathUpdate", ([int(v) for v in ids[path]], float(np.float32(new_distance))))
                k += 1
        if k < len(moves) and moves[k] == MARK:
            k += 1
    assert k == len(moves), "move list not consumed: it does not belong to this start tour"
    return path


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
