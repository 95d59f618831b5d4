"""lin_kernighan::solve — mirror of src/tsp/lin_kernighan.rs:35-100 over tl_lk."""
import ctypes as C

import numpy as np


def solve(problem, opts=None, progress_tx=None, init_tour=None, *, ctx=None, seed=1):
    """progress_tx: optional callable(kind, payload) — the reference's mpsc::Sender<ProgressMessage>.  The reference sends
    PathUpdate(best_tour, best_dist) after the first lk_pass and after every epoch that improves on it, and no Done
    (lin_kernighan.rs:71,90; nothing when n < 4, :57-59).  With a channel the solve goes through tl_lk_live: the device-side state
    machine files exactly those tours and distances and the host hands them on while the search runs (tl_lk_trace lists the same
    ones after the fact)."""
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
    args = (ctx.handle, problem.xy.ctypes.data_as(C.c_void_p), n,
            None if packed is None else packed.ctypes.data_as(C.c_void_p),
            None if init_pos is None else init_pos.ctypes.data_as(C.c_void_p), C.byref(o), int(seed),
            out.ctypes.data_as(C.c_void_p), C.byref(cost), C.byref(st))
    if progress_tx is None or n < 4:
        ctx.check(ctx.lib.tl_lk(*args))
    else:
        # with a channel: tl_lk_live calls back WHILE the device-side search runs (between two polls of its state machine), once
        # per best tour the ILS settles on, in order — the messages arrive as the reference's would, not after the solve
        err = []

        def on_best(_user, pos, nn, best_dist):
            try:
                progress_tx("PathUpdate", ([int(v) for v in problem.ids[np.ctypeslib.as_array(pos, shape=(nn,))]], float(best_dist)))
            except BaseException as exc:  # an exception must not unwind through the C frames
                err.append(exc)

        cb = _capi.LK_PROGRESS_FN(on_best)
        ctx.check(ctx.lib.tl_lk_live(*args, cb, None))
        if err:
            raise err[0]
    return Solution(cost.value, problem.ids[out], problem, st.as_dict())


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
