"""three_opt::solve — mirror of src/tsp/three_opt.rs:16-51 over tl_three_opt."""
import ctypes as C

import numpy as np


def solve(problem, opts=None, progress_tx=None, init_tour=None, *, ctx=None):
    """progress_tx: optional callable(kind, payload) — the reference's mpsc::Sender<ProgressMessage>.  The reference sends
    PathUpdate(path, 0.0) for the start path and after every apply_3opt, then Done (three_opt.rs:34,42,47-49; nothing when
    n < 4, :25-28).  With a channel the solve goes through tl_three_opt_trace, which also returns the applied moves
    (i, j, k, case) in order, and the same sequence of paths is replayed from it afterwards (replay_progress)."""
    from . import Solution, default_context
    ctx = ctx or default_context()
    n = len(problem)
    init_pos = problem.positions_of(init_tour) if init_tour is not None else None
    packed = problem.explicit_packed()
    out = np.empty(n, dtype=np.uint32)
    cost = C.c_float()
    from .. import _capi
    st = _capi.TlStats()
    args = (ctx.handle, problem.xy.ctypes.data_as(C.c_void_p), n,
            None if packed is None else packed.ctypes.data_as(C.c_void_p),
            None if init_pos is None else init_pos.ctypes.data_as(C.c_void_p),
            out.ctypes.data_as(C.c_void_p), C.byref(cost), C.byref(st))
    if progress_tx is None or n < 4:
        ctx.check(ctx.lib.tl_three_opt(*args))
    else:
        cap = max(64, 4 * n)
        while True:
            log = np.empty((cap, 4), dtype=np.uint32)
            ln = C.c_uint32()
            ctx.check(ctx.lib.tl_three_opt_trace(*args, log.ctypes.data_as(C.c_void_p), cap, C.byref(ln)))
            if ln.value <= cap:
                break
            cap = int(ln.value)  # deterministic: once more with room for every move
        start_pos = np.arange(n, dtype=np.uint32) if init_pos is None else np.asarray(init_pos, dtype=np.uint32)
        replay_progress(problem, start_pos, log[:ln.value], progress_tx)
    return Solution(cost.value, problem.ids[out], problem, st.as_dict())


def apply_3opt(path, i, j, k, case):
    """three_opt.rs:186-218 apply_3opt on a numpy path, in place: cases 1-3 reverse segments, 4-7 swap path[i+1..=j] and
    path[j+1..=k] with either reversed."""
    s1, s2 = path[i + 1:j + 1].copy(), path[j + 1:k + 1].copy()
    if case == 1:
        path[i + 1:j + 1] = s1[::-1]
    elif case == 2:
        path[j + 1:k + 1] = s2[::-1]
    elif case == 3:
        path[i + 1:j + 1] = s1[::-1]
        path[j + 1:k + 1] = s2[::-1]
    elif case in (4, 5, 6, 7):
        a = s2[::-1] if case in (6, 7) else s2
        b = s1[::-1] if case in (5, 7) else s1
        path[i + 1:k + 1] = np.concatenate([a, b])
    else:
        raise ValueError(f"apply_3opt: case must be 1-7, got {case}")


def replay_progress(problem, start_pos, moves, progress_tx):
    """The reference's message stream (three_opt.rs:34-49) from the move list of tl_three_opt_trace."""
    path = np.array(start_pos, dtype=np.uint32)
    ids = problem.ids
    progress_tx("PathUpdate", ([int(v) for v in ids[path]], 0.0))
    for i, j, k, case in np.asarray(moves, dtype=np.int64).reshape(-1, 4):
        apply_3opt(path, int(i), int(j), int(k), int(case))
        progress_tx("PathUpdate", ([int(v) for v in ids[path]], 0.0))
    progress_tx("Done", None)
    return path


def find_best_move(problem, path_pos, *, ctx=None):
    """three_opt::find_best_move (three_opt.rs:58-131) on positions; returns (i, j, k, case, savings) or None."""
    from . import default_context
    ctx = ctx or default_context()
    n = len(problem)
    path = np.ascontiguousarray(path_pos, dtype=np.uint32)
    if len(path) != n:  # the library reads n u32 values: checked before the pointer is taken
        from .. import _capi
        raise _capi.TeelineGpuError(_capi.TL_ERR_BADARG, f"path has {len(path)} positions, the problem {n} cities")
    packed = problem.explicit_packed()
    found, kase = C.c_int(), C.c_int()
    i, j, k = C.c_uint32(), C.c_uint32(), C.c_uint32()
    sav = C.c_float()
    ctx.check(ctx.lib.tl_three_opt_find_best_move(ctx.handle, problem.xy.ctypes.data_as(C.c_void_p), n,
                                                  None if packed is None else packed.ctypes.data_as(C.c_void_p),
                                                  path.ctypes.data_as(C.c_void_p), C.byref(found), C.byref(i), C.byref(j),
                                                  C.byref(k), C.byref(kase), C.byref(sav)))
    if not found.value:
        return None
    return (i.value, j.value, k.value, kase.value, np.float32(sav.value))
