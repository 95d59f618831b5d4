"""or_opt::solve — mirror of src/tsp/or_opt.rs:18-74 over tl_or_opt."""
import ctypes as C

import numpy as np


def solve(problem, opts=None, progress_tx=None, init_tour=None, *, ctx=None):
    """progress_tx: optional callable(kind, payload) — the reference's mpsc::Sender<ProgressMessage>.  The reference sends
    PathUpdate(path, 0.0) for the start path, PathUpdate(path, distances.tour_length(path)) after every apply_relocation and
    Done (or_opt.rs:40-42,62-67,70-72; nothing when n < 4, :31-34).  With a channel the solve goes through tl_or_opt_trace,
    which also returns the applied moves (i, j, seg_len, reversed) in order, and the same sequence is replayed afterwards."""
    from . import Solution, default_context
    from .. import _capi
    ctx = ctx or default_context()
    n = len(problem)
    init_pos = problem.positions_of(init_tour) if init_tour is not None else None
    packed = problem.explicit_packed()
    out = np.empty(n, dtype=np.uint32)
    cost = C.c_float()
    st = _capi.TlStats()
    args = (ctx.handle, problem.xy.ctypes.data_as(C.c_void_p), n,
            None if packed is None else packed.ctypes.data_as(C.c_void_p),
            None if init_pos is None else init_pos.ctypes.data_as(C.c_void_p),
            out.ctypes.data_as(C.c_void_p), C.byref(cost), C.byref(st))
    if progress_tx is None or n < 4:
        ctx.check(ctx.lib.tl_or_opt(*args))
    else:
        cap = max(64, 4 * n)
        while True:
            log = np.empty((cap, 4), dtype=np.uint32)
            ln = C.c_uint32()
            ctx.check(ctx.lib.tl_or_opt_trace(*args, log.ctypes.data_as(C.c_void_p), cap, C.byref(ln)))
            if ln.value <= cap:
                break
            cap = int(ln.value)  # deterministic: once more with room for every move
        start_pos = np.arange(n, dtype=np.uint32) if init_pos is None else np.asarray(init_pos, dtype=np.uint32)
        replay_progress(problem, start_pos, log[:ln.value], progress_tx)
    return Solution(cost.value, problem.ids[out], problem, st.as_dict())


def apply_relocation(tour, i, seg_len, j, reversed):
    """or_opt.rs:172-184 apply_relocation on a numpy tour; returns the new tour (drain the segment, insert it after the old
    index j, forward or reversed)."""
    seg = tour[i:i + seg_len]
    rest = np.concatenate([tour[:i], tour[i + seg_len:]])
    at = j - seg_len + 1 if j >= i + seg_len else j + 1
    return np.concatenate([rest[:at], seg[::-1] if reversed else seg, rest[at:]])


def tour_length_f32(problem, pos):
    """DistanceMatrix::tour_length (distance_matrix.rs:235-245) on the host in the reference's f32 order: the closing edge first,
    then every window in tour order, one rounding per addition — through problem.distances where the problem has a matrix,
    else KDPoint::distance (kdtree.rs:291-295)."""
    pos = np.asarray(pos, dtype=np.int64)
    if len(pos) < 2:
        return np.float32(0.0)
    a, b = np.concatenate([pos[-1:], pos[:-1]]), pos  # (last, first), (w0, w1), ...
    dm = problem.distances
    if dm is not None:
        hi, lo = np.maximum(a, b), np.minimum(a, b)
        d = np.where(hi == lo, np.float32(0.0), dm.items[np.maximum(hi * (hi - 1) // 2 + lo, 0)]).astype(np.float32)
    else:
        xy = problem.xy
        dx, dy = xy[a, 0] - xy[b, 0], xy[a, 1] - xy[b, 1]
        d = np.sqrt((dx * dx).astype(np.float32) + (dy * dy).astype(np.float32), dtype=np.float32)
    return np.float32(np.cumsum(d, dtype=np.float32)[-1])  # cumsum adds in order, one f32 rounding per term


def replay_progress(problem, start_pos, moves, progress_tx):
    """The reference's message stream (or_opt.rs:40-72) from the move list of tl_or_opt_trace."""
    tour = np.array(start_pos, dtype=np.uint32)
    ids = problem.ids
    progress_tx("PathUpdate", ([int(v) for v in ids[tour]], 0.0))
    for i, j, seg_len, rev in np.asarray(moves, dtype=np.int64).reshape(-1, 4):
        tour = apply_relocation(tour, int(i), int(seg_len), int(j), bool(rev))
        progress_tx("PathUpdate", ([int(v) for v in ids[tour]], float(tour_length_f32(problem, tour))))
    progress_tx("Done", None)
    return tour


def find_best_move(problem, path_pos, *, ctx=None):
    """or_opt::find_best_move (or_opt.rs:80-164) on positions: (delta, i, j, seg_len, reversed) or None."""
    from . import default_context
    ctx = ctx or default_context()
    n = len(problem)
    path = np.ascontiguousarray(path_pos, dtype=np.uint32)
    if len(path) != n:  # the library reads n u32 values: checked before the pointer is taken
        from .. import _capi
        raise _capi.TeelineGpuError(_capi.TL_ERR_BADARG, f"path has {len(path)} positions, the problem {n} cities")
    packed = problem.explicit_packed()
    found, rev = C.c_int(), C.c_int()
    i, j, seg = C.c_uint32(), C.c_uint32(), C.c_uint32()
    d = C.c_float()
    ctx.check(ctx.lib.tl_or_opt_find_best_move(ctx.handle, problem.xy.ctypes.data_as(C.c_void_p), n,
                                               None if packed is None else packed.ctypes.data_as(C.c_void_p),
                                               path.ctypes.data_as(C.c_void_p), C.byref(found), C.byref(d), C.byref(i),
                                               C.byref(j), C.byref(seg), C.byref(rev)))
    if not found.value:
        return None
    return (np.float32(d.value), i.value, j.value, seg.value, bool(rev.value))
