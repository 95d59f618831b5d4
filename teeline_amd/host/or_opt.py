"""or_opt::solve — mirror of src/tsp/or_opt.rs:18-74 over tl_or_opt."""
import ctypes as C

import numpy as np


def solve(problem, opts=None, progress_tx=None, init_tour=None, *, ctx=None):
    from . import Solution, default_context
    from .. import _capi
    ctx = ctx or default_context()
    n = len(problem)
    init_pos = problem.positions_of(init_tour) if init_tour is not None else None
    packed = problem.explicit_packed()
    out = np.empty(n, dtype=np.uint32)
    cost = C.c_float()
    st = _capi.TlStats()
    ctx.check(ctx.lib.tl_or_opt(ctx.handle, problem.xy.ctypes.data_as(C.c_void_p), n,
                                None if packed is None else packed.ctypes.data_as(C.c_void_p),
                                None if init_pos is None else init_pos.ctypes.data_as(C.c_void_p),
                                out.ctypes.data_as(C.c_void_p), C.byref(cost), C.byref(st)))
    route = problem.ids[out]
    if progress_tx is not None:
        progress_tx("PathUpdate", ([int(v) for v in route], float(cost.value)))
        progress_tx("Done", None)
    return Solution(cost.value, route, problem, st.as_dict())


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
