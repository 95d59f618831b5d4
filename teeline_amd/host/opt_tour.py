"""`.opt.tour` reader — mirror of src/tsp/opt_tour.rs:12-107 — and the optimal-tour comparison of teeline-cli
(teeline-cli/src/main.rs:654-698)."""
import re

import numpy as np

_KV = re.compile(r"^(?P<key>\w+)\s*:\s*(?P<val>.+)$")


class OptTour:
    def __init__(self, name, comment, dimension, route):
        self.name, self.comment, self.dimension, self.route = name, comment, dimension, route


def read_from_str(text):
    meta, route, state = {}, [], "header"
    for raw in text.splitlines():
        line = raw.strip().upper()
        if line == "EOF" or state == "end":
            break
        if state == "header":
            if line == "TOUR_SECTION":
                state = "tour"
            else:
                m = _KV.match(line)
                if m:
                    meta[m["key"]] = m["val"].strip()
        else:
            for tok in line.split():
                try:
                    v = int(tok)
                except ValueError:
                    continue
                if v == -1:
                    state = "end"
                    break
                if v > 0:
                    route.append(v)
    ty = meta.get("TYPE", "")
    if ty != "TOUR":
        raise ValueError(f"opt_tour: expected TYPE : TOUR, found TYPE : {ty}")
    try:
        dimension = int(meta.get("DIMENSION", "0").strip())
    except ValueError:
        dimension = 0
    if len(route) != dimension:
        raise ValueError(f"opt_tour: dimension mismatch — DIMENSION={dimension} but parsed {len(route)} cities")
    return OptTour(meta.get("NAME", "unknown"), meta.get("COMMENT", ""), dimension, route)


def read_from_file(path):
    try:
        with open(path) as fh:
            return read_from_str(fh.read())
    except OSError as e:
        raise ValueError(f"opt_tour: cannot open file: {e}") from None


def compute_optimal_comparison(solver_cost, problem, opt, ctx=None):
    """main.rs:660-684: (optimal_cost, gap_pct, name) in f32 arithmetic, or None on a dimension mismatch."""
    from . import default_context
    import ctypes as C
    if opt.dimension != len(problem):
        return None
    ctx = ctx or default_context()
    m = problem.id2pos()
    optimal = np.float32(0.0)
    if len(opt.route) >= 2 and all(int(v) in m for v in opt.route):  # tour_length: 0.0 on an unknown id (:221-233)
        pos = np.asarray([m[int(v)] for v in opt.route], dtype=np.uint32)
        packed = problem.explicit_packed()
        out = C.c_float()
        ctx.check(ctx.lib.tl_tour_length(ctx.handle, None if packed is not None else problem.xy.ctypes.data_as(C.c_void_p),
                                         None if packed is None else packed.ctypes.data_as(C.c_void_p), len(pos),
                                         pos.ctypes.data_as(C.c_void_p), C.byref(out)))
        optimal = np.float32(out.value)
    gap = (np.float32(solver_cost) - optimal) / optimal * np.float32(100.0) if optimal > 0 else np.float32(0.0)
    return optimal, np.float32(gap), opt.name
