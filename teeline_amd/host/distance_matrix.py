"""DistanceMatrix — mirror of src/tsp/distance_matrix.rs (packed strict lower triangle, f32).

`from_cities` / `build` run the HIP matrix-build kernel (tl_dm_build); lookups are host-side index
arithmetic on the packed array (distance_matrix.rs:177-191); `tour_length` runs on the device in the
reference's summation order (distance_matrix.rs:235-245).
"""
import ctypes as C

import numpy as np

from .. import _capi


class DistanceMatrix:
    def __init__(self, n, items, ids, kind="explicit"):
        # DistanceMatrix::new (distance_matrix.rs:96-115)
        items = np.ascontiguousarray(items, dtype=np.float32)
        if items.shape[0] != n * (n - 1) // 2:
            raise ValueError(f"distances length {items.shape[0]} != n*(n-1)/2={n * (n - 1) // 2} for n={n}")
        self.n = int(n)
        self.items = items
        self.ids = np.ascontiguousarray(ids, dtype=np.int64)
        self.kind = kind
        self._id2pos = {int(v): p for p, v in enumerate(self.ids)}
        if len(self._id2pos) != self.n:
            raise ValueError("city_idx size differs from n cities")

    def __len__(self):
        return self.items.shape[0]

    def num_cities(self):
        return self.n

    def distances(self):
        return self.items

    def distance_by_pos(self, p, q):  # :177-191
        if p == q:
            return np.float32(0.0)
        if p >= self.n or q >= self.n or p < 0 or q < 0:
            raise IndexError("position out of range")
        a, b = (p, q) if p > q else (q, p)
        return self.items[a * (a - 1) // 2 + b]

    def distance_between(self, id1, id2):  # :197-212
        if id1 == id2:
            return np.float32(0.0)
        try:
            return self.distance_by_pos(self._id2pos[int(id1)], self._id2pos[int(id2)])
        except KeyError:
            raise KeyError("city_id not in index") from None

    def city_id2pos(self, cid):
        return self._id2pos.get(int(cid))

    def tour_length_by_pos(self, path, ctx=None):  # :235-245
        from . import default_context
        path = np.ascontiguousarray(path, dtype=np.uint32)
        if len(path) < 2:
            return np.float32(0.0)
        ctx = ctx or default_context()
        out = C.c_float()
        ctx.check(ctx.lib.tl_tour_length(ctx.handle, None, self.items.ctypes.data_as(C.c_void_p), self.n,
                                         path.ctypes.data_as(C.c_void_p), C.byref(out)))
        return np.float32(out.value)

    def tour_length(self, path_ids, ctx=None):  # :221-233 unknown id -> 0.0
        if len(path_ids) < 2:
            return np.float32(0.0)
        pos = [self._id2pos.get(int(v)) for v in path_ids]
        if any(p is None for p in pos):
            return np.float32(0.0)
        return self.tour_length_by_pos(np.asarray(pos, dtype=np.uint32), ctx)


def build(ids, xy, kind="euc2d", ctx=None, return_ms=False):
    """DistanceMatrix::build (distance_matrix.rs:122-153) on the GPU."""
    from . import default_context
    xy = np.ascontiguousarray(xy, dtype=np.float32).reshape(-1, 2)
    n = xy.shape[0]
    if n < 2:
        raise ValueError("distance matrix requires at least 2 points")
    if kind not in ("euc2d", "geo"):
        raise ValueError("cannot build distance matrix from coordinates for EXPLICIT type — use DistanceMatrix(...)")
    ctx = ctx or default_context()
    out = np.empty(n * (n - 1) // 2, dtype=np.float32)
    ms = C.c_double()
    ctx.check(ctx.lib.tl_dm_build(ctx.handle, xy.ctypes.data_as(C.c_void_p), n,
                                  _capi.TL_DIST_GEO if kind == "geo" else _capi.TL_DIST_EUC2D,
                                  _capi.TL_DM_PACKED_LOWER, out.ctypes.data_as(C.c_void_p), C.byref(ms)))
    dm = DistanceMatrix(n, out, ids, kind)
    return (dm, ms.value) if return_ms else dm


def from_cities(cities, ctx=None):
    """distance_matrix::from_cities (distance_matrix.rs:78-80)."""
    return build([c.id for c in cities], [[c.coords[0], c.coords[1]] for c in cities], "euc2d", ctx)


def is_euc2d(xy, items, ctx=None):
    """tl_dm_is_euc2d: does the packed matrix `items` hold exactly the EUC_2D distances of xy?  (The reference's
    DistanceMatrix has no distance type: distance_matrix.rs:86-93.)"""
    from . import default_context
    ctx = ctx or default_context()
    xy = np.ascontiguousarray(xy, dtype=np.float32).reshape(-1, 2)
    items = np.ascontiguousarray(items, dtype=np.float32)
    n = xy.shape[0]
    if items.shape[0] != n * (n - 1) // 2:
        raise ValueError("distances length != n*(n-1)/2")
    out = C.c_int()
    ctx.check(ctx.lib.tl_dm_is_euc2d(ctx.handle, xy.ctypes.data_as(C.c_void_p), items.ctypes.data_as(C.c_void_p), n, C.byref(out)))
    return bool(out.value)
