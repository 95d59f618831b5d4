"""Host-side mirror of the reference's solver interface for the 2-opt / 3-opt / LK path.

Names, argument meaning and error behaviour follow the reference (file:line cited per item); the
work is done by libteeline_gpu.so through its C ABI.
"""
import ctypes as C
import threading

import numpy as np

from .. import _capi
from .._capi import ReferencePanics, TeelineGpuError


class Context:
    """One tl_ctx: a HIP stream + device workspace on one GPU.  Not thread-safe; make one per thread
    (the reference calls solvers from arbitrary threads, teeline-api/src/services/tsp_service.rs:295)."""

    def __init__(self, device=0, flags=_capi.TL_FLAG_NONE):
        self._lib = _capi.load()
        h = C.c_void_p()
        rc = self._lib.tl_create(int(device), int(flags), C.byref(h))
        if rc != _capi.TL_OK:
            raise TeelineGpuError(rc, self._lib.tl_last_error(None).decode())
        self._h = h
        self.device = device
        self.flags = flags

    def close(self):
        if getattr(self, "_h", None):
            self._lib.tl_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    @property
    def handle(self):
        return self._h

    @property
    def lib(self):
        return self._lib

    def check(self, rc):
        if rc == _capi.TL_OK:
            return
        if rc == _capi.TL_ERR_BUSY:  # the context's error string belongs to the thread that is inside: do not read it
            raise TeelineGpuError(rc, "the context is in use by another thread (one tl_ctx per thread)")
        msg = self._lib.tl_last_error(self._h).decode()
        if rc == _capi.TL_ERR_REF_PANICS:
            raise ReferencePanics(rc, msg)
        raise TeelineGpuError(rc, msg)

    def device_info(self):
        cus, lds = C.c_int(), C.c_int()
        arch = C.create_string_buffer(64)
        self.check(self._lib.tl_device_info(self._h, C.byref(cus), C.byref(lds), arch, 64))
        return {"cus": cus.value, "lds_bytes": lds.value, "arch": arch.value.decode()}

    def two_opt_lds_max_n(self):
        return int(self._lib.tl_two_opt_lds_max_n(self._h))

    def last_kernel_ms(self):
        ms = C.c_double()
        self.check(self._lib.tl_last_kernel_ms(self._h, C.byref(ms)))
        return ms.value

    def two_opt_last_counters(self):
        """tl_two_opt_last_counters: the 16 kernel-side counters of descent 0 of the most recent host-buffer 2-opt call (diagnostics)."""
        out = (C.c_uint64 * 16)()
        self.check(self._lib.tl_two_opt_last_counters(self._h, out))
        return [int(v) for v in out]


_default = threading.local()


def default_context():
    ctx = getattr(_default, "ctx", None)
    if ctx is None:
        ctx = Context(0)
        _default.ctx = ctx
    return ctx


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class KDPoint:
    """src/tsp/kdtree.rs:248-252  {id: usize, coords: [f32; 2]}"""
    __slots__ = ("id", "coords")

    def __init__(self, id, coords):
        self.id = int(id)
        self.coords = (np.float32(coords[0]), np.float32(coords[1]))


class HeuristicOptions:
    """src/tsp/mod.rs:596-613"""

    def __init__(self, epochs=10_000, platoo_epochs=500, n_nearest=3, verbose=False):
        self.epochs, self.platoo_epochs, self.n_nearest, self.verbose = epochs, platoo_epochs, n_nearest, verbose

    def validate(self):  # mod.rs:677-682
        if self.n_nearest == 0:
            raise ValueError("n_nearest must be >= 1")


class LKOptions:
    """src/tsp/mod.rs:1249-1267"""

    def __init__(self, heuristic=None, max_depth=5):
        self.heuristic = heuristic or HeuristicOptions(epochs=100, platoo_epochs=10, n_nearest=5)
        self.max_depth = max_depth

    def validate(self):  # mod.rs:1270-1276
        self.heuristic.validate()
        if self.max_depth == 0:
            raise ValueError("max_depth must be >= 1")


from . import distance_matrix  # noqa: E402


class TspProblem:
    """src/tsp/mod.rs:1731-1741  {cities: Vec<KDPoint>, distances: DistanceMatrix}.

    Stored as arrays: ids (int64, arbitrary, 1-based in TSPLIB files) and xy (float32 n x 2) in city
    order; position p <-> id ids[p]."""

    def __init__(self, ids, xy, distances=None):
        self.ids = np.ascontiguousarray(ids, dtype=np.int64)
        self.xy = np.ascontiguousarray(xy, dtype=np.float32).reshape(-1, 2)
        if len(self.ids) != len(self.xy):
            raise ValueError("ids / xy length mismatch")
        self.distances = distances
        self._id2pos = None

    @classmethod
    def new(cls, cities, distances=None):
        return cls([c.id for c in cities], [[c.coords[0], c.coords[1]] for c in cities], distances)

    @property
    def cities(self):
        return [KDPoint(i, c) for i, c in zip(self.ids, self.xy)]

    def __len__(self):
        return len(self.ids)

    def id2pos(self):
        if self._id2pos is None:
            self._id2pos = {int(v): p for p, v in enumerate(self.ids)}
            if len(self._id2pos) != len(self.ids):
                raise ValueError("duplicate city ids")
        return self._id2pos

    def positions_of(self, route_ids):
        """City ids -> positions for a tour that crosses the C ABI as n u32 values: the length is checked HERE, before any
        pointer is taken (a shorter tour would be read past its end by the library)."""
        m = self.id2pos()
        try:
            pos = np.asarray([m[int(v)] for v in route_ids], dtype=np.uint32)
        except KeyError as e:  # the reference: .expect("two_opt: invalid city pair") -> panic
            raise ReferencePanics(_capi.TL_ERR_REF_PANICS, f"invalid city id {e.args[0]} in init_tour") from None
        if len(pos) != len(self.ids):
            raise TeelineGpuError(_capi.TL_ERR_BADARG, f"tour has {len(pos)} cities, the problem {len(self.ids)}")
        return pos

    def explicit_packed(self):
        """Packed matrix to hand to the kernels, or None when distances are plain EUC_2D (the kernels
        then compute the identical f32 values on the fly)."""
        d = self.distances
        if d is None or d.kind == "euc2d":
            return None
        return d.items


class Solution:
    """src/tsp/mod.rs:1753-1814  {total: f32, route: Vec<usize> (ids), cities}"""

    def __init__(self, total, route_ids, problem, stats=None):
        self.total = np.float32(total)
        self._route = [int(v) for v in route_ids]
        self.problem = problem
        self.stats = stats or {}

    def route(self):
        return self._route

    def __len__(self):
        return len(self._route)


def validate_tour(tour_ids, problem):
    """src/tsp/mod.rs:1620-1634: every city id exactly once."""
    ids = sorted(int(v) for v in tour_ids)
    return ids == sorted(int(v) for v in problem.ids)


from . import lin_kernighan, multistart, nearest_neighbor, opt_tour, or_opt, pipeline, synth, three_opt, tsplib, two_opt  # noqa: E402,F401
