"""ctypes binding to the CPU oracle (oracle/libtl_oracle.so).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")


class Stats(C.Structure):
    _fields_ = [("sweeps", C.c_uint64), ("candidates", C.c_uint64), ("moves", C.c_uint64),
                ("reversed", C.c_uint64)]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


def _build():
    so = os.path.join(ORACLE_DIR, "libtl_oracle.so")
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("tl_oracle.c", "tl_oracle_lk.c", "tl_oracle_kdtree.c", "tl_oracle.h")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
    return so


_lib = None
_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(_build())
        L.tlo_dist.restype = C.c_float
        L.tlo_dist.argtypes = [C.c_float] * 4
        L.tlo_dm_lookup.restype = C.c_float
        L.tlo_tour_length.restype = C.c_float
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _xy(xy):
    return None if xy is None else np.ascontiguousarray(xy, dtype=np.float32)


def _perm(p, n):
    return None if p is None else np.ascontiguousarray(p, dtype=np.uint32)


def dist(x1, y1, x2, y2):
    return float(lib().tlo_dist(x1, y1, x2, y2))


def dm_build_packed(xy, geo=False):
    xy = _xy(xy)
    n = xy.shape[0]
    out = np.empty(n * (n - 1) // 2, dtype=np.float32)
    fn = lib().tlo_dm_build_packed_geo if geo else lib().tlo_dm_build_packed
    rc = fn(_p(xy), C.c_uint32(n), _p(out))
    if rc:
        raise ValueError(f"oracle rc={rc}")
    return out


def dm_lookup(packed, p, q):
    return float(lib().tlo_dm_lookup(_p(packed), C.c_uint32(p), C.c_uint32(q)))


def dm_expand_full(packed, n):
    out = np.empty((n, n), dtype=np.float32)
    lib().tlo_dm_expand_full(_p(packed), C.c_uint32(n), _p(out))
    return out


def tour_length(xy, packed, perm):
    xy = _xy(xy)
    perm = np.ascontiguousarray(perm, dtype=np.uint32)
    return np.float32(lib().tlo_tour_length(_p(xy), _p(packed), C.c_uint32(len(perm)), _p(perm)))


def two_opt(xy, packed, n, init=None, flavor=0, max_candidates=0, best=False, max_moves=0):
    xy = _xy(xy)
    init = _perm(init, n)
    out = np.empty(n, dtype=np.uint32)
    cost = C.c_float()
    st = Stats()
    if best:
        rc = lib().tlo_two_opt_best(_p(xy), _p(packed), C.c_uint32(n), _p(init), _p(out),
                                    C.byref(cost), C.byref(st), C.c_uint64(max_moves))
    else:
        rc = lib().tlo_two_opt(_p(xy), _p(packed), C.c_uint32(n), _p(init), _p(out), C.byref(cost),
                               C.byref(st), C.c_int(flavor), C.c_uint64(max_candidates))
    return rc, out, np.float32(cost.value), st.as_dict()


def two_opt_trace(xy, packed, n, init=None, cap=1 << 20):
    """two_opt + the applied moves: returns (rc, route, cost, stats, ij [m, 2] uint32, dist [m] float32, sweep [m] uint32) — dist[m] is
    the new_distance the reference sends with move m's PathUpdate (two_opt.rs:53-56), sweep[m] the 1-based pass it happened in."""
    xy = _xy(xy)
    init = _perm(init, n)
    out = np.empty(n, dtype=np.uint32)
    cost = C.c_float()
    st = Stats()
    ij = np.empty((cap, 2), dtype=np.uint32)
    dist = np.empty(cap, dtype=np.float32)
    sw = np.empty(cap, dtype=np.uint32)
    ln = C.c_uint64()
    rc = lib().tlo_two_opt_trace(_p(xy), _p(packed), C.c_uint32(n), _p(init), _p(out), C.byref(cost), C.byref(st), _p(ij), _p(dist), _p(sw),
                                 C.c_uint64(cap), C.byref(ln))
    m = min(int(ln.value), cap)
    return rc, out, np.float32(cost.value), st.as_dict(), ij[:m].copy(), dist[:m].copy(), sw[:m].copy()


def trace_words(ij, sweep):
    """The word list tl_two_opt_trace returns for these moves: (i << 16) | j, and 0xFFFFFFFF where a new sweep begins."""
    words, cur = [], 1
    for (i, j), s in zip(ij.tolist(), sweep.tolist()):
        while cur < s:
            words.append(0xFFFFFFFF)
            cur += 1
        words.append((i << 16) | j)
    return words, cur


def swap_2opt(path, a, b):
    path = np.ascontiguousarray(path, dtype=np.uint32).copy()
    lib().tlo_swap_2opt(_p(path), C.c_uint32(a), C.c_uint32(b))
    return path


def reconnection_costs(e12):
    e = np.ascontiguousarray(e12, dtype=np.float32)
    out = np.empty(7, dtype=np.float32)
    lib().tlo_reconnection_costs(_p(e), _p(out))
    return out


def apply_3opt(path, i, j, k, case):
    path = np.ascontiguousarray(path, dtype=np.uint32).copy()
    rc = lib().tlo_apply_3opt(_p(path), C.c_uint32(len(path)), C.c_uint32(i), C.c_uint32(j),
                              C.c_uint32(k), C.c_int(case))
    return rc, path


def three_opt_find_best_move(xy, packed, path):
    xy = _xy(xy)
    path = np.ascontiguousarray(path, dtype=np.uint32)
    i, j, k = C.c_uint32(), C.c_uint32(), C.c_uint32()
    case, sav = C.c_int(), C.c_float()
    found = lib().tlo_three_opt_find_best_move(_p(xy), _p(packed), C.c_uint32(len(path)), _p(path),
                                               C.byref(i), C.byref(j), C.byref(k), C.byref(case),
                                               C.byref(sav))
    if not found:
        return None
    return (i.value, j.value, k.value, case.value, np.float32(sav.value))


def three_opt(xy, packed, n, init=None, max_moves=0):
    xy = _xy(xy)
    init = _perm(init, n)
    out = np.empty(n, dtype=np.uint32)
    cost = C.c_float()
    st = Stats()
    rc = lib().tlo_three_opt(_p(xy), _p(packed), C.c_uint32(n), _p(init), _p(out), C.byref(cost),
                             C.byref(st), C.c_uint64(max_moves))
    return rc, out, np.float32(cost.value), st.as_dict()


def apply_relocation(tour, i, seg_len, j, reversed):
    tour = np.ascontiguousarray(tour, dtype=np.uint32).copy()
    rc = lib().tlo_apply_relocation(_p(tour), C.c_uint32(len(tour)), C.c_uint32(i), C.c_uint32(seg_len), C.c_uint32(j),
                                    C.c_int(int(reversed)))
    return rc, tour


def or_opt_find_best_move(xy, packed, path):
    xy = _xy(xy)
    path = np.ascontiguousarray(path, dtype=np.uint32)
    d = C.c_float()
    i, j, seg = C.c_uint32(), C.c_uint32(), C.c_uint32()
    rev = C.c_int()
    found = lib().tlo_or_opt_find_best_move(_p(xy), _p(packed), C.c_uint32(len(path)), _p(path), C.byref(d), C.byref(i),
                                            C.byref(j), C.byref(seg), C.byref(rev))
    if not found:
        return None
    return (np.float32(d.value), i.value, j.value, seg.value, bool(rev.value))


def or_opt(xy, packed, n, init=None, max_moves=0):
    xy = _xy(xy)
    init = _perm(init, n)
    out = np.empty(n, dtype=np.uint32)
    cost = C.c_float()
    st = Stats()
    rc = lib().tlo_or_opt(_p(xy), _p(packed), C.c_uint32(n), _p(init), _p(out), C.byref(cost), C.byref(st),
                          C.c_uint64(max_moves))
    return rc, out, np.float32(cost.value), st.as_dict()


def nearest_neighbor(xy, packed, n, n_nearest=3):
    xy = _xy(xy)
    out = np.empty(n, dtype=np.uint32)
    cost = C.c_float()
    rc = lib().tlo_nearest_neighbor(_p(xy), _p(packed), C.c_uint32(n), C.c_uint32(n_nearest), _p(out),
                                    C.byref(cost))
    return rc, out, np.float32(cost.value)


def validate_tour(perm):
    perm = np.ascontiguousarray(perm, dtype=np.uint32)
    return bool(lib().tlo_validate_tour(_p(perm), C.c_uint32(len(perm))))


def synth_xy(n, seed=0):
    xy = np.empty((n, 2), dtype=np.float32)
    lib().tlo_synth_xy(C.c_uint32(n), C.c_uint64(seed), _p(xy))
    return xy


def restart_perm(n, seed, r):
    perm = np.empty(n, dtype=np.uint32)
    lib().tlo_restart_perm(C.c_uint32(n), C.c_uint64(seed), C.c_uint64(r), _p(perm))
    return perm


def build_candidates(xy, k):
    xy = _xy(xy)
    n = xy.shape[0]
    kk = min(k, n - 1)
    out = np.empty((n, max(kk, 1)), dtype=np.uint32)
    lib().tlo_build_candidates(_p(xy), C.c_uint32(n), C.c_uint32(k), _p(out))
    return out[:, :kk]


def lk_pass(xy, tour, cand, max_depth):
    xy = _xy(xy)
    tour = np.ascontiguousarray(tour, dtype=np.uint32).copy()
    cand = np.ascontiguousarray(cand, dtype=np.uint32)
    st = Stats()
    imp = lib().tlo_lk_pass(_p(xy), C.c_uint32(len(tour)), _p(tour), _p(cand), C.c_uint32(cand.shape[1]),
                            C.c_uint32(max_depth), C.byref(st))
    return bool(imp), tour, st.as_dict()


def double_bridge(tour, r1, r2, r3):
    tour = np.ascontiguousarray(tour, dtype=np.uint32)
    out = np.empty_like(tour)
    lib().tlo_double_bridge(_p(tour), C.c_uint32(len(tour)), C.c_uint32(r1), C.c_uint32(r2),
                            C.c_uint32(r3), _p(out))
    return out


def lin_kernighan(xy, init=None, epochs=100, platoo_epochs=10, n_nearest=5, max_depth=5, seed=1, cand=None, packed=None):
    """cand: precomputed candidate lists (n x min(n_nearest, n-1)), e.g. build_candidates_kdtree's; None = brute force."""
    xy = _xy(xy)
    n = xy.shape[0]
    init = _perm(init, n)
    out = np.empty(n, dtype=np.uint32)
    cost = C.c_float()
    st = Stats()
    if cand is not None:
        cand = np.ascontiguousarray(cand, dtype=np.uint32)
        assert cand.shape == (n, min(n_nearest, n - 1))
    rc = lib().tlo_lin_kernighan_cand(_p(xy), _p(packed), C.c_uint32(n), _p(init), C.c_uint32(epochs),
                                      C.c_uint32(platoo_epochs), C.c_uint32(n_nearest), C.c_uint32(max_depth),
                                      C.c_uint64(seed), _p(cand), _p(out), C.byref(cost), C.byref(st))
    return rc, out, np.float32(cost.value), st.as_dict()


def lin_kernighan_trace(xy, init=None, epochs=100, platoo_epochs=10, n_nearest=5, max_depth=5, seed=1, packed=None, cap=256):
    """lin_kernighan (kd-tree candidate lists) + the reference's progress messages (lin_kernighan.rs:71,90): list of (tour positions, best_dist)."""
    xy = _xy(xy)
    n = xy.shape[0]
    init = _perm(init, n)
    out = np.empty(n, dtype=np.uint32)
    cost = C.c_float()
    st = Stats()
    snaps = np.zeros((cap, n), dtype=np.uint32)
    dists = np.zeros(cap, dtype=np.float32)
    cnt = C.c_uint32()
    rc = lib().tlo_lin_kernighan_trace(_p(xy), _p(packed), C.c_uint32(n), _p(init), C.c_uint32(epochs), C.c_uint32(platoo_epochs),
                                       C.c_uint32(n_nearest), C.c_uint32(max_depth), C.c_uint64(seed), _p(out), C.byref(cost), C.byref(st),
                                       _p(snaps), _p(dists), C.c_uint32(cap), C.byref(cnt))
    assert cnt.value <= cap
    return rc, out, np.float32(cost.value), st.as_dict(), [(snaps[m].copy(), np.float32(dists[m])) for m in range(cnt.value)]


def build_candidates_kdtree(xy, k):
    """(lists, tie_free): lin_kernighan.rs:12-27 through the restated kd-tree."""
    xy = _xy(xy)
    n = xy.shape[0]
    kk = min(k, n - 1)
    out = np.empty((n, max(kk, 1)), dtype=np.uint32)
    tf = C.c_int(0)
    rc = lib().tlo_build_candidates_kdtree(_p(xy), C.c_uint32(n), C.c_uint32(k), _p(out), C.byref(tf))
    assert rc == 0
    return out[:, :kk], bool(tf.value)


def kdtree_nearest(xy, q, k, qid=None, ids=None):
    """KDTree::nearest: [(pos, distance), ...] in buffer order.  qid None = u64::MAX-like sentinel (no self-exclusion)."""
    xy = _xy(xy)
    n = xy.shape[0]
    op = np.empty(max(k, 1), dtype=np.uint32)
    od = np.empty(max(k, 1), dtype=np.float32)
    idp = None if ids is None else np.ascontiguousarray(ids, dtype=np.uint64)
    cnt = lib().tlo_kdtree_nearest(_p(xy), _p(idp), C.c_uint32(n), C.c_float(q[0]), C.c_float(q[1]),
                                   C.c_uint64(0xFFFFFFFFFFFFFFFF if qid is None else qid), C.c_uint32(k), _p(op), _p(od))
    assert cnt >= 0
    return [(int(op[i]), np.float32(od[i])) for i in range(cnt)]


def kdtree_walk(xy):
    xy = _xy(xy)
    n = xy.shape[0]
    out = np.empty(max(n, 1), dtype=np.uint32)
    tf = C.c_int(0)
    lib().tlo_kdtree_walk(_p(xy), C.c_uint32(n), _p(out), C.byref(tf))
    return out[:n], bool(tf.value)


def flat_to_next_prev(tour):
    tour = np.ascontiguousarray(tour, dtype=np.uint32)
    n = len(tour)
    nxt = np.zeros(int(tour.max()) + 1, dtype=np.uint32)
    prv = np.zeros(int(tour.max()) + 1, dtype=np.uint32)
    lib().tlo_flat_to_next_prev(_p(tour), C.c_uint32(n), _p(nxt), _p(prv))
    return nxt, prv


def find_lk_move(xy, tour, cand, max_depth):
    """find_lk_move on a flat tour: the chain (list) or None."""
    xy = _xy(xy)
    tour = np.ascontiguousarray(tour, dtype=np.uint32)
    cand = np.ascontiguousarray(cand, dtype=np.uint32)
    chain = np.zeros(2 * max_depth + 4, dtype=np.uint32)
    clen = lib().tlo_find_lk_move(_p(xy), C.c_uint32(len(tour)), _p(tour), _p(cand), C.c_uint32(cand.shape[1]),
                                  C.c_uint32(max_depth), _p(chain))
    return chain[:clen].tolist() if clen else None


def apply_lk_chain(tour, chain):
    tour = np.ascontiguousarray(tour, dtype=np.uint32).copy()
    chain = np.ascontiguousarray(chain, dtype=np.uint32)
    rc = lib().tlo_apply_lk_chain(_p(tour), C.c_uint32(len(tour)), _p(chain), C.c_uint32(len(chain)))
    assert rc == 0
    return tour
