"""ctypes loader for libteeline_gpu.so — the C ABI declared in include/teeline_gpu.h.

The library is the product; this module only binds it.  There is no CPU fallback anywhere in this
package: if the shared object is missing or no gfx950 device is usable, calls raise.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# TEELINE_GPU_LIB: developer override to load a differently-built variant (tuning experiments)
LIB_PATH = os.environ.get("TEELINE_GPU_LIB") or os.path.join(_HERE, "libteeline_gpu.so")
TL_DEV_STATS_STRIDE = 16

TL_OK = 0
TL_ERR_BADARG, TL_ERR_REF_PANICS, TL_ERR_NO_DEVICE, TL_ERR_HIP = -1, -2, -3, -4
TL_ERR_NOMEM, TL_ERR_UNSUPPORTED, TL_ERR_NO_CONVERGE, TL_ERR_BUSY = -5, -6, -7, -8
TL_MODE_REF_ORDER, TL_MODE_BEST_SWEEP = 0, 1
TL_FLAG_NONE, TL_FLAG_NO_PRUNE = 0, 1
# alternative kernel forms (identical results; cross-checks of each other)
TL_FLAG_2OPT_FORCE_HBM, TL_FLAG_LK_ONE_WORKGROUP, TL_FLAG_LK_NO_SPLIT, TL_FLAG_LK_SPLIT2 = 1 << 1, 1 << 2, 1 << 3, 1 << 4
TL_FLAG_LK_NO_SUBCHAINS, TL_FLAG_KNN_4LANES, TL_FLAG_KNN_1LANE = 1 << 5, 1 << 6, 1 << 7
TL_FLAG_KNN_BRUTE = 1 << 10  # candidate lists by brute-force scan instead of the kd-tree walk
TL_FLAG_LK_SEPARATE_PICK = 1 << 11  # tl_lk: the pick / validate step as its own kernel (cross-check of the fused form)
TL_FLAG_LK_NO_GRAPH = 1 << 12  # tl_lk: no hipGraph replay of the round loop
TL_FLAG_LK_SEPARATE_STEP = 1 << 13  # tl_lk: k_lk_control + k_lk_rebuild instead of the chip-wide step kernel
TL_FLAG_2OPT_NT512, TL_FLAG_2OPT_NT256 = 1 << 14, 1 << 15  # LDS 2-opt: force the 8- / 4-wave form of a descent
TL_FLAG_2OPT_FX = 1 << 16  # LDS 2-opt: grid-coordinate form of the tour (two tours of n = 10^4 per CU) wherever it is exact
TL_FLAG_2OPT_NO_NL, TL_FLAG_2OPT_NL_ALWAYS = 1 << 18, 1 << 19  # LDS 2-opt: neighbour-list rows of the late sweeps off / wherever they fit
TL_FLAG_LK_SCAN_PERSIST = 1 << 17  # tl_lk (tuning build): the fused scan as a persistent grid striding over the window's pairs
TL_FLAG_LK_CHIP_WIDE = 1 << 20  # tl_lk: chip-wide scans at every n
TL_FLAG_LK_ILS_LDS = 1 << 21    # tl_lk: the single-workgroup LDS form (k_lk_ils) at every n it fits
TL_FLAG_MULTISTART_RCCL = 1 << 24  # multi-start over several devices of one process: RCCL min-all-reduce + broadcast inside the library
TL_FLAG_LK_CLASSIC_VIEW = 1 << 23  # tl_lk chip-wide: cand -> xy -> next -> xy look-ups instead of the packed records
TL_FLAG_LK_NO_SPECULATION = 1 << 22  # tl_lk, LDS form: epochs one after the other (default: a batch of consecutive epochs at once)
TL_FLAG_LK_SMALL = 1 << 9  # tl_lk: the LDS-resident single-workgroup form wherever it fits
TL_FLAG_COUNT_WORK = 1 << 8  # the LDS 2-opt kernel also counts the work of its cascade (stats words 5..8); ~8 % slower
TL_DM_PACKED_LOWER, TL_DM_FULL = 0, 1
TL_DIST_EUC2D, TL_DIST_GEO = 0, 1

# every symbol include/teeline_gpu.h declares (tests/test_abi.py checks header <-> library <-> this list)
SYMBOLS = [
    "tl_abi_version", "tl_version", "tl_create", "tl_destroy", "tl_last_error", "tl_device_info",
    "tl_two_opt_lds_max_n", "tl_dm_build", "tl_tour_length", "tl_two_opt", "tl_three_opt",
    "tl_three_opt_find_best_move", "tl_lk", "tl_two_opt_multistart", "tl_pack_cost_key",
    "tl_two_opt_batch_dev", "tl_last_kernel_ms", "tl_dm_build_dev", "tl_build_candidates", "tl_nearest_neighbor",
    "tl_or_opt", "tl_or_opt_find_best_move", "tl_selftest_sqrt", "tl_two_opt_population", "tl_dm_is_euc2d",
    "tl_two_opt_multistart_devices", "tl_two_opt_trace", "tl_three_opt_trace", "tl_lk_trace", "tl_or_opt_trace",
    "tl_lk_live", "tl_two_opt_neighbour_lists", "tl_two_opt_plan", "tl_multistart_shard", "tl_two_opt_last_counters",
]


class TlStats(C.Structure):
    _fields_ = [("sweeps", C.c_uint64), ("candidates", C.c_uint64), ("moves", C.c_uint64),
                ("reversed", C.c_uint64), ("kernel_ms", C.c_double), ("total_ms", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class TlLkOpts(C.Structure):
    _fields_ = [("epochs", C.c_uint32), ("platoo_epochs", C.c_uint32), ("n_nearest", C.c_uint32),
                ("max_depth", C.c_uint32)]


# tl_lk_progress_fn: void (*)(void *user, const uint32_t *best_pos, uint32_t n, float best_dist)
LK_PROGRESS_FN = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_uint32), C.c_uint32, C.c_float)


class TeelineGpuError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libteeline_gpu error {code}: {msg}")
        self.code = code


class ReferencePanics(TeelineGpuError):
    """Input on which the reference solver itself panics (e.g. two_opt with n < 3)."""


_lib = None


def load():
    """Load libteeline_gpu.so.  Raises if it has not been built — never falls back to a CPU path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -m teeline_amd.build` (hipcc, gfx950). "
            "teeline_amd has no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    vp, u32, u64, i32, f32p = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int, C.POINTER(C.c_float)
    L.tl_abi_version.restype = i32
    L.tl_version.restype = C.c_char_p
    L.tl_create.argtypes = [i32, u32, C.POINTER(vp)]
    L.tl_destroy.argtypes = [vp]
    L.tl_destroy.restype = None
    L.tl_last_error.argtypes = [vp]
    L.tl_last_error.restype = C.c_char_p
    L.tl_device_info.argtypes = [vp, C.POINTER(i32), C.POINTER(i32), C.c_char_p, C.c_size_t]
    L.tl_two_opt_lds_max_n.argtypes = [vp]
    L.tl_two_opt_lds_max_n.restype = u32
    L.tl_dm_build.argtypes = [vp, vp, u32, i32, i32, vp, C.POINTER(C.c_double)]
    L.tl_tour_length.argtypes = [vp, vp, vp, u32, vp, f32p]
    L.tl_dm_is_euc2d.argtypes = [vp, vp, vp, u32, C.POINTER(i32)]
    L.tl_two_opt.argtypes = [vp, vp, u32, vp, vp, i32, vp, f32p, C.POINTER(TlStats)]
    L.tl_three_opt.argtypes = [vp, vp, u32, vp, vp, vp, f32p, C.POINTER(TlStats)]
    L.tl_three_opt_find_best_move.argtypes = [vp, vp, u32, vp, vp, C.POINTER(i32), C.POINTER(u32),
                                              C.POINTER(u32), C.POINTER(u32), C.POINTER(i32), f32p]
    L.tl_lk.argtypes = [vp, vp, u32, vp, vp, C.POINTER(TlLkOpts), u64, vp, f32p, C.POINTER(TlStats)]
    L.tl_two_opt_multistart.argtypes = [vp, vp, u32, u64, u32, u32, i32, vp, f32p, C.POINTER(u32), vp,
                                        C.POINTER(TlStats)]
    L.tl_two_opt_multistart_devices.argtypes = [C.POINTER(vp), i32, vp, u32, u64, u32, u32, i32, vp, f32p, C.POINTER(u32), vp,
                                                C.POINTER(TlStats)]
    L.tl_two_opt_population.argtypes = [vp, vp, u32, vp, vp, u32, vp, vp, C.POINTER(TlStats)]
    L.tl_two_opt_trace.argtypes = [vp, vp, u32, vp, vp, vp, f32p, C.POINTER(TlStats), vp, u32, C.POINTER(u32)]
    L.tl_three_opt_trace.argtypes = [vp, vp, u32, vp, vp, vp, f32p, C.POINTER(TlStats), vp, u32, C.POINTER(u32)]
    L.tl_or_opt_trace.argtypes = [vp, vp, u32, vp, vp, vp, f32p, C.POINTER(TlStats), vp, u32, C.POINTER(u32)]
    L.tl_lk_trace.argtypes = [vp, vp, u32, vp, vp, C.POINTER(TlLkOpts), u64, vp, f32p, C.POINTER(TlStats), vp, vp, u32, C.POINTER(u32)]
    L.tl_lk_live.argtypes = [vp, vp, u32, vp, vp, C.POINTER(TlLkOpts), u64, vp, f32p, C.POINTER(TlStats), LK_PROGRESS_FN, vp]
    L.tl_pack_cost_key.argtypes = [C.c_float, u32]
    L.tl_pack_cost_key.restype = u64
    L.tl_two_opt_batch_dev.argtypes = [vp, vp, u32, vp, u64, u32, u32, i32, vp, vp, vp, vp]
    L.tl_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_double)]
    L.tl_two_opt_last_counters.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.tl_dm_build_dev.argtypes = [vp, vp, u32, i32, i32, vp, vp]
    L.tl_build_candidates.argtypes = [vp, vp, u32, u32, vp]
    L.tl_nearest_neighbor.argtypes = [vp, vp, vp, u32, u32, vp, f32p]
    L.tl_selftest_sqrt.argtypes = [vp, u32, u64, C.POINTER(u64), C.POINTER(u32)]
    L.tl_two_opt_plan.argtypes = [u32, u32, i32, i32, u32, C.POINTER(i32), C.POINTER(i32)]
    L.tl_multistart_shard.argtypes = [u32, u32, i32, i32, C.POINTER(u32), C.POINTER(u32)]
    L.tl_two_opt_neighbour_lists.argtypes = [vp, vp, u32, i32, vp, vp, vp, vp, C.POINTER(u32), C.POINTER(u32), C.POINTER(u32)]
    L.tl_or_opt.argtypes = [vp, vp, u32, vp, vp, vp, f32p, C.POINTER(TlStats)]
    L.tl_or_opt_find_best_move.argtypes = [vp, vp, u32, vp, vp, C.POINTER(i32), f32p, C.POINTER(u32), C.POINTER(u32),
                                           C.POINTER(u32), C.POINTER(i32)]
    _lib = L
    return L
