"""teeline_amd — MI355X-native 2-opt / 3-opt / Lin–Kernighan local-search engine.

Host-side mirror of the solver entry points of the `teeline` Rust crate (timgluz/teeline):
    two_opt.solve / three_opt.solve / lin_kernighan.solve (problem, opts, progress_tx, init_tour)
    reference: src/tsp/two_opt.rs:7-12, src/tsp/three_opt.rs:16-21, src/tsp/lin_kernighan.rs:35-40
implemented as thin wrappers over the C ABI of libteeline_gpu.so (include/teeline_gpu.h), whose hot
paths are hand-written HIP kernels for gfx950.  There is no CPU fallback: without the built library
and a gfx950 device every solver call raises.
"""
from . import _capi
from ._capi import (TL_FLAG_2OPT_FORCE_HBM, TL_FLAG_2OPT_FX, TL_FLAG_2OPT_NL_ALWAYS, TL_FLAG_2OPT_NO_NL, TL_FLAG_2OPT_NT256, TL_FLAG_2OPT_NT512, TL_FLAG_COUNT_WORK, TL_FLAG_KNN_BRUTE, TL_FLAG_KNN_1LANE, TL_FLAG_KNN_4LANES, TL_FLAG_LK_NO_SPLIT,
                    TL_FLAG_LK_CHIP_WIDE, TL_FLAG_LK_CLASSIC_VIEW, TL_FLAG_LK_ILS_LDS, TL_FLAG_LK_NO_GRAPH, TL_FLAG_LK_NO_SPECULATION, TL_FLAG_LK_NO_SUBCHAINS, TL_FLAG_LK_ONE_WORKGROUP, TL_FLAG_LK_SCAN_PERSIST, TL_FLAG_LK_SEPARATE_PICK, TL_FLAG_LK_SEPARATE_STEP, TL_FLAG_LK_SMALL, TL_FLAG_LK_SPLIT2, TL_FLAG_MULTISTART_RCCL, TL_FLAG_NONE, TL_FLAG_NO_PRUNE,
                    TL_MODE_BEST_SWEEP, TL_MODE_REF_ORDER, ReferencePanics, TeelineGpuError)
from .host import (Context, HeuristicOptions, KDPoint, LKOptions, Solution, TspProblem, default_context,
                   distance_matrix, lin_kernighan, multistart, nearest_neighbor, opt_tour, or_opt, pipeline, synth, three_opt, tsplib,
                   two_opt,
                   validate_tour)

__all__ = [
    "Context", "HeuristicOptions", "KDPoint", "LKOptions", "Solution", "TspProblem", "default_context",
    "distance_matrix", "lin_kernighan", "three_opt", "tsplib", "two_opt", "validate_tour",
    "TL_MODE_REF_ORDER", "TL_MODE_BEST_SWEEP", "TL_FLAG_NONE", "TL_FLAG_NO_PRUNE",
    "TeelineGpuError", "ReferencePanics",
]
