// lk_deep.hip — the Lin-Kernighan kernels of lk.hip once more, with room for chains of up to 16 exchanges (max_depth 7..16;
// the reference's LKOptions::max_depth is an unbounded usize, src/tsp/mod.rs:1252, recursion src/tsp/lin_kernighan.rs:265-340).
// Same source, same algorithm, same results for any max_depth both builds take; the chain arrays (34 cities instead of 14)
// no longer fit the registers of an 8-waves-per-SIMD kernel, so this build is slower per node and is only used beyond depth 6
// (tl_api.hip lk_run).
#define TL_LK_MAX_DEPTH 16
#define TL_LK_NS tl_lk_deep
#include "lk.hip"
