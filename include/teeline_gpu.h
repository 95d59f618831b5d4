/*
 * teeline_gpu.h — C ABI of libteeline_gpu.so: the MI355X (gfx950) local-search engine that slots in
 * behind the solver entry points of the `teeline` Rust crate (timgluz/teeline).
 *
 * The reference has NO FFI and no Solver trait; its plug-in boundary is three free
 * functions of identical shape selected by a `match`:
 *     two_opt::solve / three_opt::solve / lin_kernighan::solve
 *         (&TspProblem, &Options, Option<&Sender<ProgressMessage>>, Option<&[usize]>) -> Solution
 *     reference: src/tsp/two_opt.rs:7-12, src/tsp/three_opt.rs:16-21, src/tsp/lin_kernighan.rs:35-40,
 *                dispatch src/tsp/mod.rs:1690-1694,1718,1719.
 * Each entry point below is what a Rust `extern "C"` block binds to replace the body
 * of one of those functions or of a helper they call.  Plain pointers and sizes only.
 *
 * Conventions
 *  - Tours are POSITIONS: indices 0..n-1 into the caller's city array (the reference maps
 *    id<->position with HashMaps at every distance lookup, distance_matrix.rs:197-212; the shim does
 *    that once at the boundary).  u32 is enough: n <= 2^32-1.
 *  - `xy` is n x 2 f32 row-major in city order (KDPoint.coords, kdtree.rs:248-252).
 *  - `dm_packed` (optional) is the reference's packed strict-lower-triangle matrix, n(n-1)/2 f32,
 *    idx(i>j) = i(i-1)/2 + j (distance_matrix.rs:122-153,186).  Pass it for EXPLICIT / GEO problems;
 *    NULL means EUC_2D computed on the fly from xy with the reference's exact f32 arithmetic
 *    (sqrtf(dx*dx+dy*dy), no FMA; kdtree.rs:291-295).
 *  - All host pointers are borrowed for the duration of the call only.  A tl_ctx is single-threaded;
 *    distinct contexts are independent (own HIP stream, event pair and workspace; the library's only
 *    shared mutable state is a mutex-guarded set) — the reference calls solvers from arbitrary threads
 *    (teeline-api tsp_service.rs:295,328, teeline-qt solver_engine.rs:412-432), so create one context per
 *    thread; any number of threads may be inside the library at once, each with its own context
 *    (tests/test_gpu_threads.py).  Two threads on ONE context do not race: the second to arrive gets
 *    TL_ERR_BUSY back at once from every entry point that takes the context.  (tl_destroy on a context
 *    another thread is using is undefined, as free() of a buffer in use is.)
 *  - Return value: TL_OK (0) or a negative tl_status; tl_last_error(ctx) has the message.
 *  - There is NO CPU fallback: every compute entry point fails with TL_ERR_NO_DEVICE / TL_ERR_HIP when
 *    no gfx950 device is usable.
 */
#ifndef TEELINE_GPU_H
#define TEELINE_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TL_ABI_VERSION 5 /* bumped when an existing entry point, struct or code changes meaning; additions (round 5: tl_multistart_shard,
                            tl_two_opt_last_counters, new tl_create flags, k <= 64) leave it — a caller built against 5 runs unchanged */

typedef struct tl_ctx tl_ctx;

typedef enum tl_status {
    TL_OK = 0,
    TL_ERR_BADARG = -1,      /* NULL pointer, n out of range, init tour not a permutation, ...        */
    TL_ERR_REF_PANICS = -2,  /* input on which the reference itself panics (e.g. 2-opt with n < 3,     */
                             /* two_opt.rs:17,29 usize underflow); the shim turns this into panic!()   */
    TL_ERR_NO_DEVICE = -3,   /* no HIP device / not gfx950                                             */
    TL_ERR_HIP = -4,         /* a HIP runtime call failed                                              */
    TL_ERR_NOMEM = -5,       /* device or host allocation failed                                       */
    TL_ERR_UNSUPPORTED = -6, /* size/mode combination this build cannot run                            */
    TL_ERR_NO_CONVERGE = -7, /* safety cap on sweeps hit (the reference would still be looping)        */
    TL_ERR_BUSY = -8         /* another host thread is inside an entry point with this context (ABI v5);  */
                             /* nothing was touched, tl_last_error(ctx) is NOT updated (it is that thread's) */
} tl_status;

/* 2-opt evaluation order */
typedef enum tl_mode {
    /* The reference's algorithm: first-improvement, in place, lexicographic (i,j), open path
     * (two_opt.rs:26-61).  Tours are bit-identical to the reference's.  Default. */
    TL_MODE_REF_ORDER = 0,
    /* This build's own throughput mode: per sweep evaluate every (i,j), apply the single best
     * improving move (lowest f32 delta, lowest linear index on ties), repeat.  NOT the reference's
     * algorithm; parity-checked against oracle/tlo_two_opt_best. */
    TL_MODE_BEST_SWEEP = 1
} tl_mode;

/* tl_create flags */
#define TL_FLAG_NONE 0u
/* Evaluate every candidate with two fresh correctly-rounded sqrt distances instead of the exact
 * squared-distance pre-test (see DESIGN.md "Exact pruning").  Results are identical; only speed
 * changes.  Used by bench.py to report the un-pruned rate next to the default one. */
#define TL_FLAG_NO_PRUNE 1u
/* Alternative forms of a kernel, kept as cross-checks of each other (the parity tests run every form against the oracle):
 * results are identical, only speed changes.  The library never reads the environment. */
#define TL_FLAG_2OPT_FORCE_HBM (1u << 1)   /* tl_two_opt: the HBM-resident REF_ORDER variant (n > LDS limit) at every n      */
#define TL_FLAG_LK_ONE_WORKGROUP (1u << 2) /* tl_lk: the whole ILS in one persistent workgroup instead of chip-wide scans  */
#define TL_FLAG_KNN_BRUTE (1u << 10)       /* candidate lists: brute-force scan (sixteen lanes per city) in (distance, position)
                                              order instead of the kd-tree walk — the same lists unless two candidates of a city
                                              are at the same f32 distance (then the reference's order is the tree's visiting order) */
#define TL_FLAG_2OPT_NT512 (1u << 14)      /* LDS 2-opt: every descent on 8 waves (default: only when two descents share a CU)   */
#define TL_FLAG_2OPT_NT256 (1u << 15)      /* LDS 2-opt: every descent on 4 waves (default: only when four descents share a CU)  */
#define TL_FLAG_2OPT_FX (1u << 16)         /* LDS 2-opt: the grid-coordinate form (5 B per point) wherever the instance lies on a decimal grid */
#define TL_FLAG_2OPT_NO_NL (1u << 18)      /* LDS 2-opt: never read neighbour lists — every pruned row walks its tiles (default: rows of the
                                              late sweeps (once a sweep has applied fewer than n / 40 moves) of an instance with n >= 400 read the lists, csrc/two_opt_nl.hip).
                                              Matrix form (dm_packed): never cut lists from the matrix rows — every row walks its matrix rows (default: from
                                              n = 500, a sweep that follows one with at most n^2 / 4000 moves decides a row from a's 16 nearest cities, b's
                                              reverse list and the cities with a long tour edge, csrc/two_opt_dm.hip; same tours either way)                  */
#define TL_FLAG_2OPT_NL_ALWAYS (1u << 19)  /* LDS 2-opt: neighbour-list rows at every n they fit and from the second sweep on; matrix form: list
                                              rows at every n >= 8 and in every sweep that has at most 1024 cities with a long tour edge              */
#define TL_FLAG_LK_CHIP_WIDE (1u << 20)    /* tl_lk: chip-wide scans at every n (default: an instance whose search state fits one CU's LDS runs its
                                              lk_pass in ONE workgroup with all state in LDS, k_lk_ils in csrc/lk.hip — up to n = 700, and wherever
                                              it fits (n ~ 3000) with epochs and platoo_epochs >= 64)                                            */
#define TL_FLAG_LK_ILS_LDS (1u << 21)      /* tl_lk: that single-workgroup LDS form at every n it fits                                          */
#define TL_FLAG_MULTISTART_RCCL (1u << 24)  /* tl_two_opt_multistart(_devices), set on ctxs[0]: the winner is found by an RCCL min-all-reduce of the
                                              devices' packed (cost, restart) keys and its tour reaches every device by an RCCL broadcast from its
                                              owner (one process, ncclCommInitAll; librccl is loaded on first use) — the host minimum is computed
                                              all the same and must agree.  Default: the host minimum alone.                                      */
#define TL_FLAG_LK_CLASSIC_VIEW (1u << 23)  /* tl_lk, chip-wide scans at n >= 1500: the chain search reads cand -> xy -> next -> xy (default: the
                                              packed view — candidates with their distances, successor records with the successor's point and the
                                              tour edge's length: two dependent look-ups and one square root per branch instead of four and three) */
#define TL_FLAG_LK_NO_SPECULATION (1u << 22) /* tl_lk, LDS form: the epochs one after the other in one workgroup (default: one workgroup per epoch,
                                              as many consecutive epochs at once as the chip holds, taken in order up to the first accepted one) */
/* TUNING BUILDS ONLY (libteeline_gpu_tune.so, -DTL_TUNE: `python -m teeline_amd.build --tune`).  Forms that were measured and
 * rejected and stay as cross-checks for development; the product library does not carry them and tl_create
 * returns TL_ERR_UNSUPPORTED if one of these bits is set. */
#define TL_FLAG_LK_NO_SPLIT (1u << 3)      /* tl_lk: one lane per (t1, orientation) pair, no sub-search split               */
#define TL_FLAG_LK_SPLIT2 (1u << 4)        /* tl_lk: two split levels (k(k+1) sub-searches per pair) instead of three       */
#define TL_FLAG_LK_NO_SUBCHAINS (1u << 5)  /* tl_lk: the pick step walks the winning chain again instead of reading it      */
#define TL_FLAG_KNN_4LANES (1u << 6)       /* candidate lists: brute force, four lanes per city                              */
#define TL_FLAG_KNN_1LANE (1u << 7)        /* candidate lists: brute force, one lane per city                                */
#define TL_FLAG_LK_SMALL (1u << 9)         /* tl_lk: the LDS-resident single-workgroup form at every n it fits               */
#define TL_FLAG_LK_SEPARATE_PICK (1u << 11) /* tl_lk: pick and validate the pairs' first chains in a kernel of their own (k_lk_scan_pick) */
#define TL_FLAG_LK_NO_GRAPH (1u << 12)      /* tl_lk: enqueue every round's kernels separately instead of replaying 64 rounds as one hipGraph */
#define TL_FLAG_LK_SEPARATE_STEP (1u << 13) /* tl_lk: state machine (one workgroup) and tour rebuild as two kernels at every n */
#define TL_FLAG_LK_SCAN_PERSIST (1u << 17)  /* tl_lk: the fused scan as a persistent grid striding over the window's pairs (round 4, VERDICT r03
                                               item 3: measured 60.6 us per round against 47.0 for one workgroup per pair at n = 13 509) */
#define TL_TUNE_ONLY_FLAGS                                                                                                        \
    (TL_FLAG_LK_NO_SPLIT | TL_FLAG_LK_SPLIT2 | TL_FLAG_LK_NO_SUBCHAINS | TL_FLAG_KNN_4LANES | TL_FLAG_KNN_1LANE | TL_FLAG_LK_SMALL | \
     TL_FLAG_LK_SEPARATE_PICK | TL_FLAG_LK_NO_GRAPH | TL_FLAG_LK_SEPARATE_STEP | TL_FLAG_LK_SCAN_PERSIST)
/* The LDS-resident 2-opt kernel also counts the work its exact decision cascade really does (d_out_stats words 5..8: tile
 * bounds, candidates into L1 / L2 / L3).  Same results; slower (the counters are live scalar registers), so bench.py uses it for one
 * untimed launch only.  Counted: the form of one descent per CU (16 waves, float2 points); the narrower and the
 * grid-coordinate forms run uncounted (words 5..8 stay 0). */
#define TL_FLAG_COUNT_WORK (1u << 8)

/* matrix layouts for tl_dm_build */
#define TL_DM_PACKED_LOWER 0 /* reference layout, n(n-1)/2 floats (distance_matrix.rs:122-153) */
#define TL_DM_FULL 1         /* n x n row-major, zero diagonal (coalesced row gathers)           */

/* distance function for tl_dm_build */
#define TL_DIST_EUC2D 0 /* kdtree.rs:291-295            */
#define TL_DIST_GEO 1   /* distance_matrix.rs:59-75     */

/* u64 words per descent written by tl_two_opt_batch_dev into d_out_stats */
#define TL_DEV_STATS_STRIDE 16

typedef struct tl_stats {
    uint64_t sweeps;     /* 2-opt: `while improved` iterations; 3-opt: passes; LK: lk_pass scans      */
    uint64_t candidates; /* candidates whose delta test was decided, counted as the reference's loop  */
                         /* visits them: 2-opt REF_ORDER/BEST_SWEEP = sweeps * (n-3)(n-2)/2            */
    uint64_t moves;      /* improving moves applied                                                    */
    uint64_t reversed;   /* tour elements moved by segment reversals                                   */
    double kernel_ms;    /* device time of the solver kernel(s), HIP events on the context's stream   */
    double total_ms;     /* wall time of the whole call incl. transfers                                */
} tl_stats;

typedef struct tl_lk_opts { /* LKOptions, src/tsp/mod.rs:1249-1267 */
    uint32_t epochs;        /* default 100 (10000 from the CLI, mod.rs:596-613,1321-1325) */
    uint32_t platoo_epochs; /* default 10                                                   */
    uint32_t n_nearest;     /* default 5                                                    */
    uint32_t max_depth;     /* default 5                                                    */
} tl_lk_opts;

/* ---- context ------------------------------------------------------------------------------- */
int tl_abi_version(void);
const char *tl_version(void);
/* device: HIP ordinal (one process per GPU: pass LOCAL_RANK). */
int tl_create(int device, uint32_t flags, tl_ctx **out);
void tl_destroy(tl_ctx *ctx);
const char *tl_last_error(const tl_ctx *ctx); /* ctx may be NULL: last tl_create failure */
/* CU count, LDS bytes per workgroup and arch name of the context's device. */
int tl_device_info(const tl_ctx *ctx, int *cus, int *lds_bytes, char *arch, size_t arch_len);
/* Largest n the LDS-resident REF_ORDER 2-opt kernel takes (tour + coordinates live in one CU's LDS).  tl_two_opt
 * handles larger n with the HBM-resident variant, and so (one descent after the other) do the multi-start and population entries;
 * the device-resident tl_two_opt_batch_dev and the trace entry are LDS-only. */
uint32_t tl_two_opt_lds_max_n(const tl_ctx *ctx);

/* ---- numerics self-test -------------------------------------------------------------------------- */
/* Every kernel's correctly rounded sqrt (the reference's f32::sqrt, kdtree.rs:294) is a short v_sqrt_f32 + FMA
 * fix-up; this compares it on the device with the compiler's full expansion for the `count` f32 bit patterns
 * starting at `first_bits`.  *mismatches must come back 0. */
int tl_selftest_sqrt(tl_ctx *ctx, uint32_t first_bits, uint64_t count, uint64_t *mismatches, uint32_t *first_bad_bits);

/* ---- diagnostics: the neighbour lists of the LDS 2-opt descent's late sweeps ---------------------------------- */
/* Once a sweep has applied fewer than n / 40 moves (random restarts at n = 10^4: from the sixth sweep on), a descent of an instance with n >= 400 (one descent per CU, or two / four of n <= 5568 / 2176; TL_FLAG_2OPT_NO_NL: never) decides a
 * row (a, b) from per-city lists instead of walking every tile (csrc/two_opt_nl.hip: improving => c is strictly closer to a than
 * b, or b strictly closer to e than c).  This builds the lists for xy as a call would and copies them out, for tests:
 * rec [n][64] u16 — per city: [0] high half of the bits of its KA-th smallest squared distance, [2] 1 if its reverse list is
 * incomplete, [4, 4+KA) its KA nearest cities, [20, 20+RB) the cities that count it among their KB nearest (0xFFFF: empty);
 * dkb2 [n] — bits of the KB-th smallest squared distance; knn_b [n][KB] — the KB nearest cities; rcnt [n] — reverse counts.
 * KA, KB, RB come back in *ka, *kb, *rb.  form 0: the kernel the library runs (a wave per city); 1: its cross-check (a workgroup per
 * city, radix select) — the same lists.  TL_ERR_UNSUPPORTED where the lists do not apply (n <= KB + 1). */
int tl_two_opt_neighbour_lists(tl_ctx *ctx, const float *xy, uint32_t n, int form, uint16_t *rec, uint32_t *dkb2, uint16_t *knn_b, uint32_t *rcnt,
                               uint32_t *ka, uint32_t *kb, uint32_t *rb);

/* Which form of the LDS 2-opt descent a batch of `count` descents of n cities runs on a device with `cus` compute units and
 * `lds_bytes` of LDS per workgroup (tl_device_info), float2 points (the grid-coordinate form is decided per instance, on the device):
 * *threads per descent (1024 / 512 / 256 = one / two / four descents per CU; 0: n exceeds the LDS-resident limit) and whether that
 * form has the late phase on neighbour lists.  A host-only query (no context, no device): the library's own selection rule. */
int tl_two_opt_plan(uint32_t n, uint32_t count, int cus, int lds_bytes, uint32_t flags, int *threads, int *late_phase);

/* ---- distance matrix: replaces DistanceMatrix::build (distance_matrix.rs:122-153) ----------- */
/* out_host may be NULL (matrix stays on the device for later calls on this context). */
int tl_dm_build(tl_ctx *ctx, const float *xy, uint32_t n, int dist, int layout, float *out_host,
                double *kernel_ms);

/* Does dm_packed hold exactly (bit for bit) the EUC_2D distances of xy?  The reference's DistanceMatrix does not keep its
 * DistanceType (distance_matrix.rs:86-93); a shim that only has `problem.distances.distances()` asks here once and then
 * passes dm_packed = NULL (coordinate kernels, same results) when *is_euc2d comes back 1.  Costs one upload of the matrix. */
int tl_dm_is_euc2d(tl_ctx *ctx, const float *xy, const float *dm_packed, uint32_t n, int *is_euc2d);

/* ---- tour cost: replaces DistanceMatrix::tour_length_by_pos (distance_matrix.rs:235-245) ---- */
/* total = d(last,first), then += d(w0,w1) in tour order (sequential f32, bit-exact). */
int tl_tour_length(tl_ctx *ctx, const float *xy, const float *dm_packed, uint32_t n,
                   const uint32_t *perm, float *out_cost);

/* ---- 2-opt: replaces two_opt::solve (two_opt.rs:7-67) ---------------------------------------- */
/* init_pos NULL = city order (two_opt.rs:18-20).  out_pos: n entries.  out_cost = tour_length(out_pos)
 * as Solution::from_parts computes it (mod.rs:1776-1789).  With dm_packed (EXPLICIT / GEO, "matrix in HBM") every
 * distance is read from the matrix, which is expanded on the device to a full row-major n x n copy (4 n^2 bytes of
 * HBM) and the tour plus its edge lengths live in LDS: n <= ~19 900 on 160 KB (TL_ERR_UNSUPPORTED beyond). */
int tl_two_opt(tl_ctx *ctx, const float *xy, uint32_t n, const float *dm_packed,
               const uint32_t *init_pos, int mode, uint32_t *out_pos, float *out_cost,
               tl_stats *stats);

/* ---- 3-opt: replaces three_opt::solve (three_opt.rs:16-51) ---------------------------------- */
int tl_three_opt(tl_ctx *ctx, const float *xy, uint32_t n, const float *dm_packed,
                 const uint32_t *init_pos, uint32_t *out_pos, float *out_cost, tl_stats *stats);
/* One find_best_move scan (three_opt.rs:58-131).  *found = 0/1; (i,j,k,case,savings) as the reference. */
int tl_three_opt_find_best_move(tl_ctx *ctx, const float *xy, uint32_t n, const float *dm_packed,
                                const uint32_t *path, int *found, uint32_t *i, uint32_t *j,
                                uint32_t *k, int *kase, float *savings);

/* ---- Lin–Kernighan: replaces lin_kernighan::solve (lin_kernighan.rs:35-100) ----------------- */
/* seed drives the double-bridge kicks (the reference uses an unseeded thread RNG, :73).
 * The search itself is always Euclidean over xy: the reference rebuilds its own matrix from the city coordinates
 * (lin_kernighan.rs:41) whatever problem.distances holds.  dm_packed (optional; pass problem.distances for GEO / EXPLICIT
 * problems) is used exactly where the reference uses problem.distances: the nearest-neighbour seed when init_pos is NULL
 * (:47-55) and the reported total (:99 Solution::new -> tour_length). */
int tl_lk(tl_ctx *ctx, const float *xy, uint32_t n, const float *dm_packed, const uint32_t *init_pos,
          const tl_lk_opts *opts, uint64_t seed, uint32_t *out_pos, float *out_cost, tl_stats *stats);

/* ---- Or-opt: replaces or_opt::solve (or_opt.rs:18-74) — SURVEY.md §8(f) "next" row ------------ */
/* Best-improvement relocation of 1/2/3-city segments (forward and, for 2/3, reversed), threshold -1e-3 (:86). */
int tl_or_opt(tl_ctx *ctx, const float *xy, uint32_t n, const float *dm_packed, const uint32_t *init_pos,
              uint32_t *out_pos, float *out_cost, tl_stats *stats);
/* One find_best_move scan (or_opt.rs:80-164): (delta, seg_start i, insert_after j, seg_len, reversed). */
int tl_or_opt_find_best_move(tl_ctx *ctx, const float *xy, uint32_t n, const float *dm_packed, const uint32_t *path,
                             int *found, float *delta, uint32_t *i, uint32_t *j, uint32_t *seg_len, int *reversed);

/* ---- the solvers with what their progress channel carries listed (ABI v4; tl_lk_live: v5) ------- */
/* The reference's solve() functions take Option<&Sender<ProgressMessage>> (two_opt.rs:10, three_opt.rs:19, lin_kernighan.rs:38;
 * only teeline-qt passes one).  A descent here is one kernel launch (or a device-side state machine), so nothing can be sent
 * while it runs; the *_trace entries return, beside the plain entry's results, the record from which the caller replays the
 * reference's exact message sequence after the fact (the Python mirror under teeline_amd/host, teeline_gpu.hpp and integration/teeline-gpu/gpu.rs do).
 * (nearest_neighbor::solve needs no such entry: its messages — the growing path prefix and the current city per step,
 * nearest_neighbor.rs:32-34,40-42,67-69,72-74 — follow from the finished tour.)
 *
 * tl_two_opt_trace — two_opt.rs:22-24,30-32,53-56,63-65 sends PathUpdate(start), CityChange(path[i]) per outer i of every sweep,
 * PathUpdate(path, new_distance) per improving move and Done.  move_log[m] = (i << 16) | j for swap_2opt(path, i+1, j) in the
 * order the reference applies them, and the word TL_TRACE_SWEEP (0xFFFFFFFF) where a new sweep begins (a row can hold moves of
 * two consecutive sweeps back to back); new_distance = d(p[i],p[j]) + d(p[i+1],p[j+1]) on the path before the move.
 * *log_len = words = stats->moves + stats->sweeps - 1; if it exceeds log_cap the log holds the first log_cap words.  REF_ORDER;
 * coordinates (dm_packed NULL; n beyond tl_two_opt_lds_max_n or > 65535: TL_ERR_UNSUPPORTED) or the matrix form (dm_packed,
 * as tl_two_opt). */
#define TL_TRACE_SWEEP 0xFFFFFFFFu
int tl_two_opt_trace(tl_ctx *ctx, const float *xy, uint32_t n, const float *dm_packed, const uint32_t *init_pos,
                     uint32_t *out_pos, float *out_cost, tl_stats *stats, uint32_t *move_log, uint32_t log_cap,
                     uint32_t *log_len);
/* tl_three_opt_trace — three_opt.rs:34,42,47-49 sends PathUpdate(path, 0.0) for the start path and after every apply_3opt, then
 * Done.  move_log holds 4 words per applied move, in order: i, j, k, case (three_opt.rs:36-45, apply_3opt :186-218);
 * *log_len = moves (= stats->moves); the log holds the first log_cap of them. */
int tl_three_opt_trace(tl_ctx *ctx, const float *xy, uint32_t n, const float *dm_packed, const uint32_t *init_pos,
                       uint32_t *out_pos, float *out_cost, tl_stats *stats, uint32_t *move_log, uint32_t log_cap,
                       uint32_t *log_len);
/* tl_or_opt_trace — or_opt.rs:40-42,62-67,70-72 sends PathUpdate(path, 0.0) for the start path, PathUpdate(path,
 * distances.tour_length(path)) after every apply_relocation, then Done.  move_log holds 4 words per applied move, in order:
 * i (segment start), j (insert after), seg_len, reversed (or_opt.rs:45-51, apply_relocation :172-184); *log_len = moves. */
int tl_or_opt_trace(tl_ctx *ctx, const float *xy, uint32_t n, const float *dm_packed, const uint32_t *init_pos,
                    uint32_t *out_pos, float *out_cost, tl_stats *stats, uint32_t *move_log, uint32_t log_cap,
                    uint32_t *log_len);
/* tl_lk_trace — lin_kernighan.rs:71,90 sends PathUpdate(best_tour, best_dist) after the first lk_pass and after every epoch that
 * improves on it (no Done).  snap_pos holds those tours (snap_cap x n positions, in order), snap_dist their best_dist (the
 * Euclidean tour_distance of :118-122, whatever problem.distances holds); *snap_len counts them all, the buffers hold the first
 * snap_cap.  n < 4: none (:57-59).  With TL_FLAG_LK_ONE_WORKGROUP (a cross-check form that keeps no snapshots on the device) the
 * list is the reference's LAST message alone: the final best tour and its best_dist (*snap_len = 1). */
int tl_lk_trace(tl_ctx *ctx, const float *xy, uint32_t n, const float *dm_packed, const uint32_t *init_pos,
                const tl_lk_opts *opts, uint64_t seed, uint32_t *out_pos, float *out_cost, tl_stats *stats,
                uint32_t *snap_pos, float *snap_dist, uint32_t snap_cap, uint32_t *snap_len);

/* tl_lk_live (ABI v5) — the same messages WHILE the search runs: `progress` is called on the calling thread, between two polls of
 * the device-side search (every 64 rounds: a few milliseconds at n = 13 509), once per best tour the ILS settles on, in order,
 * with the tour (n positions, valid during the call only) and its best_dist — what teeline-qt's channel shows of a multi-second
 * run (teeline-qt/src/solver_engine.rs:412-434).  A call into the SAME context from inside the callback is refused with
 * TL_ERR_BUSY (the running search owns the context's stream and workspace; other contexts are free to use); the callback should
 * return quickly — the GPU idles while it runs.  With TL_FLAG_LK_ONE_WORKGROUP: one call, the
 * final tour.  (The 2-opt / 3-opt / Or-opt descents are single launches of milliseconds: their messages are replayed, above.) */
typedef void (*tl_lk_progress_fn)(void *user, const uint32_t *best_pos, uint32_t n, float best_dist);
int tl_lk_live(tl_ctx *ctx, const float *xy, uint32_t n, const float *dm_packed, const uint32_t *init_pos, const tl_lk_opts *opts,
               uint64_t seed, uint32_t *out_pos, float *out_cost, tl_stats *stats, tl_lk_progress_fn progress, void *user);

/* ---- LK candidate lists: replaces lin_kernighan::build_candidates (lin_kernighan.rs:12-27) -- */
/* out: n x min(k, n-1) u32 — the k-NN buffer of the reference's kd-tree query per city (kdtree.rs:193-212, mod.rs:1839-1889):
 * ascending f32 distance, equal distances in the tree's visiting order.  The tree is the reference's wherever its median
 * selection is unambiguous (no points comparing Equal around a pivot, kdtree.rs:63,301-317); elsewhere the reference's tree is
 * implementation-defined and the (coordinate value, position) order decides.  k <= 64. */
int tl_build_candidates(tl_ctx *ctx, const float *xy, uint32_t n, uint32_t k, uint32_t *out);

/* ---- NN seed: replaces nearest_neighbor::solve (nearest_neighbor.rs:8-76) ---------------------- */
/* First unvisited among the n_nearest closest (stable ties, mod.rs:1848-1855), else the globally
 * nearest unvisited (tie -> lowest position; the reference iterates a HashSet there).  The walk's visited flags
 * live in one CU's LDS (n bytes): n <= ~160 000 (TL_ERR_UNSUPPORTED beyond); n_nearest <= 64.
 * dm_packed NULL: EUC_2D from xy; otherwise every distance is read from the packed matrix (GEO / EXPLICIT problems,
 * distance_matrix.rs:259-297) and xy may be NULL. */
int tl_nearest_neighbor(tl_ctx *ctx, const float *xy, const float *dm_packed, uint32_t n, uint32_t n_nearest,
                        uint32_t *out_pos, float *out_cost);

/* ---- multi-start 2-opt (north-star config 4; no counterpart in the reference) ---------------- */
/* Runs restarts [first, first+count) — restart r starts from the Fisher–Yates permutation drawn
 * from splitmix64(seed + r) (specification: DESIGN.md / oracle tlo_restart_perm) — one descent per
 * workgroup, all concurrently.  Returns the best tour of this shard; out_costs (count floats,
 * optional) gets every restart's final cost.  Ranks shard [0,R) between themselves and min-reduce
 * tl_pack_cost_key(best_cost, best_restart) with RCCL. */
int tl_two_opt_multistart(tl_ctx *ctx, const float *xy, uint32_t n, uint64_t seed, uint32_t first,
                          uint32_t count, int mode, uint32_t *out_best_pos, float *out_best_cost,
                          uint32_t *out_best_restart, float *out_costs, tl_stats *stats);
/* The same job over several devices of one node from ONE process (the reference is single-process; SURVEY.md §8(b)
 * proposed tl_multistart_two_opt(..., n_gpus, ...)): ctxs holds one context per device, each from tl_create(device, ...).
 * Restarts [first, first+count) are dealt in contiguous blocks (the first count % n_ctxs contexts take one more), all
 * shards run concurrently, the winner is the minimum packed key over the shards — results do not depend on n_ctxs.
 * Errors are reported on ctxs[0].  stats->kernel_ms is the slowest shard's device time.
 * tl_multistart_shard: that deal as a host-only query (no context) — block `part` of `parts`; the ranks of a multi-process job
 * (bench.py, teeline_amd/host/multistart.py shard_total) use the same map. */
int tl_multistart_shard(uint32_t first, uint32_t count, int parts, int part, uint32_t *shard_first, uint32_t *shard_count);
int tl_two_opt_multistart_devices(tl_ctx *const *ctxs, int n_ctxs, const float *xy, uint32_t n, uint64_t seed,
                                  uint32_t first, uint32_t count, int mode, uint32_t *out_best_pos, float *out_best_cost,
                                  uint32_t *out_best_restart, float *out_costs, tl_stats *stats);
/* (float_bits(cost) << 32) | restart: order-preserving for cost >= 0, ties -> lowest restart. */
uint64_t tl_pack_cost_key(float cost, uint32_t restart);
/* A population of `count` explicit tours (init_pos: count x n positions), each refined by its own
 * REF_ORDER descent, one workgroup per tour, all concurrently; tour r of out_pos / out_costs equals what
 * tl_two_opt returns for it alone (two_opt.rs:7-67 with init_tour = Some(tour r)).  The reference has no
 * batch form: its callers loop over two_opt::solve (the north-star's GA refinement would, too).
 * dm_packed as in tl_two_opt.  Beyond tl_two_opt_lds_max_n (coordinates) the tours run one after the other through the HBM form. */
int tl_two_opt_population(tl_ctx *ctx, const float *xy, uint32_t n, const float *dm_packed,
                          const uint32_t *init_pos, uint32_t count, uint32_t *out_pos, float *out_costs,
                          tl_stats *stats);

/* ---- device-resident batch entry (bench / pipelines that keep data in HBM) ------------------- */
/* All d_* are DEVICE pointers on the context's device.  d_init: count x n u32 (NULL: seeded restarts
 * first..first+count as above; seed ignored otherwise).  d_out_pos: count x n u32, d_out_cost: count f32,
 * d_out_stats: count x TL_DEV_STATS_STRIDE u64 {sweeps, moves, reversed, status (0 ok, 1 sweep cap reached, 2 d_init holds a
 * position >= n: that descent is refused, its cost is NaN), steps, L0 tile bounds evaluated, candidates into L1, into L2,
 * into L3, shader clocks, 100 MHz ticks, reserved...}.  stream: the hipStream_t to enqueue on,
 * or NULL for the context's own stream — which is NON-BLOCKING, i.e. not ordered with the legacy default stream: a
 * caller that passes NULL must wait on tl_last_kernel_ms() (or a device synchronise) before touching the outputs.
 * Several asynchronous calls on one context may use different streams: the context's shared device workspace (the instance's
 * neighbour lists, grid coordinates) is handed from one kernel sequence to the next through an event, so a call on stream B waits
 * on the device for the previous call's descents on stream A before it rebuilds them (no host synchronisation).
 * Asynchronous: returns after enqueueing — except that a batch with more descents than the device has CUs at 7 100 < n <= 10 240
 * first asks the device whether the coordinates lie on a decimal grid (then two descents share a CU): one 4-byte read-back,
 * i.e. a synchronisation of `stream`, before the descents are enqueued. */
int tl_two_opt_batch_dev(tl_ctx *ctx, const float *d_xy, uint32_t n, const uint32_t *d_init,
                         uint64_t seed, uint32_t first, uint32_t count, int mode, uint32_t *d_out_pos,
                         float *d_out_cost, uint64_t *d_out_stats, void *stream);
/* Device time (ms) between the start and end of the most recent *_dev / solver kernel sequence on
 * this context, from HIP events recorded on the launch stream.  Synchronises on the end event. */
int tl_last_kernel_ms(tl_ctx *ctx, double *ms);
int tl_dm_build_dev(tl_ctx *ctx, const float *d_xy, uint32_t n, int dist, int layout, float *d_out,
                    void *stream);
/* Diagnostics: the 16 kernel-side counters of descent 0 of the most recent tl_two_opt / tl_two_opt_trace / tl_two_opt_population /
 * tl_two_opt_multistart call of this context (host-buffer entries; words 0-4: sweeps, moves, reversed elements, status, steps; the
 * rest is per kernel — matrix form: 5 = steps of sweeps run on the lists, 6 = such sweeps, 7 = their rows that walked the matrix rows).
 * Not part of the reference's interface; tests and timing scripts read which form ran from it. */
int tl_two_opt_last_counters(tl_ctx *ctx, uint64_t out[16]);

#ifdef __cplusplus
}
#endif
#endif /* TEELINE_GPU_H */
