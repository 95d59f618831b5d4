// tl_kernels.h — kernel argument blocks and launch entry points (internal to libteeline_gpu).
#pragma once
#include "tl_device.h"

#ifndef TL_TWO_OPT_NT
#define TL_TWO_OPT_NT 1024  // threads per descent workgroup (16 waves, 4 per SIMD)
#endif

#include <mutex>
#include <set>
#include <utility>

namespace tl {

// MaxDynamicSharedMemorySize is an attribute of the FUNCTION (per device), not of a launch or a stream: setting it per launch
// to the size of that launch lets two host threads solving different n race (set 110 KB, set 30 KB, launch with 110 KB ->
// invalid configuration).  Every launcher therefore raises it ONCE per (kernel, device) to all the LDS the device has left
// beside the kernel's static allocation, and passes its real size only as the launch parameter.
inline hipError_t allow_max_lds(const void *kern)
{
    static std::mutex mu;
    static std::set<std::pair<const void *, int>> done;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> g(mu);
    if (done.count({kern, dev})) return hipSuccess;
    int maxlds = 0;
    if ((e = hipDeviceGetAttribute(&maxlds, hipDeviceAttributeMaxSharedMemoryPerBlock, dev)) != hipSuccess) return e;
    hipFuncAttributes fa;
    if ((e = hipFuncGetAttributes(&fa, kern)) != hipSuccess) return e;
    const int dyn = maxlds - (int)fa.sharedSizeBytes;
    if ((e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, dyn)) != hipSuccess) return e;
    done.insert({kern, dev});
    return hipSuccess;
}

// u64 words per descent in out_stats: sweeps, moves, reversed, status, steps, then profile words
#define TL_STATS_STRIDE 16

enum : uint32_t { TL_INIT_IDENTITY = 0, TL_INIT_ARRAY = 1, TL_INIT_SEEDED = 2 };

// Neighbour lists of an instance (two_opt_nl.hip): what the late sweeps of the LDS descent read instead of walking tiles.
// One 128-byte record of 64 u16 per city, so that a row (a, b) is ONE wave-wide load of two cache lines — lane l reads word l of
// a's record (l < kNlRecB0, except the count) or of b's:
//   [0]       high half of the bits of the KA-th smallest squared distance from the city (i.e. rounded down)
//   [2]       1 if more than kNlRB cities have this one among their KB nearest (the list below is then incomplete), else 0
//   [4 .. 20) its KA nearest cities
//   [20 .. 56) the cities that have it among their KB nearest (the first kNlRB of them), 0xFFFF = empty slot
//   [56 .. 64) unused here: those lanes of a row's pass take the chunk's long cities
constexpr int kNlKA = 16;        // nearest cities kept per city for "c closer to a than b" (rows with a longer (a, b) take the tile path)
constexpr int kNlKB = 24;        // nearest cities per city behind the reverse lists ("b closer to e than c")
constexpr int kNlRecA0 = 4, kNlRecB0 = 20, kNlRB = 36, kNlSurv0 = 56, kNlSurvSlots = 8;
constexpr int kNlLongCap = 128;  // cities with a tour edge beyond their KB-th distance a descent can hold (more: tile path for the sweep)
struct TwoOptNl {
    const uint16_t *rec;     // [n][64] records; nullptr = no lists (tile path only)
    const uint32_t *dkb2;    // [n] bits of the KB-th smallest squared distance
    const uint16_t *knn_b;   // [n][kNlKB] the KB nearest cities of each city (the forward form of the reverse lists; diagnostics)
    const uint32_t *rcnt;    // [n] reverse counts (diagnostics)
    uint32_t sweep_min;      // first sweep of a descent that may run in the late phase (neighbour-list rows) ...
    uint32_t moves_max;      // ... once a sweep has applied fewer moves than this (in the late phase every move is a whole step)
};
size_t two_opt_nl_ws_bytes(uint32_t n);
hipError_t launch_two_opt_nl_build(const float2 *xy, uint32_t n, void *ws, bool fresh, TwoOptNl *out, hipStream_t s, int form = 0);  // form 1: the workgroup-per-city kernel (cross-check)

// two_opt_dm.hip — the matrix form's lists for its late sweeps: per city the 16 nearest by its matrix row and the
// reverse relation, cut from the full matrix once per call and shared by every descent of the batch
constexpr int kDmK = 16;     // nearest cities per city (the c side of a row: D[a][c] < D[a][b])
constexpr int kDmInv = 48;   // reverse-list slots per city (the e side: D[b][e] < D[c][e]; lanes 16..63 of a row's pass)
struct DmLists {
    const uint16_t *id;       // [n][kDmK] nearest cities in (distance, id) order, the city itself excluded; nullptr = no lists
    const float *d;           // [n][kDmK] their distances
    const float *dk;          // [n] the kDmK-th of them (NaN where the row holds a NaN: never listed)
    const uint16_t *inv_id;   // [n][kDmInv] the cities that hold this one among their kDmK (0xFFFF: empty slot)
    const float *inv_d;       // [n][kDmInv] D[that city][this one]
    const uint32_t *inv_cnt;  // [n] entries offered (beyond kDmInv: the list is incomplete, the row walks the matrix row)
    uint32_t long_max;        // a sweep runs on the lists while at most this many cities have a tour edge beyond their dk
    uint32_t moves_max;       // ... and the sweep before it applied at most this many moves
};
size_t dm_lists_ws_bytes(uint32_t n);
bool two_opt_ref_dm_late_fits(uint32_t n, int lds_budget);  // the late sweeps' state fits the LDS beside the tour
size_t two_opt_ref_dm_late_work_bytes(uint32_t n, uint32_t count);  // ... and the descents' per-city records in HBM (TwoOptBatchArgs::work)
hipError_t launch_dm_lists_build(const float *full, uint32_t n, void *ws, DmLists *out, hipStream_t s);

struct TwoOptBatchArgs {
    const float2 *xy;        // n cities, city order
    const float *dm;         // packed lower triangle (matrix kernels) or nullptr
    const float *dm_full;    // the same matrix expanded to row-major n x n (k_dm_expand_full), matrix kernels only
    const uint32_t *init;    // [count][n] when init_mode == TL_INIT_ARRAY
    uint32_t *out_pos;       // [count][n]
    float *out_cost;         // [count]
    uint64_t *out_stats;     // [count][TL_STATS_STRIDE] = sweeps, moves, reversed, status(0 ok, 1 sweep cap), steps, ...
    uint64_t seed;
    uint32_t first;          // restart index of descent 0 (seeded init)
    uint32_t n;
    uint32_t n_pad;
    uint32_t max_sweeps;
    uint32_t init_mode;
    uint32_t *work;          // per-descent global workspace (global-memory variants)
    uint32_t *move_log;      // LDS kernel, optional: [count][log_cap] words — (row << 16 | column) of every applied move in the reference's
    uint32_t log_cap;        // order (two_opt.rs:50), 0xFFFFFFFF where a new sweep begins — what a caller replays the reference's per-move
                             // progress messages from; NULL = off
    const uint2 *fx_xy;      // LDS kernel, grid-coordinate form: per city {x (20 bits) | y low 12 bits << 20, y high 8 bits} (k_fx_encode)
    double fx_inv;           // ... fl64(1 / S) of its decimal grid, 0 = plain float2 form
    TwoOptNl nl;             // LDS kernel, 16-wave float2 form: neighbour lists for the late sweeps (rec == nullptr: off)
    DmLists dml;             // matrix kernel: lists for its late sweeps (id == nullptr: off)
};

// two_opt_ref.hip
size_t two_opt_ref_lds_bytes(uint32_t n, uint32_t *n_pad_out, int nt);
size_t two_opt_ref_nl_lds_bytes(uint32_t n);  // ... with the neighbour-list state (city -> position table, long list) beside the tour
// would a batch of `count` descents run in the form that reads neighbour lists (16 waves, float2 points, lists + tour fit the LDS)?
bool two_opt_ref_nl_applies(uint32_t n, uint32_t count, int cus, int lds_budget, int force_nt);
int two_opt_ref_pick_nt(uint32_t n, uint32_t count, int cus, int lds_budget, int force_nt);   // threads per descent of such a batch
bool two_opt_ref_nl_form(uint32_t n, uint32_t count, int cus, int lds_budget, int force_nt);  // ... and whether that form reads the lists
hipError_t launch_two_opt_ref_lds(const TwoOptBatchArgs &A, uint32_t count, bool prune, hipStream_t s, bool count_work, int cus, int lds_budget,
                                  int force_nt);
// grid-coordinate form (A.fx_xy / A.fx_inv set): would it let two tours share a CU where the float2 form cannot?
bool two_opt_ref_fx_pays(uint32_t n, uint32_t count, int cus, int lds_budget);
size_t two_opt_ref_fx_lds_bytes(uint32_t n);
// encode xy on the grid 1/scale: out[c] = packed grid coordinates, *bad += cities that do not decode back bit for bit
hipError_t launch_fx_encode(const float2 *xy, uint32_t n, double scale, uint2 *out, uint32_t *bad, hipStream_t s);

// two_opt_dm.hip — same algorithm, distances gathered from the packed matrix in HBM/L2
size_t two_opt_ref_dm_lds_bytes(uint32_t n);
hipError_t launch_two_opt_ref_dm(const TwoOptBatchArgs &A, uint32_t count, int lds_budget, hipStream_t s);
hipError_t launch_dm_expand_full(const float *packed, uint32_t n, float *full, hipStream_t s);

// two_opt_best.hip — BEST_SWEEP mode (whole chip per sweep)
struct BestSweepArgs {
    const float2 *xy;
    uint32_t *perm;                // [n] in/out (device)
    float2 *P;                     // [n_pad + 1] tour-ordered coordinates
    float4 *tbox;                  // [ntile_cap] L0 boxes
    float *tmsq;                   // [ntile_cap]
    unsigned long long *partials;  // one packed key per scan workgroup
    unsigned long long *rowkey;    // [n] every row's best candidate as of the last sweep that decided it (kNoKey64: none)
    uint32_t *move;                // {a move has been applied, its row is, its column js}: what the next sweep must decide again
    uint64_t *counters;            // sweeps, moves, done, reversed
    uint32_t n, n_pad, ntile_cap, phase;
};
hipError_t launch_best_sweep_init(const BestSweepArgs &A, hipStream_t s);
hipError_t launch_best_sweep_round(const BestSweepArgs &A, hipStream_t s);
uint32_t best_sweep_scan_blocks(uint32_t n);

// two_opt_large.hip — REF_ORDER for tours beyond one CU's LDS
struct LargeTwoOptState {
    unsigned long long key;  // (row << 32) | column of the lexicographically first improving candidate of the round; ~0 = none
    uint32_t i0, j0, rows, improved, sweeps, done, status, pad_;
    uint64_t moves, reversed;
};
struct LargeTwoOptArgs {
    const float2 *xy;
    uint32_t *perm;     // [n] in/out
    float2 *P;          // [n_pad + 1]
    float4 *tbox;       // [ntile_cap]
    float *tmsq;        // [ntile_cap]
    LargeTwoOptState *state;
    uint32_t n, n_pad, ntile_cap, max_sweeps;
};
hipError_t launch_large_two_opt_init(const LargeTwoOptArgs &A, hipStream_t s);
hipError_t launch_large_two_opt_round(const LargeTwoOptArgs &A, hipStream_t s);

// A best-improvement descent (3-opt, Or-opt) runs as batches of passes enqueued back to back: the pick kernel of a pass that finds
// no move sets `done`, every kernel of a later pass of the batch returns at once, and the host polls this block once per batch
// instead of once per pass (round 4: the per-pass read-back was 40 % of an Or-opt descent at n = 5 000).
struct ScanRunState {
    uint32_t done;     // a pass found no improving move: the descent is over
    uint32_t passes;   // find_best_move calls so far (the last one finds nothing)
    uint32_t moves;    // moves applied; also the number of log entries offered
    uint32_t log_cap;  // entries (4 words each) the move log holds
};

// three_opt.hip
struct ThreeOptBest {
    float sav;
    uint32_t ij;     // i << 16 | j
    uint32_t kc;     // k << 3 | case (1..7); 0xFFFFFFFF = none
    uint32_t found;
};
struct ThreeOptArgs {
    const float2 *xy;
    const float *dm;             // packed matrix or nullptr
    uint32_t *perm;              // [n] tour positions, updated in place by k_three_opt_pick
    float2 *Pt;                  // [n+1] tour-ordered coordinates (coordinate form)
    float *E;                    // [n] tour-edge lengths, E[n-1] = closing edge
    float *Dt;                   // [n][n+1] distances between tour POSITIONS, column n == column 0 (rebuilt every pass)
    const uint32_t *chunk_prefix;  // [n-1]: chunks of rows < i
    ThreeOptBest *partials;      // one per scan workgroup
    ThreeOptBest *best;          // result of the pass
    uint64_t *counters;          // passes, moves
    uint32_t *scratch;           // [n] the move's two segments in k_three_opt_pick where they do not fit the LDS
    ScanRunState *run;           // a descent's batch state (nullptr: a single find_best_move)
    uint32_t *log;               // [run->log_cap][4] applied moves: i, j, k, case
    uint32_t n;
    uint32_t jc;                 // j values per scan workgroup
};
size_t three_opt_scan_lds_bytes(uint32_t n);
hipError_t launch_three_opt_pass(const ThreeOptArgs &A, uint32_t nblocks, bool dm, int apply, hipStream_t s, int lds_budget);

// or_opt.hip
struct OrOptBest {
    uint32_t found, delta_bits, i, j, seg_len, reversed;
};
struct OrOptArgs {
    const float2 *xy;
    const float *dm;               // packed matrix or nullptr
    uint32_t *perm;                // [n] tour positions, updated in place by k_or_pick
    float2 *Pt;                    // [n] tour-ordered coordinates
    float *E;                      // [n] tour-edge lengths, E[n-1] = closing edge
    unsigned long long *partials;  // one packed key (two words: ~delta bits, loop-order index) per scan workgroup
    OrOptBest *best;
    uint32_t *scratch;             // [n] the pre-move tour of k_or_pick where it does not fit the LDS
    ScanRunState *run;             // a descent's batch state (nullptr: a single find_best_move)
    uint32_t *log;                 // [run->log_cap][4] applied moves: i, j, seg_len, reversed
    uint32_t n;
};
hipError_t launch_or_opt_pass(const OrOptArgs &A, bool dm, int apply, hipStream_t s, int lds_budget);
uint32_t or_opt_scan_blocks(uint32_t n);

// lk.hip
struct LkState {             // device-side state machine of the multi-CU LK variant
    uint32_t key;            // min pair index with a valid chain in the current scan (0xFFFFFFFF: none)
    uint32_t finished;
    uint32_t stage;          // 0: initial lk_pass, 1: ILS epochs
    uint32_t epoch, platoo;
    float best_dist;
    uint64_t draws;
    uint64_t scans, searches, moves, exchanged;
    uint32_t window;         // pairs [0, window) the next scan looks at (a prefix: the lowest pair index wins anyway)
    uint32_t applied;        // k_lk_control applied a move into `alt`: k_lk_rebuild copies it back and rebuilds pos / next / prev
    uint32_t key2[2];        // chip-wide step: the scan's key, double-buffered by round parity
    uint32_t flip, flip_next;  // chip-wide step: which tour buffer is current (0: tour, 1: alt); handed over by the next scan
    uint32_t snaps;          // best tours recorded so far (LkArgs::snap; counted on beyond snap_cap)
    uint32_t pass_moves;     // moves of the lk_pass in progress (finished = 2 beyond lk_pass_cap: the pass is cycling)
};
struct LkArgs {
    const float2 *xy;
    const uint32_t *cand;   // [n][k] candidate lists, ascending distance
    uint32_t *tour;         // [n] working tour (in: initial tour)
    uint32_t *alt;          // [n] scratch for chain application
    uint32_t *pos;          // [n] rank of city
    uint32_t *next;         // [n]
    uint32_t *prev;         // [n]
    uint32_t *city_ids;     // [n] scan order snapshot of a pass
    uint32_t *best;         // [n] out: best tour
    uint64_t *counters;     // scans, searches, moves, exchanged edges
    LkState *state;         // multi-CU variant
    uint32_t *chains;       // multi-CU variant: [2n][kLkMaxChain + 2] chain slots
    uint32_t *pairmin;      // split scan: [2n] minimum sub-search index with a chain (0xFFFFFFFF: none); nullptr = unsplit scan
    uint32_t *subchains;    // split scan: [2n * k(k+1)][16] the chain each successful sub-search found (len, cities), or nullptr:
                            // k_lk_scan_pick then reads the winner's chain instead of walking it again
    uint64_t seed;
    uint32_t n, k, max_depth, epochs, platoo_epochs;
    uint32_t lds_budget;    // LDS bytes a scan workgroup may use for its xy/next copies
    uint32_t split_levels;  // split scan: 2 = k(k+1) sub-searches per pair, 3 = k(k+1)^2
    uint32_t chip_step;     // fused scan at n >= 1500: k_lk_control<true> (state machine + move application, chip-wide) replaces
                            // k_lk_control<false> + k_lk_rebuild
    uint32_t parity;        // round & 1 (set per launch)
    uint32_t fused_pick;    // split scan, one workgroup per pair: the workgroup picks its first chain and validates it itself
                            // (no k_lk_scan_pick launch, no pairmin / subchains traffic)
    uint32_t *snap;         // optional [snap_cap][n]: every best tour the search settles on, in order — what the reference sends as
    float *snap_dist;       // PathUpdate(best_tour, best_dist) (lin_kernighan.rs:71,90) — and its best_dist; nullptr = not recorded
    uint32_t snap_cap;
    uint32_t snap_ring;      // snap is a ring of snap_cap slots (tl_lk_live) instead of a list of the first snap_cap
    uint32_t persist_blocks; // fused three-level scan: workgroups of the persistent grid (k_lk_scan_persist); 0 = one workgroup per pair
    // k_lk_ils (the LDS-resident single-workgroup ILS of a small instance): records per level queue (lk_ils_qcap), scans per launch,
    // and how many snapshots of the ring the host has taken so far (a slice ends when the ring is full of undelivered ones)
    // the packed view of the chip-wide scan (lk.hip LkViewPk): candidate ids with their distances, successor records; nullptr = classic view
    uint2 *candd;
    float4 *nx;
    uint32_t ils_qcap, ils_slice, snap_delivered;
    uint32_t ils_threads;    // 0: by n (256 up to n = 200, else 1024); tuning: 256 / 512 / 1024
    // speculative epochs (k_lk_ils mode 2 + k_lk_ils_commit): ils_P workgroups = ils_P consecutive epochs kicked from the same best tour
    uint32_t ils_mode, ils_P;
    float *ils_dcand;        // [n][k] the candidates' distances (written by the first-pass launch, read by the epoch launches)
    uint32_t *ils_ep_tour;   // [ils_P][n] the tour each epoch ends on
    float *ils_ep_dist;      // [ils_P] its tour_distance
    uint64_t *ils_ep_cnt;    // [ils_P][4] scans, searches, moves, exchanged edges of the epoch
};
// form: 0 = default (16 lanes per city up to n = 32 K, 4 beyond), 4 = four lanes per city, 1 = one lane per city
hipError_t launch_knn(const float2 *xy, uint32_t n, uint32_t k, uint32_t *cand, hipStream_t s, int form = 0);
hipError_t launch_nn_seed(const float2 *xy, uint32_t n, const uint32_t *cand, uint32_t k, uint32_t *path, int lds_bytes, hipStream_t s);
hipError_t launch_nn_seed_dm(const float *dm, uint32_t n, uint32_t *path, int lds_bytes, hipStream_t s);
// small: the LDS-resident form (n (36 + 4k) bytes of LDS, lk_small_lds_bytes) with `threads` in {64, 256, 1024}
hipError_t launch_lk_solve(const LkArgs &G, hipStream_t s, bool small = false, int threads = 1024);
size_t lk_small_lds_bytes(uint32_t n, uint32_t k);
hipError_t launch_lk_begin(const LkArgs &G, hipStream_t s);
hipError_t launch_lk_round(const LkArgs &G, hipStream_t s, uint32_t round);
uint32_t lk_ils_qcap(uint32_t n, uint32_t k, uint32_t max_depth, size_t lds_budget);  // > 0: k_lk_ils applies (records per level queue)
size_t lk_ils_lds_bytes(uint32_t n, uint32_t k, uint32_t max_depth, uint32_t qcap);   // LDS of one k_lk_ils workgroup
hipError_t launch_lk_ils(const LkArgs &G, hipStream_t s);                             // mode 0 / 1: one slice of G.ils_slice scans; mode 2: ils_P epochs
hipError_t launch_lk_ils_commit(const LkArgs &G, hipStream_t s);                      // the verdict over a batch of epochs
size_t lk_chain_slot_words();
size_t lk_sub_slot_words();
uint32_t lk_max_depth();  // the build's compile-time recursion bound (6)

}  // namespace tl
// lk_deep.hip — lk.hip compiled with chains of up to 16 exchanges (max_depth 7..16)
namespace tl_lk_deep {
hipError_t launch_lk_solve(const tl::LkArgs &G, hipStream_t s, bool small = false, int threads = 1024);
size_t lk_small_lds_bytes(uint32_t n, uint32_t k);
hipError_t launch_lk_begin(const tl::LkArgs &G, hipStream_t s);
hipError_t launch_lk_round(const tl::LkArgs &G, hipStream_t s, uint32_t round);
uint32_t lk_ils_qcap(uint32_t n, uint32_t k, uint32_t max_depth, size_t lds_budget);
size_t lk_ils_lds_bytes(uint32_t n, uint32_t k, uint32_t max_depth, uint32_t qcap);
hipError_t launch_lk_ils(const tl::LkArgs &G, hipStream_t s);
hipError_t launch_lk_ils_commit(const tl::LkArgs &G, hipStream_t s);
size_t lk_chain_slot_words();
size_t lk_sub_slot_words();
uint32_t lk_max_depth();
}  // namespace tl_lk_deep
namespace tl {

// kdtree.hip — build_candidates through the reference's kd-tree (kdtree.rs)
struct KdNode {
    float x, y;           // the node's point
    int32_t left, right;  // node indices, -1 = None
    uint32_t pos;         // its position in the city array (= KDPoint.id of build_candidates' cities)
    uint32_t coord;       // depth % 2
};
size_t kdtree_build_ws_bytes(uint32_t n, size_t *cub_bytes_out);
hipError_t kdtree_build_dev(const float2 *xy, uint32_t n, void *ws, KdNode *nodes, hipStream_t s);  // root = node n / 2
hipError_t launch_knn_kdtree(const KdNode *nodes, const float2 *xy, uint32_t n, uint32_t k, uint32_t *cand, hipStream_t s);

// dm_build.hip
hipError_t launch_dm_build(const float2 *xy, uint32_t n, int dist, int layout, float *out, hipStream_t s);
hipError_t launch_dm_compare(const float2 *xy, uint32_t n, const float *dm, uint32_t *differs, hipStream_t s);
hipError_t launch_selftest_sqrt(uint32_t first_bits, uint64_t count, unsigned long long *mismatches, uint32_t *first_bad, hipStream_t s);
hipError_t launch_tour_length(const float2 *xy, const float *dm, uint32_t n, const uint32_t *perm,
                              float *out_cost, hipStream_t s);

}  // namespace tl
