// tl_api_lk.hip — C ABI: candidate lists (lin_kernighan::build_candidates, lin_kernighan.rs:12-27), the NN seed (nearest_neighbor.rs:8-76)
// and Lin-Kernighan (lin_kernighan.rs:35-100) with its trace / live-progress entries.
#include "tl_api_common.h"

using namespace tl;
using namespace tlapi;

// lin_kernighan cut-over (measured, NOTEBOOK.md §4.6): the LDS-resident single workgroup never wins -> 0
static constexpr uint32_t kLkSmallMaxN = 0, kLkSmallWave64MaxN = 0, kLkSmall256MaxN = 0;
#ifndef TL_LK_ILS_MAX_N
#define TL_LK_ILS_MAX_N 700u  // largest n the LDS-resident ILS (k_lk_ils) takes by default ...
#endif
static constexpr uint32_t kLkIlsMaxN = TL_LK_ILS_MAX_N;
static constexpr uint32_t kLkIlsMaxNLongPlateau = 65535u;  // ... and, with a plateau of >= 64 epochs (the speculative epochs fill the chip), whatever fits one CU's LDS (n ~ 3 400 at k = 3, ~2 900 at k = 5)
static constexpr uint32_t kLkIlsSlice = 8192u;  // scans per launch of k_lk_ils (tens of milliseconds)
static constexpr uint32_t kLkIlsEpochScans = 512u;  // scans a speculative epoch may take before it files "unfinished" (x 8 per retry as the next epoch)
static constexpr int kLkIlsBatches = 16;         // batches of speculative epochs per poll of the state (at most one progress message per batch: the ring holds 64)

static bool max_depth_ge2_split(uint32_t) { return true; }

// lin_kernighan::build_candidates (lin_kernighan.rs:12-27) into d_cand (n x k): the reference's kd-tree k-NN — tree built and
// queried on the device (kdtree.hip) — or, under the TL_FLAG_KNN_* flags, the
// brute-force scan in (distance, position) order (identical lists wherever no two candidates of a city tie in f32 distance).
// d_xy must already hold xy (enqueued on c->stream).
int tlapi::build_candidates_dev(tl_ctx *c, const float *xy_host, const float2 *d_xy, uint32_t n, uint32_t k, uint32_t *d_cand)
{
    if (k == 0) return TL_OK;
    if ((c->flags & TL_FLAG_KNN_BRUTE) | (tune_flags(c) & (TL_FLAG_KNN_4LANES | TL_FLAG_KNN_1LANE))) {
        HIPCHK(c, launch_knn(d_xy, n, k, d_cand, c->stream, knn_form(c)));
        return TL_OK;
    }
    (void)xy_host;
    int rc;
    const size_t nodes_b = (((size_t)n * sizeof(KdNode)) + 255) & ~(size_t)255;
    if ((rc = ensure(c, c->kd, nodes_b + kdtree_build_ws_bytes(n, nullptr)))) return rc;
    KdNode *nodes = (KdNode *)c->kd.p;
    HIPCHK(c, kdtree_build_dev(d_xy, n, (unsigned char *)c->kd.p + nodes_b, nodes, c->stream));
    HIPCHK(c, launch_knn_kdtree(nodes, d_xy, n, k, d_cand, c->stream));
    return TL_OK;
}  // the split scan handles every max_depth >= 1

// ------------------------------------------------------------------------------------------------
// candidate lists, NN seed, Lin-Kernighan
// ------------------------------------------------------------------------------------------------
extern "C" int tl_build_candidates(tl_ctx *c, const float *xy, uint32_t n, uint32_t k, uint32_t *out)
{
    TL_ENTER(c);
    if (!c || !xy || !out) return fail(c, TL_ERR_BADARG, "tl_build_candidates: NULL argument");
    if (n == 0) return fail(c, TL_ERR_BADARG, "tl_build_candidates: n == 0");
    if (k > n - 1) k = n - 1;  // lin_kernighan.rs:14
    if (k == 0) return TL_OK;
    if (k > 64) return fail(c, TL_ERR_UNSUPPORTED, "tl_build_candidates: k=%u > 64 (the k-nearest buffer of a query lives in registers; the reference's k is unbounded)", k);
    HIPCHK(c, hipSetDevice(c->device));
    int rc;
    if ((rc = ensure(c, c->xy, (size_t)n * 8)) || (rc = ensure(c, c->misc, (size_t)n * k * 4))) return rc;
    HIPCHK(c, hipMemcpyAsync(c->xy.p, xy, (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
    if ((rc = build_candidates_dev(c, xy, (const float2 *)c->xy.p, n, k, (uint32_t *)c->misc.p))) return rc;
    HIPCHK(c, hipMemcpyAsync(out, c->misc.p, (size_t)n * k * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return TL_OK;
}

// device-side NN seed into d_path (n u32); candidate lists are rebuilt with k = n_nearest in c->misc
int tlapi::nn_seed_dev(tl_ctx *c, const float2 *d_xy, uint32_t n, uint32_t n_nearest, uint32_t *d_path)
{
    uint32_t k = n_nearest > n - 1 ? n - 1 : n_nearest;
    if (k > 64) return fail(c, TL_ERR_UNSUPPORTED, "nearest_neighbor: n_nearest=%u > 64 (the k-nearest buffer of a query lives in registers; the reference's is unbounded)", n_nearest);
    int rc;
    if ((size_t)n + 1024 > (size_t)c->lds_bytes)
        return fail(c, TL_ERR_UNSUPPORTED, "nearest_neighbor: n=%u exceeds the LDS-resident visited flags (%d bytes of LDS)", n, c->lds_bytes);
    // The walk takes "the first unvisited among the n_nearest closest, else the globally nearest unvisited" — both in the
    // same (distance, position) order, so the tour does not depend on how long the lists are: any length gives "the first
    // unvisited city in (distance, position) order", and a longer list only turns workgroup-wide fallback scans into
    // list steps.  The list length used on the device is therefore what fits the LDS best (lists as u16 next to the
    // visited flags; the fallback scans hold their coordinates in registers up to n = 16 384).
    if (n <= 16384u && n - 1u >= 1u) {
        uint32_t kint = k;
        const size_t cap = (size_t)c->lds_bytes - 1024;
        // measured: 7 at n = 10^4 (6.5 -> 5.6 ms), 5 at n = 13 509 (9.3 -> 7.8 ms: what fits), 4 below ~8 K (the k <= 4 list
        // builder is the cheaper one and few steps fall back there)
        for (uint32_t kk = n < 8192u ? 4u : 7u; kk > k; --kk)
            if (kk <= n - 1u && (size_t)n + 16 + (size_t)n * kk * 2u + 16 <= cap) { kint = kk; break; }
        k = kint;
    }
    const size_t cand_b = ((size_t)n * (k ? k : 1) * 4 + 255) & ~(size_t)255;
    if ((rc = ensure(c, c->misc, cand_b))) return rc;
    uint32_t *d_cand = (uint32_t *)c->misc.p;
    if (k) HIPCHK(c, launch_knn(d_xy, n, k, d_cand, c->stream, knn_form(c)));
    HIPCHK(c, launch_nn_seed(d_xy, n, d_cand, k, d_path, c->lds_bytes, c->stream));
    return TL_OK;
}

extern "C" int tl_nearest_neighbor(tl_ctx *c, const float *xy, const float *dm_packed, uint32_t n, uint32_t n_nearest,
                                   uint32_t *out_pos, float *out_cost)
{
    TL_ENTER(c);
    if (!c || (!xy && !dm_packed) || !out_pos) return fail(c, TL_ERR_BADARG, "tl_nearest_neighbor: NULL argument");
    if (n == 0) return fail(c, TL_ERR_REF_PANICS, "nearest_neighbor: cities[0] on an empty problem (nearest_neighbor.rs:28)");
    if (n == 1) {  // the walk is [cities[0]]; tour_length of fewer than two cities is 0 (distance_matrix.rs:236-238)
        out_pos[0] = 0;
        if (out_cost) *out_cost = 0.0f;
        return TL_OK;
    }
    HIPCHK(c, hipSetDevice(c->device));
    int rc;
    if ((rc = ensure(c, c->out_pos, (size_t)n * 4)) || (rc = ensure(c, c->out_cost, 4))) return rc;
    const float2 *dxy = nullptr;
    const float *ddm = nullptr;
    if (dm_packed) {
        if ((size_t)n + 1024 > (size_t)c->lds_bytes)
            return fail(c, TL_ERR_UNSUPPORTED, "nearest_neighbor: n=%u exceeds the LDS-resident visited flags (%d bytes of LDS)", n, c->lds_bytes);
        const size_t b = (size_t)n * (n - 1) / 2 * 4;
        if ((rc = ensure(c, c->dm, b))) return rc;
        HIPCHK(c, hipMemcpyAsync(c->dm.p, dm_packed, b, hipMemcpyHostToDevice, c->stream));
        ddm = (const float *)c->dm.p;
        HIPCHK(c, hipEventRecord(c->ev0, c->stream));
        HIPCHK(c, launch_nn_seed_dm(ddm, n, (uint32_t *)c->out_pos.p, c->lds_bytes, c->stream));
    } else {
        if (!xy) return fail(c, TL_ERR_BADARG, "tl_nearest_neighbor: xy is NULL");
        if ((rc = ensure(c, c->xy, (size_t)n * 8))) return rc;
        HIPCHK(c, hipMemcpyAsync(c->xy.p, xy, (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
        dxy = (const float2 *)c->xy.p;
        HIPCHK(c, hipEventRecord(c->ev0, c->stream));
        if ((rc = nn_seed_dev(c, dxy, n, n_nearest, (uint32_t *)c->out_pos.p))) {
            c->ev_valid = false;
            return rc;
        }
    }
    HIPCHK(c, hipEventRecord(c->ev1, c->stream));
    c->ev_valid = true;
    HIPCHK(c, hipMemcpyAsync(out_pos, c->out_pos.p, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    if (out_cost) {
        HIPCHK(c, launch_tour_length(dxy, ddm, n, (const uint32_t *)c->out_pos.p, (float *)c->out_cost.p, c->stream));
        HIPCHK(c, hipMemcpyAsync(out_cost, c->out_cost.p, 4, hipMemcpyDeviceToHost, c->stream));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return TL_OK;
}

// snap_pos / snap_dist (optional): every best tour the search settles on, in order, and its best_dist — what the reference sends
// as PathUpdate(best_tour, best_dist) (lin_kernighan.rs:71,90); *snap_len counts them all, the buffers hold the first snap_cap
// live (optional): called on the calling thread for every such tour WHILE the search runs — the host polls the device-side state
// machine every 64 rounds and drains a ring of snapshots (then snap_pos is the caller-less staging buffer of this function)
static int lk_run(tl_ctx *c, const float *xy, uint32_t n, const float *dm_packed, const uint32_t *init_pos, const tl_lk_opts *opts,
                  uint64_t seed, uint32_t *out_pos, float *out_cost, tl_stats *stats, uint32_t *snap_pos, float *snap_dist, uint32_t snap_cap,
                  uint32_t *snap_len, tl_lk_progress_fn live = nullptr, void *live_user = nullptr)
{
    std::vector<uint32_t> live_tour;
    if (live) {  // a ring of 64 snapshots on the device: at most one snapshot per round, 64 rounds per poll
        snap_cap = 64;
        live_tour.resize(n ? n : 1);
        snap_pos = live_tour.data();  // (only marks "snapshots wanted"; the ring is drained slot by slot below)
    }
    if (snap_len) *snap_len = 0;
    if (!c || !xy || !out_pos) return fail(c, TL_ERR_BADARG, "tl_lk: NULL argument");
    if (n == 0) return fail(c, TL_ERR_BADARG, "tl_lk: n == 0");
    tl_lk_opts o{100, 10, 5, 5};  // LKOptions::default(), mod.rs:1255-1267
    if (opts) o = *opts;
    if (o.n_nearest == 0) return fail(c, TL_ERR_BADARG, "n_nearest must be >= 1");   // mod.rs:677-682
    if (o.max_depth == 0) return fail(c, TL_ERR_BADARG, "max_depth must be >= 1");   // mod.rs:1270-1276
    // chains of up to 6 exchanges live in registers (lk.hip); 7..16 run the same kernels built with larger chain arrays (lk_deep.hip)
    const bool deep = o.max_depth > lk_max_depth();
    if (o.max_depth > tl_lk_deep::lk_max_depth())
        return fail(c, TL_ERR_UNSUPPORTED, "tl_lk: max_depth=%u > %u (the largest chain this build holds; the reference's max_depth is unbounded, mod.rs:1252)",
                    o.max_depth, tl_lk_deep::lk_max_depth());
    if (o.n_nearest > 64) return fail(c, TL_ERR_UNSUPPORTED, "tl_lk: n_nearest=%u > 64 (the k-nearest buffer of a query lives in registers; the reference's n_nearest is unbounded, mod.rs:1252)", o.n_nearest);
    if (init_pos && !is_permutation(init_pos, n)) return fail(c, TL_ERR_BADARG, "tl_lk: init tour is not a permutation of 0..n-1");
    const auto t0 = std::chrono::steady_clock::now();
    if (stats) memset(stats, 0, sizeof(*stats));
    if (n < 4) {
        // lin_kernighan.rs:45-59: the initial tour (given, or the NN seed over problem.distances) is returned untouched
        int rc;
        if (init_pos) memcpy(out_pos, init_pos, (size_t)n * 4);
        else if ((rc = tl_nearest_neighbor(c, xy, dm_packed, n, 3, out_pos, nullptr))) return rc;
        if (out_cost && (rc = tl_tour_length(c, dm_packed ? nullptr : xy, dm_packed, n, out_pos, out_cost))) return rc;
        if (stats) stats->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        return TL_OK;
    }
    HIPCHK(c, hipSetDevice(c->device));
    int rc;
    const uint32_t k = o.n_nearest > n - 1 ? n - 1 : o.n_nearest;
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t arr = up((size_t)n * 4);
    const size_t o_cand = 0, o_tour = up((size_t)n * (k ? k : 1) * 4), o_alt = o_tour + arr, o_pos = o_alt + arr, o_next = o_pos + arr,
                 o_prev = o_next + arr, o_ids = o_prev + arr, o_best = o_ids + arr, o_cnt = o_best + arr, o_state = o_cnt + 256,
                 o_chains = o_state + 256;
    // default: scans spread over all CUs; TL_FLAG_LK_ONE_WORKGROUP runs the whole ILS in one persistent workgroup instead
    // (kept as a cross-check of the state machine)
    // An LDS-resident single-workgroup form (k_lk_solve<NT, true>) exists for small instances; measured on MI355X it loses to
    // the chip-wide scans at every size (scripts/timing_lk.py), so the cut-over kLkSmallMaxN is 0 and the
    // form only runs under TL_FLAG_LK_SMALL (a cross-check).
    const uint32_t k_small = o.n_nearest > n - 1 ? n - 1 : o.n_nearest;
    uint32_t small_max_n = kLkSmallMaxN;
    int small_nt = n <= kLkSmallWave64MaxN ? 64 : (n <= kLkSmall256MaxN ? 256 : 1024);
#ifdef TL_TUNE  // tuning builds only (python -m teeline_amd.build --tune): the product library never reads the environment
    if (const char *e = getenv("TL_LK_SMALL_MAX_N")) small_max_n = (uint32_t)atoi(e);
    if (const char *e = getenv("TL_LK_SMALL_NT")) small_nt = atoi(e);
#endif
    const uint32_t tf = tune_flags(c);  // rejected forms: tuning build only
    const uint32_t variant_flags = TL_FLAG_LK_ONE_WORKGROUP | TL_FLAG_LK_NO_SPLIT | TL_FLAG_LK_SPLIT2 | TL_FLAG_LK_NO_SUBCHAINS | TL_FLAG_LK_SEPARATE_PICK | TL_FLAG_LK_NO_GRAPH | TL_FLAG_LK_SEPARATE_STEP | TL_FLAG_LK_SCAN_PERSIST;
    const bool lk_small = ((tf & TL_FLAG_LK_SMALL) || (!((c->flags | tf) & variant_flags) && n <= small_max_n)) &&
                          lk_small_lds_bytes(n, k_small) + 4096 <= (size_t)c->lds_bytes;
    // Small instances (round 5): the whole ILS as ONE persistent workgroup with every array in LDS and a level-synchronous search
    // (k_lk_ils, lk.hip) — no kernel boundary per round.  Wherever its state and its level queues fit one CU's LDS, up to the size
    // from which the chip-wide scans are faster (kLkIlsMaxN, measured: scripts/timing_lk_ils.py); TL_FLAG_LK_ILS_LDS: wherever it
    // fits; TL_FLAG_LK_CHIP_WIDE: never.
    const uint32_t ils_qcap = deep ? tl_lk_deep::lk_ils_qcap(n, k_small, o.max_depth, (size_t)c->lds_bytes) : lk_ils_qcap(n, k_small, o.max_depth, (size_t)c->lds_bytes);
    // (the cut-over: one workgroup's scan loses to the chip-wide one from n ~ 800 — scripts/timing_lk_ils.py: n = 400 6.6 against 14.2 ms,
    //  n = 1000 27.5 against 21.1 with the library's default 100 epochs / plateau 10 — unless the plateau is long enough for the
    //  speculative epochs to fill the chip: the CLI's 10 000 / 500 runs 256-512 epochs at once)
    const bool ils_size = n <= kLkIlsMaxN || (n <= kLkIlsMaxNLongPlateau && o.platoo_epochs >= 64u && o.epochs >= 64u);
    const bool lk_ils = ils_qcap != 0u && !lk_small && !((c->flags | tf) & (variant_flags | TL_FLAG_LK_CHIP_WIDE)) &&
                        (ils_size || (c->flags & TL_FLAG_LK_ILS_LDS));
    const bool multi_cu = !(c->flags & TL_FLAG_LK_ONE_WORKGROUP) && !lk_small && !lk_ils;
    const size_t slot_words = deep ? tl_lk_deep::lk_chain_slot_words() : lk_chain_slot_words();
    const size_t sub_bytes = (deep ? tl_lk_deep::lk_sub_slot_words() : lk_sub_slot_words()) * 4;  // 64 at depth <= 6
    const size_t o_pairmin = o_chains + (multi_cu ? up((size_t)2 * n * slot_words * 4) : 0);
    const bool split_scan = multi_cu && max_depth_ge2_split(o.max_depth) && !(tf & TL_FLAG_LK_NO_SPLIT);
    // every successful sub-search keeps its chain (64 B) so that the pick step does not walk the winner again; sized for
    // 288 GB of HBM (45 MB at n = 13 509, k = 5), skipped beyond 4 GB
    // three split levels (k(k+1)^2 sub-searches per pair: the sequential part of a walk shrinks to k^2 nodes) while their
    // kept chains fit 4 GB, else two
    const uint32_t levels = (split_scan && !(tf & TL_FLAG_LK_SPLIT2) && (size_t)2 * n * k * (k + 1) * (k + 1) * sub_bytes <= ((size_t)4 << 30)) ? 3u : 2u;
    const size_t sub_b = split_scan ? (size_t)2 * n * k * (k + 1) * (levels == 3u ? k + 1 : 1) * sub_bytes : 0;
    // one workgroup per pair (k(k+1)^2 or k(k+1) <= 1024 threads): the scan picks and validates the pair's first chain itself
    const bool fused_pick = split_scan && (size_t)k * (k + 1) * (levels == 3u ? k + 1 : 1) <= 1024 &&
                            !(tf & (TL_FLAG_LK_SEPARATE_PICK | TL_FLAG_LK_NO_SUBCHAINS));
    const bool keep_sub = split_scan && !fused_pick && sub_b <= ((size_t)4 << 30) && !(tf & TL_FLAG_LK_NO_SUBCHAINS);
    const size_t o_sub = o_pairmin + (split_scan ? up((size_t)2 * n * 4) : 0);
    // k_lk_ils with speculative epochs: the candidates' distances and, per epoch of a batch, its final tour, length and counters
    const bool ils_spec = lk_ils && !(c->flags & TL_FLAG_LK_NO_SPECULATION);
    const size_t ils_lds = lk_ils ? (deep ? tl_lk_deep::lk_ils_lds_bytes(n, k_small, o.max_depth, ils_qcap) : lk_ils_lds_bytes(n, k_small, o.max_depth, ils_qcap)) : 0;
    const uint32_t ils_P = ils_spec ? (uint32_t)c->cus * (2 * ils_lds <= (size_t)c->lds_bytes ? 2u : 1u) : 0u;  // two epochs per CU where their images fit
    const size_t o_ils_dc = o_sub + (keep_sub ? up(sub_b) : 0), o_ils_tour = o_ils_dc + (ils_spec ? up((size_t)n * k_small * 4) : 0),
                 o_ils_dist = o_ils_tour + up((size_t)ils_P * n * 4), o_ils_cnt = o_ils_dist + up((size_t)ils_P * 4);
    // the packed view of the chip-wide scan (LkViewPk): candidates with their distances, successor records
    const bool chip_step_form = fused_pick && levels == 3u && n >= 1500u && !(tf & TL_FLAG_LK_SEPARATE_STEP);
    const bool packed_view = chip_step_form && !(c->flags & TL_FLAG_LK_CLASSIC_VIEW);
    const size_t o_pk_c = o_ils_cnt + up((size_t)ils_P * 32), o_pk_n = o_pk_c + (packed_view ? up((size_t)n * k * 8) : 0);
    const size_t total = o_pk_n + (packed_view ? up((size_t)n * 16) : 0);
    // every mode / size check and every allocation comes before the first event record and the first enqueue: a rejected call
    // leaves the previous kernel sequence's event pair intact and nothing in flight
    // (the single-workgroup forms keep no snapshots on the device: a trace of theirs is the final best tour alone, below)
    const bool snap_dev = snap_pos && (multi_cu || lk_ils);
    if (!init_pos && dm_packed && (size_t)n + 1024 > (size_t)c->lds_bytes)
        return fail(c, TL_ERR_UNSUPPORTED, "nearest_neighbor: n=%u exceeds the LDS-resident visited flags", n);
    if ((rc = ensure(c, c->xy, (size_t)n * 8)) || (rc = ensure(c, c->work, total)) || (rc = ensure(c, c->out_cost, 4))) return rc;
    if (snap_dev && ((rc = ensure(c, c->out_pos, (size_t)(snap_cap ? snap_cap : 1) * n * 4)) ||
                     (rc = ensure(c, c->out_stats, (size_t)(snap_cap ? snap_cap : 1) * 4))))
        return rc;
    unsigned char *w = (unsigned char *)c->work.p;
    HIPCHK(c, hipMemcpyAsync(c->xy.p, xy, (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
    const float *ddm = nullptr;
    if (dm_packed) {  // problem.distances of a GEO / EXPLICIT problem: the NN seed and the reported total read it
        const size_t b = (size_t)n * (n - 1) / 2 * 4;
        if ((rc = ensure(c, c->dm, b))) return rc;
        HIPCHK(c, hipMemcpyAsync(c->dm.p, dm_packed, b, hipMemcpyHostToDevice, c->stream));
        ddm = (const float *)c->dm.p;
    }
    HIPCHK(c, hipEventRecord(c->ev0, c->stream));
    c->ev_valid = false;
    if (init_pos) {
        HIPCHK(c, hipMemcpyAsync(w + o_tour, init_pos, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
    } else if (ddm) {
        HIPCHK(c, launch_nn_seed_dm(ddm, n, (uint32_t *)(w + o_tour), c->lds_bytes, c->stream));
    } else {
        // lin_kernighan.rs:47-55: nearest_neighbor::solve with HeuristicOptions::default() (n_nearest = 3)
        if ((rc = nn_seed_dev(c, (const float2 *)c->xy.p, n, 3, (uint32_t *)(w + o_tour)))) return rc;
    }
    if ((rc = build_candidates_dev(c, xy, (const float2 *)c->xy.p, n, k, (uint32_t *)(w + o_cand)))) return rc;  // :43 build_candidates
    HIPCHK(c, hipMemsetAsync(w + o_cnt, 0, 64, c->stream));
    LkArgs G{};
    G.xy = (const float2 *)c->xy.p;
    G.cand = (const uint32_t *)(w + o_cand);
    G.tour = (uint32_t *)(w + o_tour);
    G.alt = (uint32_t *)(w + o_alt);
    G.pos = (uint32_t *)(w + o_pos);
    G.next = (uint32_t *)(w + o_next);
    G.prev = (uint32_t *)(w + o_prev);
    G.city_ids = (uint32_t *)(w + o_ids);
    G.best = (uint32_t *)(w + o_best);
    G.counters = (uint64_t *)(w + o_cnt);
    G.seed = seed;
    G.n = n;
    G.k = k;
    G.max_depth = o.max_depth;
    G.epochs = o.epochs;
    G.platoo_epochs = o.platoo_epochs;
    G.lds_budget = (uint32_t)c->lds_bytes;
    G.state = (LkState *)(w + o_state);
    G.chains = (uint32_t *)(w + o_chains);
    G.pairmin = split_scan ? (uint32_t *)(w + o_pairmin) : nullptr;
    G.subchains = keep_sub ? (uint32_t *)(w + o_sub) : nullptr;
    G.split_levels = levels;
    G.fused_pick = fused_pick ? 1u : 0u;
    // tuning build: the persistent scan grid (measured and rejected, DESIGN.md / NOTEBOOK.md): what the chip holds of these
    // workgroups at 8 waves per SIMD (32 wave slots per CU), or TL_LK_PERSIST_BLOCKS from the environment
    if (tf & TL_FLAG_LK_SCAN_PERSIST) {
        const uint32_t wg_waves = (uint32_t)((k * (k + 1) * (k + 1) + 63) / 64);
        const uint32_t per_cu = wg_waves ? 32u / wg_waves : 0u;
        G.persist_blocks = (fused_pick && levels == 3u && per_cu) ? (uint32_t)c->cus * per_cu : 0u;
#ifdef TL_TUNE
        if (const char *e = getenv("TL_LK_PERSIST_BLOCKS")) G.persist_blocks = G.persist_blocks ? (uint32_t)atoi(e) : 0u;
#endif
    }
    G.chip_step = chip_step_form ? 1u : 0u;
    if (packed_view) {  // (filled by k_lk_begin, kept by the step kernel)
        G.candd = (uint2 *)(w + o_pk_c);
        G.nx = (float4 *)(w + o_pk_n);
    }
    if (snap_dev) {
        G.snap = (uint32_t *)c->out_pos.p;
        G.snap_dist = (float *)c->out_stats.p;
        G.snap_cap = snap_cap;
        G.snap_ring = live ? 1u : 0u;
    }
    uint32_t delivered = 0;  // live: snapshots handed to the callback so far
    if (split_scan) HIPCHK(c, hipMemsetAsync(G.pairmin, 0xFF, (size_t)2 * n * 4, c->stream));
    uint64_t cnt[4] = {0, 0, 0, 0};
    if (lk_ils) {
        // slices of kLkIlsSlice scans: between two of them the host reads the state words (finished, counters, snapshots so far) and,
        // with a progress callback, hands the new best tours on — the reference's send_progress (lin_kernighan.rs:71,90) while it runs
        G.ils_qcap = ils_qcap;
        G.ils_slice = kLkIlsSlice;
        if (ils_spec) {
            G.ils_P = ils_P;
            G.ils_dcand = (float *)(w + o_ils_dc);
            G.ils_ep_tour = (uint32_t *)(w + o_ils_tour);
            G.ils_ep_dist = (float *)(w + o_ils_dist);
            G.ils_ep_cnt = (uint64_t *)(w + o_ils_cnt);
        }
        HIPCHK(c, hipMemsetAsync(G.state, 0, sizeof(LkState), c->stream));
        LkState hs{};
        int rc_loop = TL_OK;
        // Phase 1 (sequential by nature): the first lk_pass — or, without speculation, the whole ILS — in ONE workgroup, slice by slice.
        // Phase 2: the epochs.  A kick starts from the best tour and the RNG draws of its epoch number alone, so the epochs
        // e, e + 1, ... are independent of each other as long as none of them is accepted: ils_P workgroups run ils_P consecutive
        // epochs at once, k_lk_ils_commit takes them in order up to the first accepted one.  Same tours, counters and messages as
        // the sequential loop (lin_kernighan.rs:75-97); what is thrown away is work on otherwise idle CUs.
        auto drain = [&]() -> hipError_t {
            hipError_t e = hipSuccess;
            for (; live && delivered < hs.snaps; ++delivered) {
                const uint32_t at = delivered % snap_cap;
                float bd = 0.0f;
                e = hipMemcpyAsync(live_tour.data(), G.snap + (size_t)at * n, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream);
                if (e == hipSuccess) e = hipMemcpyAsync(&bd, G.snap_dist + at, 4, hipMemcpyDeviceToHost, c->stream);
                if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
                if (e != hipSuccess) break;
                c->in_callback = true;
                live(live_user, live_tour.data(), n, bd);
                c->in_callback = false;
            }
            return e;
        };
        auto ils_launch = [&]() { return deep ? tl_lk_deep::launch_lk_ils(G, c->stream) : launch_lk_ils(G, c->stream); };
        auto ils_commit = [&]() { return deep ? tl_lk_deep::launch_lk_ils_commit(G, c->stream) : launch_lk_ils_commit(G, c->stream); };
        G.ils_mode = ils_spec ? 1u : 0u;
        for (;;) {
            G.snap_delivered = delivered;
            hipError_t e = ils_launch();
            if (e == hipSuccess) e = hipMemcpyAsync(&hs, G.state, sizeof(hs), hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
            if (e == hipSuccess) e = drain();
            if (e != hipSuccess) {
                (void)hipGetLastError();
                rc_loop = fail(c, TL_ERR_HIP, "tl_lk: %s", hipGetErrorString(e));
                break;
            }
            if (hs.finished || (ils_spec && hs.stage == 1u)) break;
        }
        if (rc_loop == TL_OK && ils_spec && !hs.finished) {
            G.ils_mode = 2u;
            G.ils_slice = kLkIlsEpochScans;  // (mode 2: an epoch's scan budget)
            for (;;) {
                hipError_t e = hipSuccess;
                for (int b = 0; b < kLkIlsBatches && e == hipSuccess; ++b) {  // (both kernels are no-ops once `finished` is set)
                    e = ils_launch();
                    if (e == hipSuccess) e = ils_commit();
                }
                if (e == hipSuccess) e = hipMemcpyAsync(&hs, G.state, sizeof(hs), hipMemcpyDeviceToHost, c->stream);
                if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
                if (e == hipSuccess) e = drain();
                if (e != hipSuccess) {
                    (void)hipGetLastError();
                    rc_loop = fail(c, TL_ERR_HIP, "tl_lk: %s", hipGetErrorString(e));
                    break;
                }
                if (hs.finished) break;
            }
        }
        if (rc_loop != TL_OK) return rc_loop;
        if (hs.finished == 2u)
            return fail(c, TL_ERR_NO_CONVERGE, "tl_lk: an lk_pass does not terminate (the reference's loop cycles on this input: chains of rounded f32 gain that lead back to the same tour)");
        cnt[0] = hs.scans;
        cnt[1] = hs.searches;
        cnt[2] = hs.moves;
        cnt[3] = hs.exchanged;
        if (snap_pos && !live) {
            if (snap_len) *snap_len = hs.snaps;
            const uint32_t have = hs.snaps < snap_cap ? hs.snaps : snap_cap;
            if (have) {
                HIPCHK(c, hipMemcpyAsync(snap_pos, G.snap, (size_t)have * n * 4, hipMemcpyDeviceToHost, c->stream));
                if (snap_dist) HIPCHK(c, hipMemcpyAsync(snap_dist, G.snap_dist, (size_t)have * 4, hipMemcpyDeviceToHost, c->stream));
            }
        }
    } else if (!multi_cu) {
        HIPCHK(c, deep ? tl_lk_deep::launch_lk_solve(G, c->stream, lk_small, small_nt) : launch_lk_solve(G, c->stream, lk_small, small_nt));
    } else {
        HIPCHK(c, deep ? tl_lk_deep::launch_lk_begin(G, c->stream) : launch_lk_begin(G, c->stream));
        auto lk_round = [&](uint32_t r) { return deep ? tl_lk_deep::launch_lk_round(G, c->stream, r) : launch_lk_round(G, c->stream, r); };
        LkState hs{};
        // 64 rounds per poll of `finished` (the kernels are no-ops once it is set).  The first batch is enqueued launch by
        // launch; a search that is still running after it replays the same 64 rounds as ONE hipGraph launch per poll — a round
        // is 2-3 short kernels (tens of microseconds), and the host's per-launch cost and the gaps between separately
        // enqueued kernels are a visible part of it.
        hipGraph_t graph = nullptr;
        hipGraphExec_t gexec = nullptr;
        bool first = true, graph_ok = !(tf & TL_FLAG_LK_NO_GRAPH);
        int rc_loop = TL_OK;
        for (;;) {
            if (!first && graph_ok && !gexec) {
                graph_ok = hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal) == hipSuccess;
                if (graph_ok) {
                    hipError_t le = hipSuccess;
                    for (int r = 0; r < 64 && le == hipSuccess; ++r) le = lk_round((uint32_t)r);
                    const hipError_t ce = hipStreamEndCapture(c->stream, &graph);
                    graph_ok = le == hipSuccess && ce == hipSuccess && graph &&
                               hipGraphInstantiate(&gexec, graph, nullptr, nullptr, 0) == hipSuccess;
                }
                if (!graph_ok) {
                    // separately enqueued launches from here on — after making sure the stream has left capture mode (a capture
                    // that another thread's legacy-stream operation invalidated stays "active, invalidated" until it is ended)
                    (void)hipGetLastError();
                    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
                    if (hipStreamIsCapturing(c->stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) {
                        hipGraph_t g2 = nullptr;
                        (void)hipStreamEndCapture(c->stream, &g2);
                        if (g2) (void)hipGraphDestroy(g2);
                    }
                    (void)hipGetLastError();
                    if (graph) {
                        (void)hipGraphDestroy(graph);
                        graph = nullptr;
                    }
                }
            }
            hipError_t e = hipSuccess;
            if (gexec) e = hipGraphLaunch(gexec, c->stream);
            else for (int r = 0; r < 64 && e == hipSuccess; ++r) e = lk_round((uint32_t)r);
            if (e == hipSuccess) e = hipMemcpyAsync(&hs, G.state, sizeof(hs), hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
            if (e != hipSuccess) {
                rc_loop = fail(c, TL_ERR_HIP, "tl_lk: %s", hipGetErrorString(e));
                break;
            }
            first = false;
            if (live) {  // lin_kernighan.rs:71,90 send_progress(best_tour, best_dist), as the search goes
                for (; delivered < hs.snaps; ++delivered) {
                    const uint32_t at = delivered % snap_cap;
                    float bd = 0.0f;
                    e = hipMemcpyAsync(live_tour.data(), G.snap + (size_t)at * n, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream);
                    if (e == hipSuccess) e = hipMemcpyAsync(&bd, G.snap_dist + at, 4, hipMemcpyDeviceToHost, c->stream);
                    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
                    if (e != hipSuccess) break;
                    c->in_callback = true;   // an entry into THIS context from inside the callback gets TL_ERR_BUSY (CtxUse): the search owns it
                    live(live_user, live_tour.data(), n, bd);
                    c->in_callback = false;
                }
                if (e != hipSuccess) {
                    rc_loop = fail(c, TL_ERR_HIP, "tl_lk_live: %s", hipGetErrorString(e));
                    break;
                }
            }
            if (hs.finished) break;
        }
        if (gexec) (void)hipGraphExecDestroy(gexec);
        if (graph) (void)hipGraphDestroy(graph);
        if (rc_loop != TL_OK) return rc_loop;
        if (hs.finished == 2u)
            return fail(c, TL_ERR_NO_CONVERGE, "tl_lk: an lk_pass does not terminate (the reference's loop cycles on this input: chains of rounded f32 gain that lead back to the same tour)");
        cnt[0] = hs.scans;
        cnt[1] = hs.searches;
        cnt[2] = hs.moves;
        cnt[3] = hs.exchanged;
        if (snap_pos && !live) {
            if (snap_len) *snap_len = hs.snaps;
            const uint32_t have = hs.snaps < snap_cap ? hs.snaps : snap_cap;
            if (have) {
                HIPCHK(c, hipMemcpyAsync(snap_pos, G.snap, (size_t)have * n * 4, hipMemcpyDeviceToHost, c->stream));
                if (snap_dist) HIPCHK(c, hipMemcpyAsync(snap_dist, G.snap_dist, (size_t)have * 4, hipMemcpyDeviceToHost, c->stream));
            }
        }
    }
    // lin_kernighan.rs:99 Solution::new -> total through problem.distances.tour_length (closing edge first)
    HIPCHK(c, launch_tour_length(ddm ? nullptr : G.xy, ddm, n, G.best, (float *)c->out_cost.p, c->stream));
    HIPCHK(c, hipEventRecord(c->ev1, c->stream));
    c->ev_valid = true;
    float cost = 0.f;
    HIPCHK(c, hipMemcpyAsync(out_pos, G.best, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    if (!multi_cu && !lk_ils) HIPCHK(c, hipMemcpyAsync(cnt, G.counters, 32, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(&cost, c->out_cost.p, 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (out_cost) *out_cost = cost;
    if (snap_pos && !multi_cu && !lk_ils) {
        // TL_FLAG_LK_ONE_WORKGROUP: the reference's last PathUpdate only — the final best tour with its best_dist, the Euclidean
        // tour_distance of lin_kernighan.rs:118-122 (edges in tour order, the closing edge last; f32, as KDPoint::distance)
        float bd = 0.0f;
        for (uint32_t q = 0; q < n; ++q) {
            const float *p0 = xy + 2 * (size_t)out_pos[q], *p1 = xy + 2 * (size_t)out_pos[(q + 1u) % n];
            const float dx = p0[0] - p1[0], dy = p0[1] - p1[1];
            const float sq = dx * dx + dy * dy;  // -ffp-contract=off: three roundings
            bd += sqrtf(sq);
        }
        if (live) {
            live(live_user, out_pos, n, bd);
        } else {
            if (snap_len) *snap_len = 1;
            if (snap_cap) {
                memcpy(snap_pos, out_pos, (size_t)n * 4);
                if (snap_dist) snap_dist[0] = bd;
            }
        }
    }
    if (stats) {
        stats->sweeps = cnt[0];
        stats->candidates = cnt[1];
        stats->moves = cnt[2];
        stats->reversed = cnt[3];
        double kms = 0;
        tl_last_kernel_ms(c, &kms);
        stats->kernel_ms = kms;
        stats->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    return TL_OK;
}

extern "C" int tl_lk(tl_ctx *c, const float *xy, uint32_t n, const float *dm_packed, const uint32_t *init_pos, const tl_lk_opts *opts,
                     uint64_t seed, uint32_t *out_pos, float *out_cost, tl_stats *stats)
{
    TL_ENTER(c);
    return lk_run(c, xy, n, dm_packed, init_pos, opts, seed, out_pos, out_cost, stats, nullptr, nullptr, 0, nullptr);
}

// lin_kernighan::solve with the best tours it passes through listed (the reference's progress side channel: one
// PathUpdate(best_tour, best_dist) after the first lk_pass and one per improving epoch, lin_kernighan.rs:71,90) — the device-side
// state machine copies each into the caller's list as it settles on it.
extern "C" int tl_lk_trace(tl_ctx *c, const float *xy, uint32_t n, const float *dm_packed, const uint32_t *init_pos, const tl_lk_opts *opts,
                           uint64_t seed, uint32_t *out_pos, float *out_cost, tl_stats *stats, uint32_t *snap_pos, float *snap_dist,
                           uint32_t snap_cap, uint32_t *snap_len)
{
    TL_ENTER(c);
    if (!snap_pos || !snap_dist || !snap_len) return fail(c, TL_ERR_BADARG, "tl_lk_trace: NULL argument");
    return lk_run(c, xy, n, dm_packed, init_pos, opts, seed, out_pos, out_cost, stats, snap_pos, snap_dist, snap_cap, snap_len);
}

// lin_kernighan::solve with its progress messages sent WHILE it runs (VERDICT r03 "missing 5": the reference's Qt front end watches
// a multi-second LK run through its channel, teeline-qt/src/solver_engine.rs:412-434): `progress` is called on the calling thread,
// between two polls of the device-side search, once for every best tour the ILS settles on, in order — the reference's
// PathUpdate(best_tour, best_dist) of lin_kernighan.rs:71,90.  Same tours and distances as tl_lk_trace lists after the fact.
extern "C" int tl_lk_live(tl_ctx *c, const float *xy, uint32_t n, const float *dm_packed, const uint32_t *init_pos, const tl_lk_opts *opts,
                          uint64_t seed, uint32_t *out_pos, float *out_cost, tl_stats *stats, tl_lk_progress_fn progress, void *user)
{
    TL_ENTER(c);
    if (!progress) return fail(c, TL_ERR_BADARG, "tl_lk_live: progress is NULL");
    return lk_run(c, xy, n, dm_packed, init_pos, opts, seed, out_pos, out_cost, stats, nullptr, nullptr, 0, nullptr, progress, user);
}

