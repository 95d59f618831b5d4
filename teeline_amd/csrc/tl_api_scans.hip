// tl_api_scans.hip — C ABI, the best-improvement scans: tl_three_opt* (src/tsp/three_opt.rs:16-218) and tl_or_opt* (src/tsp/or_opt.rs:18-184).
#include "tl_api_common.h"

using namespace tl;
using namespace tlapi;

static constexpr int kScanBatch = 16;  // passes of a 3-opt / Or-opt descent enqueued per host poll (later ones return at once when it ends)

// ------------------------------------------------------------------------------------------------
// 3-opt
// ------------------------------------------------------------------------------------------------
struct ThreeOptSetup {
    ThreeOptArgs A{};
    uint32_t nblocks = 0;
    bool dm = false;
};

static uint32_t three_opt_max_n(const tl_ctx *)
{
    // (i, j) and (k, case) travel as packed 16-bit fields; the workspace holds an n x (n+1) f32 matrix (17 GB at this limit: sized
    // for 288 GB of HBM).  (Round 4: k_three_opt_pick stages the move's segments in the workspace where they do not fit the LDS —
    // the limit was ~40 K before.)
    return 65535u;
}

// uploads inputs, lays out the workspace in c->work and fills the kernel argument block
static int three_opt_setup(tl_ctx *c, const float *xy, uint32_t n, const float *dm_packed, const uint32_t *path, ThreeOptSetup &S)
{
    if (n > three_opt_max_n(c))
        return fail(c, TL_ERR_UNSUPPORTED, "three_opt: n=%u exceeds the limit %u of this build (packed 16-bit indices)", n, three_opt_max_n(c));
    int rc;
    S.dm = dm_packed != nullptr;
    const uint32_t jc = n <= 256 ? 4u : 16u;
    std::vector<uint32_t> prefix(n - 1);
    uint32_t acc = 0;
    for (uint32_t i = 0; i + 2 < n; ++i) {
        prefix[i] = acc;
        acc += ((n - 2u - i) + jc - 1u) / jc;  // j in [i+1, n-1)
    }
    prefix[n - 2] = acc;
    S.nblocks = acc;
    if (S.dm) {
        const size_t b = (size_t)n * (n - 1) / 2 * 4;
        if ((rc = ensure(c, c->dm, b))) return rc;
        HIPCHK(c, hipMemcpyAsync(c->dm.p, dm_packed, b, hipMemcpyHostToDevice, c->stream));
    } else {
        if ((rc = ensure(c, c->xy, (size_t)n * 8))) return rc;
        HIPCHK(c, hipMemcpyAsync(c->xy.p, xy, (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
    }
    // workspace: perm | Pt | E | prefix | partials | best | counters
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t o_perm = 0, o_pt = up(o_perm + (size_t)n * 4), o_e = up(o_pt + (size_t)(n + 1) * 8), o_pre = up(o_e + (size_t)n * 4),
                 o_par = up(o_pre + (size_t)(n - 1) * 4), o_best = up(o_par + (size_t)S.nblocks * sizeof(ThreeOptBest)),
                 o_cnt = up(o_best + sizeof(ThreeOptBest)), o_scr = up(o_cnt + 16), o_dt = up(o_scr + (size_t)n * 4), total = up(o_dt + (size_t)n * (n + 1) * 4);
    if ((rc = ensure(c, c->work, total))) return rc;
    unsigned char *w = (unsigned char *)c->work.p;
    S.A.Dt = (float *)(w + o_dt);
    std::vector<uint32_t> ident;
    if (!path) {
        ident.resize(n);
        for (uint32_t i = 0; i < n; ++i) ident[i] = i;
        path = ident.data();
    }
    HIPCHK(c, hipMemcpyAsync(w + o_perm, path, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(w + o_pre, prefix.data(), (size_t)(n - 1) * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemsetAsync(w + o_cnt, 0, 16, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));  // host vectors above go out of scope
    S.A.xy = (const float2 *)c->xy.p;
    S.A.dm = S.dm ? (const float *)c->dm.p : nullptr;
    S.A.perm = (uint32_t *)(w + o_perm);
    S.A.Pt = (float2 *)(w + o_pt);
    S.A.E = (float *)(w + o_e);
    S.A.chunk_prefix = (const uint32_t *)(w + o_pre);
    S.A.partials = (ThreeOptBest *)(w + o_par);
    S.A.best = (ThreeOptBest *)(w + o_best);
    S.A.counters = (uint64_t *)(w + o_cnt);
    S.A.scratch = (uint32_t *)(w + o_scr);
    S.A.n = n;
    S.A.jc = jc;
    return TL_OK;
}

extern "C" int tl_three_opt_find_best_move(tl_ctx *c, const float *xy, uint32_t n, const float *dm_packed, const uint32_t *path,
                                           int *found, uint32_t *oi, uint32_t *oj, uint32_t *ok, int *kase, float *savings)
{
    TL_ENTER(c);
    if (!c || (!xy && !dm_packed) || !path || !found) return fail(c, TL_ERR_BADARG, "tl_three_opt_find_best_move: NULL argument");
    *found = 0;
    if (n < 4) return TL_OK;
    if (!is_permutation(path, n)) return fail(c, TL_ERR_BADARG, "tl_three_opt_find_best_move: path is not a permutation of 0..n-1");
    HIPCHK(c, hipSetDevice(c->device));
    ThreeOptSetup S;
    int rc;
    if ((rc = three_opt_setup(c, xy, n, dm_packed, path, S))) return rc;
    HIPCHK(c, hipEventRecord(c->ev0, c->stream));
    HIPCHK(c, launch_three_opt_pass(S.A, S.nblocks, S.dm, 0, c->stream, c->lds_bytes));
    HIPCHK(c, hipEventRecord(c->ev1, c->stream));
    c->ev_valid = true;
    ThreeOptBest b{};
    HIPCHK(c, hipMemcpyAsync(&b, S.A.best, sizeof(b), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (b.found) {
        *found = 1;
        if (oi) *oi = b.ij >> 16;
        if (oj) *oj = b.ij & 0xFFFFu;
        if (ok) *ok = b.kc >> 3;
        if (kase) *kase = (int)(b.kc & 7u);
        if (savings) *savings = b.sav;
    }
    return TL_OK;
}

// move_log (optional): 4 words per applied move — i, j, k, case of three_opt.rs:36-45 in order — at most log_cap moves; *log_len = moves
static int three_opt_run(tl_ctx *c, const float *xy, uint32_t n, const float *dm_packed, const uint32_t *init_pos,
                         uint32_t *out_pos, float *out_cost, tl_stats *stats, uint32_t *move_log, uint32_t log_cap, uint32_t *log_len)
{
    if (log_len) *log_len = 0;
    if (!c || (!xy && !dm_packed) || !out_pos) return fail(c, TL_ERR_BADARG, "tl_three_opt: NULL argument");
    const auto t0 = std::chrono::steady_clock::now();
    if (stats) memset(stats, 0, sizeof(*stats));
    if (n < 4) {  // three_opt.rs:25-28: returns the cities order, init_tour ignored
        for (uint32_t i = 0; i < n; ++i) out_pos[i] = i;
        if (out_cost) {
            if (n < 2) *out_cost = 0.0f;
            else {
                int rc = tl_tour_length(c, xy, dm_packed, n, out_pos, out_cost);
                if (rc) return rc;
            }
        }
        return TL_OK;
    }
    if (init_pos && !is_permutation(init_pos, n)) return fail(c, TL_ERR_BADARG, "tl_three_opt: init tour is not a permutation of 0..n-1");
    HIPCHK(c, hipSetDevice(c->device));
    ThreeOptSetup S;
    int rc;
    if ((rc = three_opt_setup(c, xy, n, dm_packed, init_pos, S))) return rc;
    // The descent runs as batches of passes enqueued back to back (tl_kernels.h ScanRunState): the pick kernel counts passes and
    // moves, files every move (i, j, k, case) and sets `done` when a pass finds none; the host looks once per batch.
    const uint32_t dev_log_cap = move_log ? log_cap : 0u;
    if ((rc = ensure(c, c->misc, 256 + (size_t)(dev_log_cap ? dev_log_cap : 1u) * 16))) return rc;
    S.A.run = (ScanRunState *)c->misc.p;
    S.A.log = (uint32_t *)((unsigned char *)c->misc.p + 256);
    const ScanRunState hs0{0u, 0u, 0u, dev_log_cap};
    ScanRunState hs = hs0;
    HIPCHK(c, hipMemcpyAsync(S.A.run, &hs0, sizeof(hs0), hipMemcpyHostToDevice, c->stream));
    c->ev_valid = false;  // (an early return below must not leave this ev0 paired with an older sequence's ev1: ADVICE r04)
    HIPCHK(c, hipEventRecord(c->ev0, c->stream));
    const uint64_t cap = 64ull * n + 1024;  // safety cap, far above any observed pass count
    for (;;) {
        for (int b = 0; b < kScanBatch; ++b) HIPCHK(c, launch_three_opt_pass(S.A, S.nblocks, S.dm, 1, c->stream, c->lds_bytes));
        HIPCHK(c, hipMemcpyAsync(&hs, S.A.run, sizeof(hs), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (hs.done) break;  // three_opt.rs:36-45
        if (hs.passes > cap) return fail(c, TL_ERR_NO_CONVERGE, "three_opt: pass cap reached");
    }
    const uint64_t passes = hs.passes, moves = hs.moves;
    HIPCHK(c, hipEventRecord(c->ev1, c->stream));
    c->ev_valid = true;
    HIPCHK(c, hipMemcpyAsync(out_pos, S.A.perm, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    if (move_log && moves && dev_log_cap)
        HIPCHK(c, hipMemcpyAsync(move_log, S.A.log, (size_t)(moves < dev_log_cap ? moves : dev_log_cap) * 16, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (out_cost) {
        // Solution::from_parts -> tour_length (mod.rs:1776-1789)
        if ((rc = ensure(c, c->out_cost, 4))) return rc;
        HIPCHK(c, launch_tour_length(S.dm ? nullptr : S.A.xy, S.A.dm, n, S.A.perm, (float *)c->out_cost.p, c->stream));
        HIPCHK(c, hipMemcpyAsync(out_cost, c->out_cost.p, 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    if (stats) {
        stats->sweeps = passes;
        stats->moves = moves;
        const uint64_t nn = n;
        stats->candidates = passes * (nn * (nn - 1) * (nn - 2) / 6 - (nn - 2));  // C(n,3) - (n-2) triples per pass
        double kms = 0;
        tl_last_kernel_ms(c, &kms);
        stats->kernel_ms = kms;
        stats->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    if (log_len) *log_len = (uint32_t)moves;
    return TL_OK;
}

extern "C" int tl_three_opt(tl_ctx *c, const float *xy, uint32_t n, const float *dm_packed, const uint32_t *init_pos,
                            uint32_t *out_pos, float *out_cost, tl_stats *stats)
{
    TL_ENTER(c);
    return three_opt_run(c, xy, n, dm_packed, init_pos, out_pos, out_cost, stats, nullptr, 0, nullptr);
}

// three_opt::solve with its moves listed: the reference sends the path after every apply_3opt (three_opt.rs:34,42,47-49); the
// pick kernel files every move it applies, the list is read back once.
extern "C" int tl_three_opt_trace(tl_ctx *c, const float *xy, uint32_t n, const float *dm_packed, const uint32_t *init_pos,
                                  uint32_t *out_pos, float *out_cost, tl_stats *stats, uint32_t *move_log, uint32_t log_cap, uint32_t *log_len)
{
    TL_ENTER(c);
    if (!move_log || !log_len) return fail(c, TL_ERR_BADARG, "tl_three_opt_trace: NULL argument");
    return three_opt_run(c, xy, n, dm_packed, init_pos, out_pos, out_cost, stats, move_log, log_cap, log_len);
}

// ------------------------------------------------------------------------------------------------
// Or-opt (or_opt.rs)
// ------------------------------------------------------------------------------------------------
static int or_opt_setup(tl_ctx *c, const float *xy, uint32_t n, const float *dm_packed, const uint32_t *path, OrOptArgs &A, bool &dm)
{
    int rc;  // (round 4: a 96-bit argmin key and a workspace copy of the tour beyond the LDS — no size limit of its own any more)
    dm = dm_packed != nullptr;
    if (dm) {
        const size_t b = (size_t)n * (n - 1) / 2 * 4;
        if ((rc = ensure(c, c->dm, b))) return rc;
        HIPCHK(c, hipMemcpyAsync(c->dm.p, dm_packed, b, hipMemcpyHostToDevice, c->stream));
    } else {
        if ((rc = ensure(c, c->xy, (size_t)n * 8))) return rc;
        HIPCHK(c, hipMemcpyAsync(c->xy.p, xy, (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
    }
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const uint32_t nblocks = or_opt_scan_blocks(n);
    const size_t o_perm = 0, o_pt = up((size_t)n * 4), o_e = up(o_pt + (size_t)n * 8), o_par = up(o_e + (size_t)n * 4),
                 o_best = up(o_par + (size_t)nblocks * 16), o_old = up(o_best + 256), total = o_old + (size_t)n * 4 + 256;
    if ((rc = ensure(c, c->work, total))) return rc;
    unsigned char *w = (unsigned char *)c->work.p;
    std::vector<uint32_t> ident;
    if (!path) {
        ident.resize(n);
        for (uint32_t i = 0; i < n; ++i) ident[i] = i;
        path = ident.data();
    }
    HIPCHK(c, hipMemcpyAsync(w + o_perm, path, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    A.xy = (const float2 *)c->xy.p;
    A.dm = dm ? (const float *)c->dm.p : nullptr;
    A.perm = (uint32_t *)(w + o_perm);
    A.Pt = (float2 *)(w + o_pt);
    A.E = (float *)(w + o_e);
    A.partials = (unsigned long long *)(w + o_par);
    A.best = (OrOptBest *)(w + o_best);
    A.scratch = (uint32_t *)(w + o_old);
    A.n = n;
    return TL_OK;
}

extern "C" int tl_or_opt_find_best_move(tl_ctx *c, const float *xy, uint32_t n, const float *dm_packed, const uint32_t *path,
                                        int *found, float *delta, uint32_t *oi, uint32_t *oj, uint32_t *seg_len, int *reversed)
{
    TL_ENTER(c);
    if (!c || (!xy && !dm_packed) || !path || !found) return fail(c, TL_ERR_BADARG, "tl_or_opt_find_best_move: NULL argument");
    *found = 0;
    if (n < 4) return TL_OK;
    if (!is_permutation(path, n)) return fail(c, TL_ERR_BADARG, "tl_or_opt_find_best_move: path is not a permutation of 0..n-1");
    HIPCHK(c, hipSetDevice(c->device));
    OrOptArgs A{};
    bool dm;
    int rc;
    if ((rc = or_opt_setup(c, xy, n, dm_packed, path, A, dm))) return rc;
    HIPCHK(c, hipEventRecord(c->ev0, c->stream));
    HIPCHK(c, launch_or_opt_pass(A, dm, 0, c->stream, c->lds_bytes));
    HIPCHK(c, hipEventRecord(c->ev1, c->stream));
    c->ev_valid = true;
    OrOptBest b{};
    HIPCHK(c, hipMemcpyAsync(&b, A.best, sizeof(b), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (b.found) {
        *found = 1;
        if (delta) memcpy(delta, &b.delta_bits, 4);
        if (oi) *oi = b.i;
        if (oj) *oj = b.j;
        if (seg_len) *seg_len = b.seg_len;
        if (reversed) *reversed = (int)b.reversed;
    }
    return TL_OK;
}

// move_log (optional): 4 words per applied move — i, j, seg_len, reversed of or_opt.rs:45-51 in order — at most log_cap moves; *log_len = moves
static int or_opt_run(tl_ctx *c, const float *xy, uint32_t n, const float *dm_packed, const uint32_t *init_pos,
                      uint32_t *out_pos, float *out_cost, tl_stats *stats, uint32_t *move_log, uint32_t log_cap, uint32_t *log_len)
{
    if (log_len) *log_len = 0;
    if (!c || (!xy && !dm_packed) || !out_pos) return fail(c, TL_ERR_BADARG, "tl_or_opt: NULL argument");
    const auto t0 = std::chrono::steady_clock::now();
    if (stats) memset(stats, 0, sizeof(*stats));
    if (n < 4) {  // or_opt.rs:31-34: returns the cities order
        for (uint32_t i = 0; i < n; ++i) out_pos[i] = i;
        if (out_cost) {
            if (n < 2) *out_cost = 0.0f;
            else {
                int rc = tl_tour_length(c, xy, dm_packed, n, out_pos, out_cost);
                if (rc) return rc;
            }
        }
        return TL_OK;
    }
    if (init_pos && !is_permutation(init_pos, n)) return fail(c, TL_ERR_BADARG, "tl_or_opt: init tour is not a permutation of 0..n-1");
    HIPCHK(c, hipSetDevice(c->device));
    OrOptArgs A{};
    bool dm;
    int rc;
    if ((rc = or_opt_setup(c, xy, n, dm_packed, init_pos, A, dm))) return rc;
    // batches of passes enqueued back to back; k_or_pick counts, files every move (i, j, seg_len, reversed) and ends the descent
    const uint32_t dev_log_cap = move_log ? log_cap : 0u;
    if ((rc = ensure(c, c->misc, 256 + (size_t)(dev_log_cap ? dev_log_cap : 1u) * 16))) return rc;
    A.run = (ScanRunState *)c->misc.p;
    A.log = (uint32_t *)((unsigned char *)c->misc.p + 256);
    const ScanRunState hs0{0u, 0u, 0u, dev_log_cap};
    ScanRunState hs = hs0;
    HIPCHK(c, hipMemcpyAsync(A.run, &hs0, sizeof(hs0), hipMemcpyHostToDevice, c->stream));
    c->ev_valid = false;
    HIPCHK(c, hipEventRecord(c->ev0, c->stream));
    const uint64_t cap = 64ull * n + 1024;
    for (;;) {  // or_opt.rs:45 while let Some(best) = find_best_move(..)
        for (int b = 0; b < kScanBatch; ++b) HIPCHK(c, launch_or_opt_pass(A, dm, 1, c->stream, c->lds_bytes));
        HIPCHK(c, hipMemcpyAsync(&hs, A.run, sizeof(hs), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (hs.done) break;
        if (hs.passes > cap) return fail(c, TL_ERR_NO_CONVERGE, "or_opt: pass cap reached");
    }
    const uint64_t passes = hs.passes, moves = hs.moves;
    HIPCHK(c, hipEventRecord(c->ev1, c->stream));
    c->ev_valid = true;
    HIPCHK(c, hipMemcpyAsync(out_pos, A.perm, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    if (move_log && moves && dev_log_cap)
        HIPCHK(c, hipMemcpyAsync(move_log, A.log, (size_t)(moves < dev_log_cap ? moves : dev_log_cap) * 16, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (out_cost) {
        if ((rc = ensure(c, c->out_cost, 4))) return rc;
        HIPCHK(c, launch_tour_length(dm ? nullptr : A.xy, A.dm, n, A.perm, (float *)c->out_cost.p, c->stream));
        HIPCHK(c, hipMemcpyAsync(out_cost, c->out_cost.p, 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    if (stats) {
        stats->sweeps = passes;
        stats->moves = moves;
        // deltas evaluated per pass: seg_len 1: n(n-2) forward; seg_len 2: (n-1)(n-3) x 2; seg_len 3: (n-2)(n-4) x 2
        const uint64_t nn = n;
        uint64_t per = 0;
        if (nn > 2) per += nn * (nn - 2);
        if (nn > 3) per += 2 * (nn - 1) * (nn - 3);
        if (nn > 4) per += 2 * (nn - 2) * (nn - 4);
        stats->candidates = passes * per;
        double kms = 0;
        tl_last_kernel_ms(c, &kms);
        stats->kernel_ms = kms;
        stats->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    if (log_len) *log_len = (uint32_t)moves;
    return TL_OK;
}

extern "C" int tl_or_opt(tl_ctx *c, const float *xy, uint32_t n, const float *dm_packed, const uint32_t *init_pos,
                         uint32_t *out_pos, float *out_cost, tl_stats *stats)
{
    TL_ENTER(c);
    return or_opt_run(c, xy, n, dm_packed, init_pos, out_pos, out_cost, stats, nullptr, 0, nullptr);
}

// or_opt::solve with its moves listed: the reference sends the path and its tour_length after every apply_relocation
// (or_opt.rs:40-42,62-67,70-72); the pick kernel files every move it applies.
extern "C" int tl_or_opt_trace(tl_ctx *c, const float *xy, uint32_t n, const float *dm_packed, const uint32_t *init_pos,
                               uint32_t *out_pos, float *out_cost, tl_stats *stats, uint32_t *move_log, uint32_t log_cap, uint32_t *log_len)
{
    TL_ENTER(c);
    if (!move_log || !log_len) return fail(c, TL_ERR_BADARG, "tl_or_opt_trace: NULL argument");
    return or_opt_run(c, xy, n, dm_packed, init_pos, out_pos, out_cost, stats, move_log, log_cap, log_len);
}

