// tl_api_two_opt.hip — C ABI, 2-opt family: tl_two_opt (REF_ORDER on coordinates / matrix / beyond the LDS, BEST_SWEEP), its trace,
// batch, multi-start (one and several devices) and population entries.  Replaces two_opt::solve (src/tsp/two_opt.rs:7-67).
#include "tl_api_common.h"

using namespace tl;
using namespace tlapi;

#ifndef TL_NL_MIN_N
#define TL_NL_MIN_N 400u  // smallest instance whose descents build and read the neighbour lists
#endif
#ifndef TL_DM_LISTS_MIN_N
#define TL_DM_LISTS_MIN_N 500u  // matrix form: smallest instance whose descents build and read the lists of the late sweeps (n = 200 / 300: the lists
                                // cost 3-20 % more than they save, n = 532: even, n = 1 002: -33 % from the NN tour; scripts/timing_dm_late.py)
#endif
#ifndef TL_DM_LONG_MAX
#define TL_DM_LONG_MAX 1024u  // ... a sweep runs on them while at most this many cities have a tour edge beyond their 16th distance (the descent's list holds 1024)
#endif
#ifndef TL_DM_MOVES_DIV
#define TL_DM_MOVES_DIV 4000u  // ... and the sweep before it applied at most n^2 / 4000 moves: on the lists a sweep costs ~n rows + a step per move, in the
#endif                         // other block shapes ~n^2 candidates (measured best at n = 1002 / 2000 / 3000: n / 4, n, n; scripts/timing_dm_late_knobs.py)
#ifndef TL_NL_SWEEP_MIN
#define TL_NL_SWEEP_MIN 3  // first sweep of a descent that may run in the late phase (tuning builds override it) ...
#endif
#ifndef TL_NL_MOVES_DIV
#define TL_NL_MOVES_DIV 40u  // ... once a sweep has applied fewer than n / 40 moves (n = 10^4: 250 — random restarts: the sixth sweep; NN start: the fourth)
#endif

// ------------------------------------------------------------------------------------------------
// 2-opt
// ------------------------------------------------------------------------------------------------
static int two_opt_enqueue(tl_ctx *c, const float2 *d_xy, const float *d_dm, uint32_t n, const uint32_t *d_init,
                           uint32_t init_mode, uint64_t seed, uint32_t first, uint32_t count, int mode,
                           uint32_t *d_out_pos, float *d_out_cost, uint64_t *d_out_stats, hipStream_t s,
                           uint32_t *d_move_log = nullptr, uint32_t log_cap = 0)
{
    if (mode != TL_MODE_REF_ORDER) return fail(c, TL_ERR_UNSUPPORTED, "batch 2-opt supports TL_MODE_REF_ORDER only");
    if (n < 3) return fail(c, TL_ERR_REF_PANICS, "two_opt: n=%u < 3 — the reference underflows `n_indices - 2` (two_opt.rs:17,29)", n);
    if (count == 0) return TL_OK;
    TwoOptBatchArgs A{};
    A.xy = d_xy;
    A.dm = d_dm;
    A.init = d_init;
    A.out_pos = d_out_pos;
    A.out_cost = d_out_cost;
    A.out_stats = d_out_stats;
    A.seed = seed;
    A.first = first;
    A.n = n;
    A.max_sweeps = TL_MAX_SWEEPS;
    A.init_mode = init_mode;
    A.move_log = d_move_log;
    A.log_cap = log_cap;
    // every size / mode check comes before the first event record: a rejected call must leave the event pair of the
    // previous kernel sequence intact
    if (d_dm) {
        if (init_mode == TL_INIT_SEEDED) return fail(c, TL_ERR_UNSUPPORTED, "seeded restarts need coordinates (dm_packed must be NULL)");
        if (two_opt_ref_dm_lds_bytes(n) > (size_t)c->lds_bytes || n > 65535)
            return fail(c, TL_ERR_UNSUPPORTED, "two_opt (matrix form): n=%u exceeds the LDS tour capacity", n);
    } else if (n > lds_max_n(c->lds_bytes)) {
        return fail(c, TL_ERR_UNSUPPORTED, "two_opt (on-the-fly form): n=%u exceeds the LDS-resident limit %u", n, lds_max_n(c->lds_bytes));
    }
    c->ev_valid = false;
    {   // nl / fx / dmfull may still be read by an asynchronous batch this context enqueued on another stream
        int rco;
        if ((rco = ws_order(c, s))) return rco;
    }
    HIPCHK(c, hipEventRecord(c->ev0, s));
    if (d_dm) {
        // the packed triangle (reference layout) is expanded to a full row-major matrix once per call: a row scan then
        // gathers inside one 4n-byte row instead of one cache line per column (two_opt_dm.hip)
        int rc2;
        // ... and, where the descents' late sweeps can use them, the lists cut from it (two_opt_dm.hip: the 16 nearest per city and the
        // reverse relation, one pass of a wave per matrix row), behind the matrix in the same buffer
        // (the descents' per-city records are 264 bytes per city and descent: a batch that would need more than 8 GB of them runs without lists)
        const bool lists = !(c->flags & (TL_FLAG_NO_PRUNE | TL_FLAG_2OPT_NO_NL)) && (n >= TL_DM_LISTS_MIN_N || (c->flags & TL_FLAG_2OPT_NL_ALWAYS)) &&
                           two_opt_ref_dm_late_fits(n, c->lds_bytes) && two_opt_ref_dm_late_work_bytes(n, count) <= ((size_t)8 << 30);
        const size_t full_bytes = ((size_t)n * n * 4 + 255u) & ~(size_t)255u;
        if ((rc2 = ensure(c, c->dmfull, full_bytes + (lists ? dm_lists_ws_bytes(n) : 0u)))) return rc2;
        HIPCHK(c, launch_dm_expand_full(d_dm, n, (float *)c->dmfull.p, s));
        A.dm_full = (const float *)c->dmfull.p;
        if (lists) {
            if ((rc2 = ensure(c, c->dmx, two_opt_ref_dm_late_work_bytes(n, count)))) return rc2;  // (not c->work: tl_two_opt_trace keeps the move log there)
            A.work = (uint32_t *)c->dmx.p;
            HIPCHK(c, launch_dm_lists_build(A.dm_full, n, (unsigned char *)c->dmfull.p + full_bytes, &A.dml, s));
            A.dml.long_max = (c->flags & TL_FLAG_2OPT_NL_ALWAYS) ? 1024u : (uint32_t)TL_DM_LONG_MAX;
            A.dml.moves_max = (c->flags & TL_FLAG_2OPT_NL_ALWAYS) ? 0xFFFFFFFFu : (uint32_t)((uint64_t)n * n / TL_DM_MOVES_DIV > 8u ? (uint64_t)n * n / TL_DM_MOVES_DIV : 8u);
#ifdef TL_TUNE  // tuning builds only: the product library never reads the environment
            if (const char *e = getenv("TL_DM_LONG_MAX")) A.dml.long_max = (uint32_t)atoi(e);
            if (const char *e = getenv("TL_DM_MOVES_MAX")) A.dml.moves_max = (uint32_t)atoi(e);
#endif
        }
        HIPCHK(c, launch_two_opt_ref_dm(A, count, c->lds_bytes, s));
    } else {
        const int force_nt = (c->flags & TL_FLAG_2OPT_NT256) ? 256 : (c->flags & TL_FLAG_2OPT_NT512) ? 512 : 0;
        // Grid-coordinate form: where two tours fit the LDS at 7 B per city but not at 10 (n = 10^4), a batch with more
        // descents than CUs runs two per CU — if the instance lies on a decimal grid 1/S whose decode reproduces every
        // coordinate bit for bit (checked here, on the device, with the kernel's own decode; one 4-byte read-back).
        if (!force_nt && n <= 10240u && ((c->flags & TL_FLAG_2OPT_FX) || two_opt_ref_fx_pays(n, count, c->cus, c->lds_bytes)) &&
            2 * two_opt_ref_fx_lds_bytes(n) <= (size_t)c->lds_bytes) {
            int rc3;
            if ((rc3 = ensure(c, c->fx, (size_t)n * 8 + 16))) return rc3;
            uint2 *g = (uint2 *)c->fx.p;
            uint32_t *bad = (uint32_t *)((unsigned char *)c->fx.p + (size_t)n * 8);
            static const double scales[] = {1.0, 10.0, 100.0, 1000.0, 10000.0};
            for (double sc : scales) {
                uint32_t hbad = 1;
                HIPCHK(c, hipMemsetAsync(bad, 0, 4, s));
                HIPCHK(c, launch_fx_encode(d_xy, n, sc, g, bad, s));
                HIPCHK(c, hipMemcpyAsync(&hbad, bad, 4, hipMemcpyDeviceToHost, s));
                HIPCHK(c, hipStreamSynchronize(s));
                if (hbad == 0) {
                    A.fx_xy = g;
                    A.fx_inv = 1.0 / sc;
                    break;
                }
            }
        }
        // Neighbour lists of the instance for the late sweeps (two_opt_nl.hip): built once per call, read by every descent of the
        // batch — where the 16-wave float2 form runs, the lists' state fits the LDS beside the tour, and the instance is large
        // enough for the lists (~0.3 ms at n = 10^4) to pay.
        if (!A.fx_xy && !(c->flags & (TL_FLAG_NO_PRUNE | TL_FLAG_2OPT_NO_NL)) && (n >= TL_NL_MIN_N || (c->flags & TL_FLAG_2OPT_NL_ALWAYS)) &&
            two_opt_ref_nl_applies(n, count, c->cus, c->lds_bytes, force_nt)) {
            int rc4;
            const void *had = c->nl.p;
            const size_t had_cap = c->nl.cap;
            if ((rc4 = ensure(c, c->nl, two_opt_nl_ws_bytes(n)))) return rc4;
            // (built once per instance: a later call with the same coordinates — compared on the device — finds them in place)
            HIPCHK(c, launch_two_opt_nl_build(d_xy, n, c->nl.p, had != c->nl.p || had_cap != c->nl.cap, &A.nl, s));
            A.nl.sweep_min = (c->flags & TL_FLAG_2OPT_NL_ALWAYS) ? 2u : (uint32_t)TL_NL_SWEEP_MIN;
            A.nl.moves_max = (c->flags & TL_FLAG_2OPT_NL_ALWAYS) ? 0xFFFFFFFFu : (n / (uint32_t)TL_NL_MOVES_DIV > 8u ? n / (uint32_t)TL_NL_MOVES_DIV : 8u);
        }
        HIPCHK(c, launch_two_opt_ref_lds(A, count, !(c->flags & TL_FLAG_NO_PRUNE), s, (c->flags & TL_FLAG_COUNT_WORK) != 0, c->cus, c->lds_bytes, force_nt));
    }
    HIPCHK(c, hipEventRecord(c->ev1, s));
    c->ev_valid = true;
    return ws_mark(c, s);
}

extern "C" int tl_two_opt_last_counters(tl_ctx *c, uint64_t out[16])
{
    TL_ENTER(c);
    if (!c || !out) return fail(c, TL_ERR_BADARG, "tl_two_opt_last_counters: NULL argument");
    static_assert(TL_STATS_STRIDE == 16, "the header documents 16 words");
    if (!c->out_stats.p || c->out_stats.cap < (size_t)TL_STATS_STRIDE * 8) return fail(c, TL_ERR_BADARG, "tl_two_opt_last_counters: no 2-opt call on this context yet");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(out, c->out_stats.p, (size_t)TL_STATS_STRIDE * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return TL_OK;
}

extern "C" int tl_two_opt_plan(uint32_t n, uint32_t count, int cus, int lds_bytes, uint32_t flags, int *threads, int *late_phase)
{
    if (!threads || !late_phase || cus <= 0 || lds_bytes <= 0) return TL_ERR_BADARG;
    *threads = 0;
    *late_phase = 0;
    if (n < 3 || n > lds_max_n(lds_bytes)) return TL_OK;
    const int force_nt = (flags & TL_FLAG_2OPT_NT256) ? 256 : (flags & TL_FLAG_2OPT_NT512) ? 512 : 0;
    *threads = two_opt_ref_pick_nt(n, count, cus, lds_bytes, force_nt);
    const bool wanted = !(flags & (TL_FLAG_NO_PRUNE | TL_FLAG_2OPT_NO_NL)) && (n >= TL_NL_MIN_N || (flags & TL_FLAG_2OPT_NL_ALWAYS));
    *late_phase = wanted && two_opt_ref_nl_form(n, count, cus, lds_bytes, force_nt) ? 1 : 0;
    return TL_OK;
}

extern "C" int tl_two_opt_neighbour_lists(tl_ctx *c, const float *xy, uint32_t n, int form, uint16_t *rec, uint32_t *dkb2, uint16_t *knn_b, uint32_t *rcnt,
                                          uint32_t *ka, uint32_t *kb, uint32_t *rb)
{
    TL_ENTER(c);
    if (!c || !xy || !rec || !dkb2 || !knn_b || !rcnt) return fail(c, TL_ERR_BADARG, "tl_two_opt_neighbour_lists: NULL argument");
    if (n <= (uint32_t)kNlKB + 1u || n > 65535u) return fail(c, TL_ERR_UNSUPPORTED, "tl_two_opt_neighbour_lists: n=%u outside (%d, 65535]", n, kNlKB + 1);
    HIPCHK(c, hipSetDevice(c->device));
    int rc;
    if ((rc = ensure(c, c->xy, (size_t)n * 8)) || (rc = ensure(c, c->nl, two_opt_nl_ws_bytes(n))) || (rc = ws_order(c, c->stream))) return rc;
    HIPCHK(c, hipMemcpyAsync(c->xy.p, xy, (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
    TwoOptNl L{};
    HIPCHK(c, hipMemsetAsync(c->nl.p, 0, 256, c->stream));  // (always a fresh build here)
    HIPCHK(c, launch_two_opt_nl_build((const float2 *)c->xy.p, n, c->nl.p, true, &L, c->stream, form == 1 ? 1 : 0));
    HIPCHK(c, hipMemcpyAsync(rec, L.rec, (size_t)n * 128, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(dkb2, L.dkb2, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(knn_b, L.knn_b, (size_t)n * kNlKB * 2, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(rcnt, L.rcnt, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (ka) *ka = kNlKA;
    if (kb) *kb = kNlKB;
    if (rb) *rb = kNlRB;
    return TL_OK;
}

extern "C" int tl_two_opt_batch_dev(tl_ctx *c, const float *d_xy, uint32_t n, const uint32_t *d_init, uint64_t seed,
                                    uint32_t first, uint32_t count, int mode, uint32_t *d_out_pos, float *d_out_cost,
                                    uint64_t *d_out_stats, void *stream)
{
    TL_ENTER(c);
    if (!c || !d_xy || !d_out_pos || !d_out_cost || !d_out_stats) return fail(c, TL_ERR_BADARG, "tl_two_opt_batch_dev: NULL argument");
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    return two_opt_enqueue(c, (const float2 *)d_xy, nullptr, n, d_init, d_init ? TL_INIT_ARRAY : TL_INIT_SEEDED, seed, first,
                           count, mode, d_out_pos, d_out_cost, d_out_stats, s);
}

static void fill_stats(tl_stats *st, uint32_t n, const uint64_t *raw, uint32_t count, double kernel_ms, double total_ms)
{
    if (!st) return;
    memset(st, 0, sizeof(*st));
    const uint64_t per_sweep = n >= 4 ? (uint64_t)(n - 3) * (n - 2) / 2 : 0;
    for (uint32_t r = 0; r < count; ++r) {
        st->sweeps += raw[TL_STATS_STRIDE * r + 0];
        st->moves += raw[TL_STATS_STRIDE * r + 1];
        st->reversed += raw[TL_STATS_STRIDE * r + 2];
    }
    st->candidates = st->sweeps * per_sweep;
    st->kernel_ms = kernel_ms;
    st->total_ms = total_ms;
}

static int two_opt_best_sweep(tl_ctx *c, const float *xy, uint32_t n, const float *dm_packed, const uint32_t *init_pos,
                              uint32_t *out_pos, float *out_cost, tl_stats *stats);

// REF_ORDER for n beyond the LDS-resident kernel: tour state in HBM, scan spread over the chip (two_opt_large.hip)
static int two_opt_ref_large(tl_ctx *c, const float *xy, uint32_t n, const uint32_t *init_pos, uint32_t *out_pos, float *out_cost,
                             tl_stats *stats)
{
    const auto t0 = std::chrono::steady_clock::now();
    HIPCHK(c, hipSetDevice(c->device));
    int rc;
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const uint32_t n_pad = ((n + 64u + 63u) / 64u) * 64u, ntile_cap = (((n_pad >> 6) + 63u) / 64u) * 64u;
    const size_t o_perm = 0, o_P = up((size_t)n * 4), o_box = up(o_P + (size_t)(n_pad + 1) * 8), o_msq = up(o_box + (size_t)ntile_cap * 16),
                 o_st = up(o_msq + (size_t)ntile_cap * 4), total = o_st + 256;
    if ((rc = ensure(c, c->xy, (size_t)n * 8)) || (rc = ensure(c, c->work, total)) || (rc = ensure(c, c->out_cost, 4))) return rc;
    unsigned char *w = (unsigned char *)c->work.p;
    std::vector<uint32_t> ident;
    if (!init_pos) {
        ident.resize(n);
        for (uint32_t i = 0; i < n; ++i) ident[i] = i;
        init_pos = ident.data();
    }
    HIPCHK(c, hipMemcpyAsync(c->xy.p, xy, (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(w + o_perm, init_pos, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    LargeTwoOptArgs A{};
    A.xy = (const float2 *)c->xy.p;
    A.perm = (uint32_t *)(w + o_perm);
    A.P = (float2 *)(w + o_P);
    A.tbox = (float4 *)(w + o_box);
    A.tmsq = (float *)(w + o_msq);
    A.state = (LargeTwoOptState *)(w + o_st);
    A.n = n;
    A.n_pad = n_pad;
    A.ntile_cap = ntile_cap;
    A.max_sweeps = TL_MAX_SWEEPS;
    HIPCHK(c, hipEventRecord(c->ev0, c->stream));
    HIPCHK(c, launch_large_two_opt_init(A, c->stream));
    LargeTwoOptState hs{};
    for (;;) {
        for (int r = 0; r < 64; ++r) HIPCHK(c, launch_large_two_opt_round(A, c->stream));  // kernels no-op once done
        HIPCHK(c, hipMemcpyAsync(&hs, A.state, sizeof(hs), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (hs.done) break;
    }
    if (hs.status) return fail(c, TL_ERR_NO_CONVERGE, "two_opt: sweep cap reached");
    HIPCHK(c, launch_tour_length(A.xy, nullptr, n, A.perm, (float *)c->out_cost.p, c->stream));
    HIPCHK(c, hipEventRecord(c->ev1, c->stream));
    c->ev_valid = true;
    float cost = 0.f;
    HIPCHK(c, hipMemcpyAsync(out_pos, A.perm, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(&cost, c->out_cost.p, 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (out_cost) *out_cost = cost;
    if (stats) {
        memset(stats, 0, sizeof(*stats));
        stats->sweeps = hs.sweeps;
        stats->moves = hs.moves;
        stats->reversed = hs.reversed;
        stats->candidates = (uint64_t)hs.sweeps * ((uint64_t)(n - 3) * (n - 2) / 2);
        double kms = 0;
        tl_last_kernel_ms(c, &kms);
        stats->kernel_ms = kms;
        stats->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    return TL_OK;
}

extern "C" int tl_two_opt(tl_ctx *c, const float *xy, uint32_t n, const float *dm_packed, const uint32_t *init_pos, int mode,
                          uint32_t *out_pos, float *out_cost, tl_stats *stats)
{
    TL_ENTER(c);
    if (!c || (!xy && !dm_packed) || !out_pos) return fail(c, TL_ERR_BADARG, "tl_two_opt: NULL argument");
    if (mode != TL_MODE_REF_ORDER && mode != TL_MODE_BEST_SWEEP) return fail(c, TL_ERR_BADARG, "tl_two_opt: bad mode %d", mode);
    if (n < 3) return fail(c, TL_ERR_REF_PANICS, "two_opt: n=%u < 3 — the reference underflows `n_indices - 2` (two_opt.rs:17,29)", n);
    if (init_pos && !is_permutation(init_pos, n)) return fail(c, TL_ERR_BADARG, "tl_two_opt: init tour is not a permutation of 0..n-1");
    if (mode == TL_MODE_BEST_SWEEP) return two_opt_best_sweep(c, xy, n, dm_packed, init_pos, out_pos, out_cost, stats);
    if (!dm_packed && (n > lds_max_n(c->lds_bytes) || (c->flags & TL_FLAG_2OPT_FORCE_HBM))) {
        if (n < 4) {  // n == 3: the reference's loops are empty
            for (uint32_t i = 0; i < n; ++i) out_pos[i] = init_pos ? init_pos[i] : i;
            if (stats) { memset(stats, 0, sizeof(*stats)); stats->sweeps = 1; }
            return out_cost ? tl_tour_length(c, xy, nullptr, n, out_pos, out_cost) : TL_OK;
        }
        return two_opt_ref_large(c, xy, n, init_pos, out_pos, out_cost, stats);
    }
    const auto t0 = std::chrono::steady_clock::now();
    HIPCHK(c, hipSetDevice(c->device));
    int rc;
    if ((rc = ensure(c, c->out_pos, (size_t)n * 4)) || (rc = ensure(c, c->out_cost, 4)) || (rc = ensure(c, c->out_stats, TL_STATS_STRIDE * 8))) return rc;
    const float2 *dxy = nullptr;
    const float *ddm = nullptr;
    if (dm_packed) {
        const size_t b = (size_t)n * (n - 1) / 2 * 4;
        if ((rc = ensure(c, c->dm, b))) return rc;
        HIPCHK(c, hipMemcpyAsync(c->dm.p, dm_packed, b, hipMemcpyHostToDevice, c->stream));
        ddm = (const float *)c->dm.p;
    }
    if (xy) {
        if ((rc = ensure(c, c->xy, (size_t)n * 8))) return rc;
        HIPCHK(c, hipMemcpyAsync(c->xy.p, xy, (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
        dxy = (const float2 *)c->xy.p;
    }
    const uint32_t *dinit = nullptr;
    if (init_pos) {
        if ((rc = ensure(c, c->init, (size_t)n * 4))) return rc;
        HIPCHK(c, hipMemcpyAsync(c->init.p, init_pos, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
        dinit = (const uint32_t *)c->init.p;
    }
    if ((rc = two_opt_enqueue(c, dxy, ddm, n, dinit, dinit ? TL_INIT_ARRAY : TL_INIT_IDENTITY, 0, 0, 1, mode,
                              (uint32_t *)c->out_pos.p, (float *)c->out_cost.p, (uint64_t *)c->out_stats.p, c->stream)))
        return rc;
    uint64_t raw[TL_STATS_STRIDE];
    float cost = 0.f;
    HIPCHK(c, hipMemcpyAsync(out_pos, c->out_pos.p, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(&cost, c->out_cost.p, 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(raw, c->out_stats.p, TL_STATS_STRIDE * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (raw[3] == 2) return fail(c, TL_ERR_BADARG, "two_opt: the initial tour holds a position >= n");
    if (raw[3] != 0) return fail(c, TL_ERR_NO_CONVERGE, "two_opt: sweep cap reached");
    if (out_cost) *out_cost = cost;
    double kms = 0;
    tl_last_kernel_ms(c, &kms);
    fill_stats(stats, n, raw, 1, kms, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    return TL_OK;
}

// tl_two_opt + the list of the moves it applied, in the reference's order: what a caller that was handed a progress channel
// (two_opt.rs:9-10; only teeline-qt passes one) replays CityChange / PathUpdate from.  The control wave of the descent's
// workgroup (coordinates) or thread 0 (matrix form) writes the list (row << 16 | column per move, 0xFFFFFFFF where a sweep
// begins); nothing else about the descent changes.
extern "C" int tl_two_opt_trace(tl_ctx *c, const float *xy, uint32_t n, const float *dm_packed, const uint32_t *init_pos, uint32_t *out_pos,
                                float *out_cost, tl_stats *stats, uint32_t *move_log, uint32_t log_cap, uint32_t *log_len)
{
    TL_ENTER(c);
    if (log_len) *log_len = 0;  // every error return leaves an empty log, like the 3-opt / Or-opt / LK variants
    if (!c || (!xy && !dm_packed) || !out_pos || !move_log || !log_len) return fail(c, TL_ERR_BADARG, "tl_two_opt_trace: NULL argument");
    if (n < 3) return fail(c, TL_ERR_REF_PANICS, "two_opt: n=%u < 3 — the reference underflows `n_indices - 2` (two_opt.rs:17,29)", n);
    if (init_pos && !is_permutation(init_pos, n)) return fail(c, TL_ERR_BADARG, "tl_two_opt_trace: init tour is not a permutation of 0..n-1");
    if (!dm_packed && (n > lds_max_n(c->lds_bytes) || n > 65535u))
        return fail(c, TL_ERR_UNSUPPORTED, "tl_two_opt_trace: n=%u exceeds the LDS-resident descent (%u): no move log beyond it", n, lds_max_n(c->lds_bytes));
    const auto t0 = std::chrono::steady_clock::now();
    HIPCHK(c, hipSetDevice(c->device));
    int rc;
    if ((rc = ensure(c, c->out_pos, (size_t)n * 4)) || (rc = ensure(c, c->out_cost, 4)) || (rc = ensure(c, c->out_stats, TL_STATS_STRIDE * 8)) ||
        (rc = ensure(c, c->work, (size_t)(log_cap ? log_cap : 1) * 4)))
        return rc;
    const float2 *dxy = nullptr;
    const float *ddm = nullptr;
    if (dm_packed) {
        const size_t b = (size_t)n * (n - 1) / 2 * 4;
        if ((rc = ensure(c, c->dm, b))) return rc;
        HIPCHK(c, hipMemcpyAsync(c->dm.p, dm_packed, b, hipMemcpyHostToDevice, c->stream));
        ddm = (const float *)c->dm.p;
    }
    if (xy) {
        if ((rc = ensure(c, c->xy, (size_t)n * 8))) return rc;
        HIPCHK(c, hipMemcpyAsync(c->xy.p, xy, (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
        dxy = (const float2 *)c->xy.p;
    }
    const uint32_t *dinit = nullptr;
    if (init_pos) {
        if ((rc = ensure(c, c->init, (size_t)n * 4))) return rc;
        HIPCHK(c, hipMemcpyAsync(c->init.p, init_pos, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
        dinit = (const uint32_t *)c->init.p;
    }
    HIPCHK(c, hipMemsetAsync(c->out_stats.p, 0, TL_STATS_STRIDE * 8, c->stream));
    if ((rc = two_opt_enqueue(c, dxy, ddm, n, dinit, dinit ? TL_INIT_ARRAY : TL_INIT_IDENTITY, 0, 0, 1, TL_MODE_REF_ORDER,
                              (uint32_t *)c->out_pos.p, (float *)c->out_cost.p, (uint64_t *)c->out_stats.p, c->stream, (uint32_t *)c->work.p, log_cap)))
        return rc;
    uint64_t raw[TL_STATS_STRIDE];
    float cost = 0.f;
    HIPCHK(c, hipMemcpyAsync(out_pos, c->out_pos.p, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(&cost, c->out_cost.p, 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(raw, c->out_stats.p, TL_STATS_STRIDE * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (raw[3] == 2) return fail(c, TL_ERR_BADARG, "two_opt: the initial tour holds a position >= n");
    if (raw[3] != 0) return fail(c, TL_ERR_NO_CONVERGE, "two_opt: sweep cap reached");
    *log_len = (uint32_t)raw[15];  // words: moves applied + one mark per sweep after the first; more than log_cap: the log holds the first log_cap
    const uint32_t have = *log_len < log_cap ? *log_len : log_cap;
    // (on the context's own stream: a synchronous hipMemcpy goes through the legacy default stream, which may not meet another
    //  thread's capturing stream — tl_lk records its round loop as a hipGraph; found by tests/test_gpu_threads.py)
    if (have) {
        HIPCHK(c, hipMemcpyAsync(move_log, c->work.p, (size_t)have * 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    if (out_cost) *out_cost = cost;
    double kms = 0;
    tl_last_kernel_ms(c, &kms);
    fill_stats(stats, n, raw, 1, kms, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    return TL_OK;
}

extern "C" uint64_t tl_pack_cost_key(float cost, uint32_t restart)
{
    uint32_t bits;
    memcpy(&bits, &cost, 4);
    return ((uint64_t)bits << 32) | restart;
}

// multi-start = enqueue (asynchronous: upload, descent kernel) + finish (read back, pick the shard's best)
static int multistart_begin(tl_ctx *c, const float *xy, uint32_t n, uint64_t seed, uint32_t first, uint32_t count, int mode)
{
    HIPCHK(c, hipSetDevice(c->device));
    int rc;
    if ((rc = ensure(c, c->xy, (size_t)n * 8)) || (rc = ensure(c, c->out_pos, (size_t)count * n * 4)) ||
        (rc = ensure(c, c->out_cost, (size_t)count * 4)) || (rc = ensure(c, c->out_stats, (size_t)count * TL_STATS_STRIDE * 8)))
        return rc;
    HIPCHK(c, hipMemcpyAsync(c->xy.p, xy, (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
    return two_opt_enqueue(c, (const float2 *)c->xy.p, nullptr, n, nullptr, TL_INIT_SEEDED, seed, first, count, mode,
                           (uint32_t *)c->out_pos.p, (float *)c->out_cost.p, (uint64_t *)c->out_stats.p, c->stream);
}

struct ShardBest {
    uint64_t key = ~0ull;
    uint32_t local = 0;  // index inside the shard
};

static int multistart_finish(tl_ctx *c, uint32_t n, uint32_t first, uint32_t count, float *costs /*count*/, uint64_t *raw /*count x stride*/,
                             ShardBest &best)
{
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(costs, c->out_cost.p, (size_t)count * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(raw, c->out_stats.p, (size_t)count * TL_STATS_STRIDE * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (uint32_t r = 0; r < count; ++r) {
        if (raw[TL_STATS_STRIDE * r + 3] != 0) return fail(c, TL_ERR_NO_CONVERGE, "two_opt: sweep cap reached in restart %u", first + r);
        const uint64_t k = tl_pack_cost_key(costs[r], first + r);
        if (k < best.key) {
            best.key = k;
            best.local = r;
        }
    }
    (void)n;
    return TL_OK;
}

extern "C" int tl_two_opt_multistart(tl_ctx *c, const float *xy, uint32_t n, uint64_t seed, uint32_t first, uint32_t count,
                                     int mode, uint32_t *out_best_pos, float *out_best_cost, uint32_t *out_best_restart,
                                     float *out_costs, tl_stats *stats)
{
    TL_ENTER(c);
    tl_ctx *one[1] = {c};
    return tl_two_opt_multistart_devices(one, 1, xy, n, seed, first, count, mode, out_best_pos, out_best_cost, out_best_restart,
                                         out_costs, stats);
}

// ------------------------------------------------------------------------------------------------
// RCCL inside the library (TL_FLAG_MULTISTART_RCCL): SURVEY.md §8(e)'s collective for ONE process that drives several devices —
// every device reduces its shard's packed (cost, restart) keys to one word, ncclAllReduce(ncclMin, uint64) over the devices'
// communicators (ncclCommInitAll: one process, no bootstrap), ncclBroadcast of the winner's tour from its owner.  librccl is
// loaded on first use (dlopen): a caller that never asks for it pays nothing at start-up and the library has no link-time
// dependency on it.  The host minimum of tl_two_opt_multistart_devices is computed all the same and must agree.
// ------------------------------------------------------------------------------------------------
#include <dlfcn.h>
#include <map>
#include <mutex>
namespace {
struct Rccl {
    void *h = nullptr;
    int (*CommInitAll)(void **, int, const int *) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*Broadcast)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    std::map<std::vector<int>, std::vector<void *>> comms;  // per device list, for the life of the process
    std::mutex mu;
    bool load()
    {
        if (h) return true;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (h) break;
        }
        if (!h) return false;
        CommInitAll = (decltype(CommInitAll))dlsym(h, "ncclCommInitAll");
        AllReduce = (decltype(AllReduce))dlsym(h, "ncclAllReduce");
        Broadcast = (decltype(Broadcast))dlsym(h, "ncclBroadcast");
        GroupStart = (decltype(GroupStart))dlsym(h, "ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))dlsym(h, "ncclGroupEnd");
        GetErrorString = (decltype(GetErrorString))dlsym(h, "ncclGetErrorString");
        if (!CommInitAll || !AllReduce || !Broadcast || !GroupStart || !GroupEnd || !GetErrorString) {
            dlclose(h);
            h = nullptr;
            return false;
        }
        return true;
    }
};
Rccl g_rccl;
constexpr int kNcclUint32 = 3, kNcclUint64 = 5, kNcclMin = 3;  // rccl.h: ncclDataType_t, ncclRedOp_t

// one workgroup: min over a shard of tl_pack_cost_key(cost, first + r) — (f32 bits << 32) | restart, ~0 for an empty shard
__global__ __launch_bounds__(256) void k_shard_key(const float *__restrict__ cost, uint32_t first, uint32_t count, unsigned long long *__restrict__ out)
{
    __shared__ unsigned long long sk[256];
    unsigned long long best = ~0ull;
    for (uint32_t r = threadIdx.x; r < count; r += 256u) {
        const unsigned long long k = ((unsigned long long)__float_as_uint(cost[r]) << 32) | (unsigned long long)(first + r);
        best = k < best ? k : best;
    }
    sk[threadIdx.x] = best;
    __syncthreads();
    for (uint32_t s2 = 128u; s2 > 0u; s2 >>= 1) {
        if (threadIdx.x < s2 && sk[threadIdx.x + s2] < sk[threadIdx.x]) sk[threadIdx.x] = sk[threadIdx.x + s2];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = sk[0];
}
}  // namespace

// restart r's start permutation on the host — the same stream as the kernels' (two_opt_ref.hip: draw k = splitmix64_at(seed + r, n - 1 - i) % (i + 1) for
// i = n-1 .. 1, swaps applied in that order; oracle tlo_restart_perm) — for the descents that run one after the other through the HBM form
static void restart_perm_host(uint32_t n, uint64_t seed, uint64_t r, std::vector<uint32_t> &perm)
{
    perm.resize(n);
    for (uint32_t i = 0; i < n; ++i) perm[i] = i;
    const uint64_t s0 = seed + r;
    for (uint32_t i = n - 1; i >= 1; --i) {
        uint64_t z = s0 + ((uint64_t)(n - 1 - i) + 1) * 0x9E3779B97F4A7C15ULL;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
        z ^= z >> 31;
        const uint32_t j = (uint32_t)(z % ((uint64_t)i + 1));
        const uint32_t t = perm[i];
        perm[i] = perm[j];
        perm[j] = t;
    }
}

// The deal of restarts [first, first + count) over `parts` devices / ranks: contiguous blocks, the first count % parts take one more.
// Host-only; tl_two_opt_multistart_devices deals with it, and a multi-process job's ranks (teeline_amd/host/multistart.py shard_total)
// use the same map, so that a run's result does not depend on how it was spread.
extern "C" int tl_multistart_shard(uint32_t first, uint32_t count, int parts, int part, uint32_t *shard_first, uint32_t *shard_count)
{
    if (!shard_first || !shard_count || parts <= 0 || part < 0 || part >= parts) return TL_ERR_BADARG;
    const uint32_t base = count / (uint32_t)parts, extra = count % (uint32_t)parts;
    *shard_first = first + (uint32_t)part * base + ((uint32_t)part < extra ? (uint32_t)part : extra);
    *shard_count = base + ((uint32_t)part < extra ? 1u : 0u);
    return TL_OK;
}

// North-star config 4 from ONE host process (what the Rust caller has: the reference is single-process): the restarts
// [first, first + count) are dealt in contiguous blocks to the caller's contexts — one per device, created once with
// tl_create(device, ...) — every shard is enqueued before any is waited for, and the winner is the minimum of at most
// n_ctxs packed (cost, restart) keys on the host.  No collective is needed inside the library; ranks of a multi-process
// job (bench.py) min-all-reduce the same key over RCCL instead.
extern "C" int tl_two_opt_multistart_devices(tl_ctx *const *ctxs, int n_ctxs, const float *xy, uint32_t n, uint64_t seed, uint32_t first,
                                             uint32_t count, int mode, uint32_t *out_best_pos, float *out_best_cost,
                                             uint32_t *out_best_restart, float *out_costs, tl_stats *stats)
{
    tl_ctx *c0 = (ctxs && n_ctxs > 0) ? ctxs[0] : nullptr;
    if (!c0 || !xy || !out_best_pos) return fail(c0, TL_ERR_BADARG, "tl_two_opt_multistart: NULL argument");
    for (int d = 0; d < n_ctxs; ++d)
        if (!ctxs[d]) return fail(c0, TL_ERR_BADARG, "tl_two_opt_multistart_devices: context %d is NULL", d);
    std::vector<std::unique_ptr<CtxUse>> uses;
    for (int d = 0; d < n_ctxs; ++d) {
        uses.emplace_back(new CtxUse(ctxs[d]));
        if (!uses.back()->ok) return TL_ERR_BUSY;
    }
    if (count == 0) return fail(c0, TL_ERR_BADARG, "tl_two_opt_multistart: count == 0");
    if (n < 3) return fail(c0, TL_ERR_REF_PANICS, "two_opt: n=%u < 3", n);
    const auto t0 = std::chrono::steady_clock::now();
    struct Shard {
        uint32_t first, count;
    };
    std::vector<Shard> shard((size_t)n_ctxs);
    for (int d = 0; d < n_ctxs; ++d) (void)tl_multistart_shard(first, count, n_ctxs, d, &shard[d].first, &shard[d].count);
    int rc;
    if (n > lds_max_n(c0->lds_bytes)) {
        // beyond the LDS-resident descent (round 5): the restarts one after the other through the HBM form, each
        // shard on its own context — the same start permutations, so the same tours as a (hypothetical) batch would give
        if (mode != TL_MODE_REF_ORDER) return fail(c0, TL_ERR_UNSUPPORTED, "batch 2-opt supports TL_MODE_REF_ORDER only");
        std::vector<uint32_t> perm, pos(n), bestpos(n);
        uint64_t bestkey = ~0ull;
        tl_stats acc{};
        std::vector<float> costs(count);
        for (int d = 0; d < n_ctxs; ++d)
            for (uint32_t r = shard[d].first; r < shard[d].first + shard[d].count; ++r) {
                restart_perm_host(n, seed, r, perm);
                float cst = 0.f;
                tl_stats st1{};
                if ((rc = two_opt_ref_large(ctxs[d], xy, n, perm.data(), pos.data(), &cst, &st1))) {
                    if (d) fail(c0, rc, "device shard %d: %s", d, ctxs[d]->err.c_str());
                    return rc;
                }
                costs[r - first] = cst;
                acc.sweeps += st1.sweeps;
                acc.moves += st1.moves;
                acc.reversed += st1.reversed;
                acc.candidates += st1.candidates;
                acc.kernel_ms += st1.kernel_ms;
                const uint64_t key = tl_pack_cost_key(cst, r);
                if (key < bestkey) {
                    bestkey = key;
                    bestpos = pos;
                }
            }
        memcpy(out_best_pos, bestpos.data(), (size_t)n * 4);
        const uint32_t br = (uint32_t)(bestkey & 0xFFFFFFFFull);
        if (out_best_cost) *out_best_cost = costs[br - first];
        if (out_best_restart) *out_best_restart = br;
        if (out_costs) memcpy(out_costs, costs.data(), (size_t)count * 4);
        if (stats) {
            *stats = acc;
            stats->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        }
        return TL_OK;
    }
    for (int d = 0; d < n_ctxs; ++d)
        if (shard[d].count && (rc = multistart_begin(ctxs[d], xy, n, seed, shard[d].first, shard[d].count, mode))) {
            if (d) fail(c0, rc, "device shard %d: %s", d, ctxs[d]->err.c_str());
            return rc;
        }
    std::vector<float> costs(count);
    std::vector<uint64_t> raw((size_t)count * TL_STATS_STRIDE);
    ShardBest best;
    int best_dev = 0;
    double kms_max = 0;
    for (int d = 0; d < n_ctxs; ++d) {
        if (!shard[d].count) continue;
        const uint32_t off = shard[d].first - first;
        ShardBest b;
        if ((rc = multistart_finish(ctxs[d], n, shard[d].first, shard[d].count, costs.data() + off, raw.data() + (size_t)off * TL_STATS_STRIDE, b))) {
            if (d) fail(c0, rc, "device shard %d: %s", d, ctxs[d]->err.c_str());
            return rc;
        }
        if (b.key < best.key) {
            best = b;
            best_dev = d;
        }
        double kms = 0;
        tl_last_kernel_ms(ctxs[d], &kms);
        kms_max = kms > kms_max ? kms : kms_max;
    }
    tl_ctx *cb = ctxs[best_dev];
    if (c0->flags & TL_FLAG_MULTISTART_RCCL) {
        // ---- the collective over RCCL: key min-all-reduce, then the winner's tour from its owner to every device; device 0 hands it out
        std::lock_guard<std::mutex> lk(g_rccl.mu);
        if (!g_rccl.load()) return fail(c0, TL_ERR_UNSUPPORTED, "TL_FLAG_MULTISTART_RCCL: librccl could not be loaded (%s)", dlerror());
        std::vector<int> devs;
        for (int d = 0; d < n_ctxs; ++d) devs.push_back(ctxs[d]->device);
        auto it = g_rccl.comms.find(devs);
        if (it == g_rccl.comms.end()) {
            std::vector<void *> cm((size_t)n_ctxs, nullptr);
            const int r = g_rccl.CommInitAll(cm.data(), n_ctxs, devs.data());
            if (r != 0) return fail(c0, TL_ERR_HIP, "ncclCommInitAll over %d device(s): %s", n_ctxs, g_rccl.GetErrorString(r));
            it = g_rccl.comms.emplace(devs, cm).first;
        }
        const std::vector<void *> &cm = it->second;
        for (int d = 0; d < n_ctxs; ++d) {  // per device: [key in | key out | tour] in its misc buffer; the shard's key on the device
            tl_ctx *cd = ctxs[d];
            HIPCHK(cd, hipSetDevice(cd->device));
            if ((rc = ensure(cd, cd->misc, 256 + (size_t)n * 4))) return rc;
            if (shard[d].count) hipLaunchKernelGGL(k_shard_key, dim3(1), dim3(256), 0, cd->stream, (const float *)cd->out_cost.p, shard[d].first, shard[d].count, (unsigned long long *)cd->misc.p);
            else HIPCHK(cd, hipMemsetAsync(cd->misc.p, 0xFF, 8, cd->stream));
        }
        int r = g_rccl.GroupStart();
        for (int d = 0; d < n_ctxs && r == 0; ++d) {
            HIPCHK(ctxs[d], hipSetDevice(ctxs[d]->device));
            r = g_rccl.AllReduce(ctxs[d]->misc.p, (unsigned char *)ctxs[d]->misc.p + 64, 1, kNcclUint64, kNcclMin, cm[(size_t)d], ctxs[d]->stream);
        }
        if (r == 0) r = g_rccl.GroupEnd();
        if (r != 0) return fail(c0, TL_ERR_HIP, "ncclAllReduce: %s", g_rccl.GetErrorString(r));
        unsigned long long gkey = 0;
        HIPCHK(c0, hipSetDevice(c0->device));
        HIPCHK(c0, hipMemcpyAsync(&gkey, (unsigned char *)c0->misc.p + 64, 8, hipMemcpyDeviceToHost, c0->stream));
        HIPCHK(c0, hipStreamSynchronize(c0->stream));
        if (gkey != best.key) return fail(c0, TL_ERR_HIP, "RCCL min-all-reduce gave key %llx, the host minimum is %llx", gkey, (unsigned long long)best.key);
        // the owner of restart (gkey & 0xFFFFFFFF) is the root of the broadcast
        r = g_rccl.GroupStart();
        for (int d = 0; d < n_ctxs && r == 0; ++d) {
            HIPCHK(ctxs[d], hipSetDevice(ctxs[d]->device));
            const void *src = d == best_dev ? (const void *)((const uint32_t *)ctxs[d]->out_pos.p + (size_t)best.local * n) : (const void *)((unsigned char *)ctxs[d]->misc.p + 256);
            r = g_rccl.Broadcast(src, (unsigned char *)ctxs[d]->misc.p + 256, n, kNcclUint32, best_dev, cm[(size_t)d], ctxs[d]->stream);
        }
        if (r == 0) r = g_rccl.GroupEnd();
        if (r != 0) return fail(c0, TL_ERR_HIP, "ncclBroadcast: %s", g_rccl.GetErrorString(r));
        HIPCHK(c0, hipSetDevice(c0->device));
        HIPCHK(c0, hipMemcpyAsync(out_best_pos, (unsigned char *)c0->misc.p + 256, (size_t)n * 4, hipMemcpyDeviceToHost, c0->stream));
        HIPCHK(c0, hipStreamSynchronize(c0->stream));
        for (int d = 1; d < n_ctxs; ++d) {  // (every device's stream has finished its part before the call returns)
            HIPCHK(ctxs[d], hipSetDevice(ctxs[d]->device));
            HIPCHK(ctxs[d], hipStreamSynchronize(ctxs[d]->stream));
        }
    } else {
        HIPCHK(cb, hipSetDevice(cb->device));
        HIPCHK(cb, hipMemcpyAsync(out_best_pos, (const uint32_t *)cb->out_pos.p + (size_t)best.local * n, (size_t)n * 4, hipMemcpyDeviceToHost, cb->stream));
        HIPCHK(cb, hipStreamSynchronize(cb->stream));  // never the legacy stream: see tl_two_opt_trace
    }
    const uint32_t best_restart = (uint32_t)(best.key & 0xFFFFFFFFull);
    if (out_best_cost) *out_best_cost = costs[best_restart - first];
    if (out_best_restart) *out_best_restart = best_restart;
    if (out_costs) memcpy(out_costs, costs.data(), (size_t)count * 4);
    fill_stats(stats, n, raw.data(), count, kms_max, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    return TL_OK;
}

// A population of explicit tours, each refined by its own REF_ORDER descent (one workgroup per individual): what a
// memetic GA or any caller holding several seeds needs; every individual's result equals tl_two_opt on it alone.
extern "C" int tl_two_opt_population(tl_ctx *c, const float *xy, uint32_t n, const float *dm_packed, const uint32_t *init_pos,
                                     uint32_t count, uint32_t *out_pos, float *out_costs, tl_stats *stats)
{
    TL_ENTER(c);
    if (!c || (!xy && !dm_packed) || !init_pos || !out_pos) return fail(c, TL_ERR_BADARG, "tl_two_opt_population: NULL argument");
    if (count == 0) return fail(c, TL_ERR_BADARG, "tl_two_opt_population: count == 0");
    if (n < 3) return fail(c, TL_ERR_REF_PANICS, "two_opt: n=%u < 3", n);
    for (uint32_t r = 0; r < count; ++r)
        if (!is_permutation(init_pos + (size_t)r * n, n))
            return fail(c, TL_ERR_BADARG, "tl_two_opt_population: tour %u is not a permutation of 0..n-1", r);
    const auto t0 = std::chrono::steady_clock::now();
    HIPCHK(c, hipSetDevice(c->device));
    int rc;
    if (!dm_packed && n > lds_max_n(c->lds_bytes)) {
        // beyond the LDS-resident descent: the tours one after the other through the HBM form (every individual's result equals tl_two_opt on it alone)
        tl_stats acc{};
        for (uint32_t r = 0; r < count; ++r) {
            float cst = 0.f;
            tl_stats st1{};
            if ((rc = two_opt_ref_large(c, xy, n, init_pos + (size_t)r * n, out_pos + (size_t)r * n, &cst, &st1))) return rc;
            if (out_costs) out_costs[r] = cst;
            acc.sweeps += st1.sweeps;
            acc.moves += st1.moves;
            acc.reversed += st1.reversed;
            acc.candidates += st1.candidates;
            acc.kernel_ms += st1.kernel_ms;
        }
        if (stats) {
            *stats = acc;
            stats->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        }
        return TL_OK;
    }
    const size_t dm_bytes = dm_packed ? (size_t)n * (n - 1) / 2 * 4 : 0;
    if ((rc = ensure(c, c->init, (size_t)count * n * 4)) || (rc = ensure(c, c->out_pos, (size_t)count * n * 4)) ||
        (rc = ensure(c, c->out_cost, (size_t)count * 4)) || (rc = ensure(c, c->out_stats, (size_t)count * TL_STATS_STRIDE * 8)))
        return rc;
    if (dm_packed) {
        if ((rc = ensure(c, c->dm, dm_bytes))) return rc;
        HIPCHK(c, hipMemcpyAsync(c->dm.p, dm_packed, dm_bytes, hipMemcpyHostToDevice, c->stream));
    } else {
        if ((rc = ensure(c, c->xy, (size_t)n * 8))) return rc;
        HIPCHK(c, hipMemcpyAsync(c->xy.p, xy, (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
    }
    HIPCHK(c, hipMemcpyAsync(c->init.p, init_pos, (size_t)count * n * 4, hipMemcpyHostToDevice, c->stream));
    if ((rc = two_opt_enqueue(c, dm_packed ? nullptr : (const float2 *)c->xy.p, dm_packed ? (const float *)c->dm.p : nullptr, n,
                              (const uint32_t *)c->init.p, TL_INIT_ARRAY, 0, 0, count, TL_MODE_REF_ORDER, (uint32_t *)c->out_pos.p,
                              (float *)c->out_cost.p, (uint64_t *)c->out_stats.p, c->stream)))
        return rc;
    std::vector<uint64_t> raw((size_t)count * TL_STATS_STRIDE);
    HIPCHK(c, hipMemcpyAsync(out_pos, c->out_pos.p, (size_t)count * n * 4, hipMemcpyDeviceToHost, c->stream));
    std::vector<float> costs(count);
    HIPCHK(c, hipMemcpyAsync(costs.data(), c->out_cost.p, (size_t)count * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(raw.data(), c->out_stats.p, (size_t)count * TL_STATS_STRIDE * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (uint32_t r = 0; r < count; ++r)
        if (raw[TL_STATS_STRIDE * r + 3] != 0) return fail(c, TL_ERR_NO_CONVERGE, "two_opt: sweep cap reached in tour %u", r);
    if (out_costs) memcpy(out_costs, costs.data(), (size_t)count * 4);
    double kms = 0;
    tl_last_kernel_ms(c, &kms);
    fill_stats(stats, n, raw.data(), count, kms, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    return TL_OK;
}

// ------------------------------------------------------------------------------------------------
// 2-opt, TL_MODE_BEST_SWEEP (this build's own mode; specification: oracle tlo_two_opt_best)
// ------------------------------------------------------------------------------------------------
static int two_opt_best_sweep(tl_ctx *c, const float *xy, uint32_t n, const float *dm_packed, const uint32_t *init_pos,
                              uint32_t *out_pos, float *out_cost, tl_stats *stats)
{
    if (dm_packed || !xy) return fail(c, TL_ERR_UNSUPPORTED, "TL_MODE_BEST_SWEEP needs EUC_2D coordinates (dm_packed must be NULL)");
    if (n > 65535) return fail(c, TL_ERR_UNSUPPORTED, "TL_MODE_BEST_SWEEP: n=%u > 65535 (packed (i,j) key)", n);
    const auto t0 = std::chrono::steady_clock::now();
    HIPCHK(c, hipSetDevice(c->device));
    if (stats) memset(stats, 0, sizeof(*stats));
    int rc;
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const uint32_t n_pad = ((n + 64u + 63u) / 64u) * 64u, ntile_cap = (((n_pad >> 6) + 63u) / 64u) * 64u;
    const uint32_t nblocks = n >= 4 ? best_sweep_scan_blocks(n) : 1;
    const size_t o_perm = 0, o_P = up((size_t)n * 4), o_box = up(o_P + (size_t)(n_pad + 1) * 8), o_msq = up(o_box + (size_t)ntile_cap * 16),
                 o_par = up(o_msq + (size_t)ntile_cap * 4), o_rk = up(o_par + (size_t)nblocks * 8), o_cnt = up(o_rk + (size_t)n * 8), total = o_cnt + 256;
    if ((rc = ensure(c, c->xy, (size_t)n * 8)) || (rc = ensure(c, c->work, total)) || (rc = ensure(c, c->out_cost, 4))) return rc;
    unsigned char *w = (unsigned char *)c->work.p;
    std::vector<uint32_t> ident;
    if (!init_pos) {
        ident.resize(n);
        for (uint32_t i = 0; i < n; ++i) ident[i] = i;
        init_pos = ident.data();
    }
    HIPCHK(c, hipMemcpyAsync(c->xy.p, xy, (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(w + o_perm, init_pos, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemsetAsync(w + o_cnt, 0, 128, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    BestSweepArgs A{};
    A.xy = (const float2 *)c->xy.p;
    A.perm = (uint32_t *)(w + o_perm);
    A.P = (float2 *)(w + o_P);
    A.tbox = (float4 *)(w + o_box);
    A.tmsq = (float *)(w + o_msq);
    A.partials = (unsigned long long *)(w + o_par);
    A.rowkey = (unsigned long long *)(w + o_rk);
    A.move = (uint32_t *)(w + o_cnt + 64);  // (zeroed with the counters: no move yet)
    A.counters = (uint64_t *)(w + o_cnt);
    A.n = n;
    A.n_pad = n_pad;
    A.ntile_cap = ntile_cap;
    uint64_t cnt[4] = {1, 0, 1, 0};  // n == 3: one empty sweep
    HIPCHK(c, hipEventRecord(c->ev0, c->stream));
    if (n >= 4) {
        HIPCHK(c, launch_best_sweep_init(A, c->stream));
        const uint64_t cap = 64ull * n + 1024;
        // 64 sweeps (scan + apply) per poll of the done flag; after the first batch the same 64 sweeps replay as ONE hipGraph launch — a
        // sweep is two short dependent kernels, and what it costs is mostly the gaps between separately enqueued launches (as in tl_lk)
        constexpr int kSweepsPerPoll = 64;
        hipGraph_t graph = nullptr;
        hipGraphExec_t gexec = nullptr;
        bool first = true, graph_ok = true;
        int rc_loop = TL_OK;
        for (;;) {
            if (!first && graph_ok && !gexec) {
                graph_ok = hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal) == hipSuccess;
                if (graph_ok) {
                    hipError_t le = hipSuccess;
                    for (int r = 0; r < kSweepsPerPoll && le == hipSuccess; ++r) le = launch_best_sweep_round(A, c->stream);
                    const hipError_t ce = hipStreamEndCapture(c->stream, &graph);
                    graph_ok = le == hipSuccess && ce == hipSuccess && graph && hipGraphInstantiate(&gexec, graph, nullptr, nullptr, 0) == hipSuccess;
                }
                if (!graph_ok) {  // separately enqueued launches from here on, after making sure the stream has left capture mode
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
            else for (int r = 0; r < kSweepsPerPoll && e == hipSuccess; ++r) e = launch_best_sweep_round(A, c->stream);  // kernels no-op once done
            if (e == hipSuccess) e = hipMemcpyAsync(cnt, A.counters, 32, hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
            if (e != hipSuccess) {
                (void)hipGetLastError();
                rc_loop = fail(c, TL_ERR_HIP, "two_opt (BEST_SWEEP): %s", hipGetErrorString(e));
                break;
            }
            first = false;
            if (cnt[2]) break;
            if (cnt[0] > cap) {
                rc_loop = fail(c, TL_ERR_NO_CONVERGE, "two_opt (BEST_SWEEP): sweep cap reached");
                break;
            }
        }
        if (gexec) (void)hipGraphExecDestroy(gexec);
        if (graph) (void)hipGraphDestroy(graph);
        if (rc_loop != TL_OK) return rc_loop;
    }
    HIPCHK(c, launch_tour_length(A.xy, nullptr, n, A.perm, (float *)c->out_cost.p, c->stream));
    HIPCHK(c, hipEventRecord(c->ev1, c->stream));
    c->ev_valid = true;
    float cost = 0.f;
    HIPCHK(c, hipMemcpyAsync(out_pos, A.perm, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(&cost, c->out_cost.p, 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (out_cost) *out_cost = cost;
    if (stats) {
        stats->sweeps = cnt[0];
        stats->moves = cnt[1];
        stats->reversed = cnt[3];
        stats->candidates = cnt[0] * (n >= 4 ? (uint64_t)(n - 3) * (n - 2) / 2 : 0);
        double kms = 0;
        tl_last_kernel_ms(c, &kms);
        stats->kernel_ms = kms;
        stats->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    return TL_OK;
}

