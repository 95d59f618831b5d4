// tl_api.hip — host side of the C ABI declared in include/teeline_gpu.h.
//
// A tl_ctx owns one HIP stream, a grow-only device workspace and a pair of HIP events; every entry
// point validates its arguments on the host (shapes, permutation validity, size limits of the
// kernel it is about to launch) before anything reaches the GPU.  There is no CPU fallback.
#include "../../include/teeline_gpu.h"
#include "tl_kernels.h"

#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <atomic>
#include <string>
#include <thread>
#include <vector>

using namespace tl;

#ifndef TL_MAX_SWEEPS
#define TL_MAX_SWEEPS (1u << 20)  // status 1 beyond (never reached by a descent: every move shortens the tour); tuning variants
                                  // are built with a small cap so that a wrong experimental kernel ends instead of hanging the GPU
#endif

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
};

struct tl_ctx {
    int device = 0;
    uint32_t flags = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool ev_valid = false;
    int cus = 0, lds_bytes = 0;
    std::string arch;
    std::string err;
    DevBuf xy, dm, init, out_pos, out_cost, out_stats, misc, work, dmfull, kd, fx;
    uint32_t dm_n = 0;
    int dm_layout = -1;
    // which host thread is inside an entry point with this context (default id: none) and how deep (entries call entries)
    std::atomic<std::thread::id> owner{};
    int depth = 0;
};

// A tl_ctx is single-threaded (include/teeline_gpu.h): its stream, event pair, workspace and error string belong to the call in
// progress.  Every entry point that takes a context enters through this guard; a second host thread that arrives while another is
// inside gets TL_ERR_BUSY back at once — nothing of the context is touched, not even its error string — instead of racing on
// the workspace.  Re-entry by the owning thread (tl_lk -> tl_nearest_neighbor -> tl_tour_length ...) is counted.
struct CtxUse {
    tl_ctx *c;
    bool ok = true;
    explicit CtxUse(tl_ctx *c_) : c(c_)
    {
        if (!c) return;
        const std::thread::id me = std::this_thread::get_id();
        if (c->owner.load(std::memory_order_acquire) == me) {
            ++c->depth;
            return;
        }
        std::thread::id none{};
        if (c->owner.compare_exchange_strong(none, me, std::memory_order_acq_rel)) {
            c->depth = 1;
            return;
        }
        ok = false;
    }
    ~CtxUse()
    {
        if (c && ok && --c->depth == 0) c->owner.store(std::thread::id(), std::memory_order_release);
    }
    CtxUse(const CtxUse &) = delete;
    CtxUse &operator=(const CtxUse &) = delete;
};
#define TL_ENTER(c)      \
    CtxUse tl_use_((c)); \
    if (!tl_use_.ok) return TL_ERR_BUSY

static thread_local std::string g_create_err;  // tl_last_error(NULL): the calling thread's last tl_create failure

static int knn_form(const tl_ctx *c);
// lin_kernighan cut-over (measured): the LDS-resident single workgroup never wins -> 0
static constexpr uint32_t kLkSmallMaxN = 0, kLkSmallWave64MaxN = 0, kLkSmall256MaxN = 0;

static int fail(tl_ctx *c, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (c) c->err = buf;
    else g_create_err = buf;
    return code;
}

#define HIPCHK(c, expr)                                                                           \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess) {                                                                   \
            (void)hipGetLastError(); /* the thread's sticky error: a later launch check must not report this one again */ \
            return fail((c), _e == hipErrorOutOfMemory ? TL_ERR_NOMEM : TL_ERR_HIP, "%s: %s", #expr, \
                        hipGetErrorString(_e));                                                   \
        }                                                                                         \
    } while (0)

static int ensure(tl_ctx *c, DevBuf &b, size_t bytes)
{
    if (bytes <= b.cap) return TL_OK;
    if (b.p) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, hipFree(b.p));
        b.p = nullptr;
        b.cap = 0;
    }
    size_t cap = bytes + bytes / 8 + 256;
    HIPCHK(c, hipMalloc(&b.p, cap));
    b.cap = cap;
    return TL_OK;
}

static bool is_permutation(const uint32_t *p, uint32_t n)  // validate_tour, src/tsp/mod.rs:1620-1634
{
    std::vector<unsigned char> seen(n, 0);
    for (uint32_t i = 0; i < n; ++i) {
        if (p[i] >= n || seen[p[i]]) return false;
        seen[p[i]] = 1;
    }
    return true;
}

// ------------------------------------------------------------------------------------------------
extern "C" int tl_abi_version(void) { return TL_ABI_VERSION; }
#if defined(TL_JITTER)
extern "C" const char *tl_version(void) { return "teeline-gpu 0.1 (gfx950) +jitter"; }  // race-stress build (tl_device.h)
#elif defined(TL_TUNE)
extern "C" const char *tl_version(void) { return "teeline-gpu 0.1 (gfx950) +tune"; }  // tuning build: rejected kernel forms included
#else
extern "C" const char *tl_version(void) { return "teeline-gpu 0.1 (gfx950)"; }
#endif

extern "C" const char *tl_last_error(const tl_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

extern "C" int tl_create(int device, uint32_t flags, tl_ctx **out)
{
    if (!out) return fail(nullptr, TL_ERR_BADARG, "tl_create: out is NULL");
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(nullptr, TL_ERR_NO_DEVICE, "tl_create: no HIP device (%s) — libteeline_gpu has no CPU fallback",
                    e != hipSuccess ? hipGetErrorString(e) : "count=0");
    if (device < 0 || device >= count)
        return fail(nullptr, TL_ERR_BADARG, "tl_create: device %d out of range [0,%d)", device, count);
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device)) != hipSuccess)
        return fail(nullptr, TL_ERR_HIP, "hipGetDeviceProperties: %s", hipGetErrorString(e));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, TL_ERR_NO_DEVICE, "tl_create: device %d is %s; this library is built for gfx950 only",
                    device, prop.gcnArchName);
#ifndef TL_TUNE
    if (flags & TL_TUNE_ONLY_FLAGS)
        return fail(nullptr, TL_ERR_UNSUPPORTED, "tl_create: flags 0x%x name kernel forms only the tuning build carries (libteeline_gpu_tune.so)",
                    flags & TL_TUNE_ONLY_FLAGS);
#endif
    tl_ctx *c = new tl_ctx();
    c->device = device;
    c->flags = flags;
    c->cus = prop.multiProcessorCount;
    c->lds_bytes = (int)prop.sharedMemPerBlock;
    {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, device) == hipSuccess && v > c->lds_bytes)
            c->lds_bytes = v;
    }
    c->arch = prop.gcnArchName;
    if ((e = hipSetDevice(device)) != hipSuccess || (e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipEventCreate(&c->ev0)) != hipSuccess || (e = hipEventCreate(&c->ev1)) != hipSuccess) {
        int rc = fail(nullptr, TL_ERR_HIP, "tl_create: %s", hipGetErrorString(e));
        delete c;
        return rc;
    }
    *out = c;
    return TL_OK;
}

extern "C" void tl_destroy(tl_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (DevBuf *b : {&c->xy, &c->dm, &c->init, &c->out_pos, &c->out_cost, &c->out_stats, &c->misc, &c->work, &c->dmfull, &c->kd, &c->fx})
        if (b->p) (void)hipFree(b->p);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" int tl_device_info(const tl_ctx *c, int *cus, int *lds_bytes, char *arch, size_t arch_len)
{
    if (!c) return TL_ERR_BADARG;
    if (cus) *cus = c->cus;
    if (lds_bytes) *lds_bytes = c->lds_bytes;
    if (arch && arch_len) snprintf(arch, arch_len, "%s", c->arch.c_str());
    return TL_OK;
}

static uint32_t lds_max_n(int lds_bytes)
{
    // largest n whose LDS image (10 B per city on a padded length + control + queues) fits one workgroup
    uint32_t lo = 0, hi = 65535;  // u16 tour entries, (i<<16|j) keys
    while (lo < hi) {
        const uint32_t mid = lo + (hi - lo + 1) / 2;
        if (two_opt_ref_lds_bytes(mid, nullptr, TL_TWO_OPT_NT) <= (size_t)lds_bytes) lo = mid;
        else hi = mid - 1;
    }
    return lo;
}

extern "C" uint32_t tl_two_opt_lds_max_n(const tl_ctx *c) { return c ? lds_max_n(c->lds_bytes) : 0u; }

// ------------------------------------------------------------------------------------------------
extern "C" int tl_last_kernel_ms(tl_ctx *c, double *ms)
{
    TL_ENTER(c);
    if (!c || !ms) return TL_ERR_BADARG;
    if (!c->ev_valid) return fail(c, TL_ERR_BADARG, "tl_last_kernel_ms: no kernel sequence recorded yet");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipEventSynchronize(c->ev1));
    float f = 0.f;
    HIPCHK(c, hipEventElapsedTime(&f, c->ev0, c->ev1));
    *ms = (double)f;
    return TL_OK;
}

extern "C" int tl_dm_build_dev(tl_ctx *c, const float *d_xy, uint32_t n, int dist, int layout, float *d_out, void *stream)
{
    TL_ENTER(c);
    if (!c || !d_xy || !d_out) return fail(c, TL_ERR_BADARG, "tl_dm_build_dev: NULL argument");
    if (n < 2) return fail(c, TL_ERR_BADARG, "distance matrix requires at least 2 points");  // distance_matrix.rs:124-126
    if ((dist != TL_DIST_EUC2D && dist != TL_DIST_GEO) || (layout != TL_DM_PACKED_LOWER && layout != TL_DM_FULL))
        return fail(c, TL_ERR_BADARG, "tl_dm_build_dev: bad dist/layout");
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    HIPCHK(c, hipEventRecord(c->ev0, s));
    HIPCHK(c, launch_dm_build(reinterpret_cast<const float2 *>(d_xy), n, dist, layout, d_out, s));
    HIPCHK(c, hipEventRecord(c->ev1, s));
    c->ev_valid = true;
    return TL_OK;
}

extern "C" int tl_selftest_sqrt(tl_ctx *c, uint32_t first_bits, uint64_t count, uint64_t *mismatches, uint32_t *first_bad_bits)
{
    TL_ENTER(c);
    if (!c || !mismatches) return fail(c, TL_ERR_BADARG, "tl_selftest_sqrt: NULL argument");
    HIPCHK(c, hipSetDevice(c->device));
    int rc;
    if ((rc = ensure(c, c->misc, 16))) return rc;
    unsigned long long h[2] = {0ull, 0xFFFFFFFFull};
    HIPCHK(c, hipMemcpyAsync(c->misc.p, h, 16, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, launch_selftest_sqrt(first_bits, count, (unsigned long long *)c->misc.p, (uint32_t *)((char *)c->misc.p + 8), c->stream));
    HIPCHK(c, hipMemcpyAsync(h, c->misc.p, 16, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    *mismatches = h[0];
    if (first_bad_bits) *first_bad_bits = (uint32_t)h[1];
    return TL_OK;
}

extern "C" int tl_dm_build(tl_ctx *c, const float *xy, uint32_t n, int dist, int layout, float *out_host, double *kernel_ms)
{
    TL_ENTER(c);
    if (!c || !xy) return fail(c, TL_ERR_BADARG, "tl_dm_build: NULL argument");
    if (n < 2) return fail(c, TL_ERR_BADARG, "distance matrix requires at least 2 points");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t elems = layout == TL_DM_FULL ? (size_t)n * n : (size_t)n * (n - 1) / 2;
    int rc;
    if ((rc = ensure(c, c->xy, (size_t)n * 8))) return rc;
    DevBuf &dst = layout == TL_DM_FULL ? c->dmfull : c->dm;
    if ((rc = ensure(c, dst, elems * 4))) return rc;
    HIPCHK(c, hipMemcpyAsync(c->xy.p, xy, (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
    if ((rc = tl_dm_build_dev(c, (const float *)c->xy.p, n, dist, layout, (float *)dst.p, nullptr))) return rc;
    if (out_host) HIPCHK(c, hipMemcpyAsync(out_host, dst.p, elems * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (kernel_ms) tl_last_kernel_ms(c, kernel_ms);
    return TL_OK;
}

extern "C" int tl_dm_is_euc2d(tl_ctx *c, const float *xy, const float *dm_packed, uint32_t n, int *is_euc2d)
{
    TL_ENTER(c);
    if (!c || !xy || !dm_packed || !is_euc2d) return fail(c, TL_ERR_BADARG, "tl_dm_is_euc2d: NULL argument");
    *is_euc2d = 1;
    if (n < 2) return TL_OK;
    HIPCHK(c, hipSetDevice(c->device));
    int rc;
    const size_t b = (size_t)n * (n - 1) / 2 * 4;
    if ((rc = ensure(c, c->xy, (size_t)n * 8)) || (rc = ensure(c, c->dm, b)) || (rc = ensure(c, c->misc, 16))) return rc;
    HIPCHK(c, hipMemcpyAsync(c->xy.p, xy, (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->dm.p, dm_packed, b, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemsetAsync(c->misc.p, 0, 4, c->stream));
    HIPCHK(c, launch_dm_compare((const float2 *)c->xy.p, n, (const float *)c->dm.p, (uint32_t *)c->misc.p, c->stream));
    uint32_t differs = 0;
    HIPCHK(c, hipMemcpyAsync(&differs, c->misc.p, 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    *is_euc2d = differs ? 0 : 1;
    return TL_OK;
}

extern "C" int tl_tour_length(tl_ctx *c, const float *xy, const float *dm_packed, uint32_t n, const uint32_t *perm, float *out_cost)
{
    TL_ENTER(c);
    if (!c || (!xy && !dm_packed) || !perm || !out_cost) return fail(c, TL_ERR_BADARG, "tl_tour_length: NULL argument");
    if (n < 2) {  // distance_matrix.rs:236-238
        *out_cost = 0.0f;
        return TL_OK;
    }
    for (uint32_t i = 0; i < n; ++i)
        if (perm[i] >= n) return fail(c, TL_ERR_BADARG, "tl_tour_length: position %u out of range at %u", perm[i], i);
    HIPCHK(c, hipSetDevice(c->device));
    int rc;
    if ((rc = ensure(c, c->init, (size_t)n * 4)) || (rc = ensure(c, c->out_cost, 4))) return rc;
    const float2 *dxy = nullptr;
    const float *ddm = nullptr;
    if (dm_packed) {
        const size_t b = (size_t)n * (n - 1) / 2 * 4;
        if ((rc = ensure(c, c->dm, b))) return rc;
        HIPCHK(c, hipMemcpyAsync(c->dm.p, dm_packed, b, hipMemcpyHostToDevice, c->stream));
        ddm = (const float *)c->dm.p;
    } else {
        if ((rc = ensure(c, c->xy, (size_t)n * 8))) return rc;
        HIPCHK(c, hipMemcpyAsync(c->xy.p, xy, (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
        dxy = (const float2 *)c->xy.p;
    }
    HIPCHK(c, hipMemcpyAsync(c->init.p, perm, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, launch_tour_length(dxy, ddm, n, (const uint32_t *)c->init.p, (float *)c->out_cost.p, c->stream));
    HIPCHK(c, hipMemcpyAsync(out_cost, c->out_cost.p, 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return TL_OK;
}

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
    HIPCHK(c, hipEventRecord(c->ev0, s));
    if (d_dm) {
        // the packed triangle (reference layout) is expanded to a full row-major matrix once per call: a row scan then
        // gathers inside one 4n-byte row instead of one cache line per column (two_opt_dm.hip)
        int rc2;
        if ((rc2 = ensure(c, c->dmfull, (size_t)n * n * 4))) return rc2;
        HIPCHK(c, launch_dm_expand_full(d_dm, n, (float *)c->dmfull.p, s));
        A.dm_full = (const float *)c->dmfull.p;
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
        HIPCHK(c, launch_two_opt_ref_lds(A, count, !(c->flags & TL_FLAG_NO_PRUNE), s, (c->flags & TL_FLAG_COUNT_WORK) != 0, c->cus, c->lds_bytes, force_nt));
    }
    HIPCHK(c, hipEventRecord(c->ev1, s));
    c->ev_valid = true;
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
    const uint32_t base = count / (uint32_t)n_ctxs, extra = count % (uint32_t)n_ctxs;
    uint32_t at = first;
    for (int d = 0; d < n_ctxs; ++d) {
        shard[d] = {at, base + ((uint32_t)d < extra ? 1u : 0u)};
        at += shard[d].count;
    }
    int rc;
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
    HIPCHK(cb, hipSetDevice(cb->device));
    HIPCHK(cb, hipMemcpyAsync(out_best_pos, (const uint32_t *)cb->out_pos.p + (size_t)best.local * n, (size_t)n * 4, hipMemcpyDeviceToHost, cb->stream));
    HIPCHK(cb, hipStreamSynchronize(cb->stream));  // never the legacy stream: see tl_two_opt_trace
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
                 o_par = up(o_msq + (size_t)ntile_cap * 4), o_cnt = up(o_par + (size_t)nblocks * 8), total = o_cnt + 256;
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
    HIPCHK(c, hipMemsetAsync(w + o_cnt, 0, 64, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    BestSweepArgs A{};
    A.xy = (const float2 *)c->xy.p;
    A.perm = (uint32_t *)(w + o_perm);
    A.P = (float2 *)(w + o_P);
    A.tbox = (float4 *)(w + o_box);
    A.tmsq = (float *)(w + o_msq);
    A.partials = (unsigned long long *)(w + o_par);
    A.counters = (uint64_t *)(w + o_cnt);
    A.n = n;
    A.n_pad = n_pad;
    A.ntile_cap = ntile_cap;
    uint64_t cnt[4] = {1, 0, 1, 0};  // n == 3: one empty sweep
    HIPCHK(c, hipEventRecord(c->ev0, c->stream));
    if (n >= 4) {
        HIPCHK(c, launch_best_sweep_init(A, c->stream));
        const uint64_t cap = 64ull * n + 1024;
        for (;;) {
            for (int r = 0; r < 32; ++r) HIPCHK(c, launch_best_sweep_round(A, c->stream));  // kernels no-op once done
            HIPCHK(c, hipMemcpyAsync(cnt, A.counters, 32, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            if (cnt[2]) break;
            if (cnt[0] > cap) return fail(c, TL_ERR_NO_CONVERGE, "two_opt (BEST_SWEEP): sweep cap reached");
        }
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

// ------------------------------------------------------------------------------------------------
// 3-opt
// ------------------------------------------------------------------------------------------------
struct ThreeOptSetup {
    ThreeOptArgs A{};
    uint32_t nblocks = 0;
    bool dm = false;
};

static uint32_t three_opt_max_n(const tl_ctx *c)
{
    // (i, j) and (k, case) travel as packed 16-bit fields; k_three_opt_pick stages the tour in LDS (4 B per city next to
    // its static block); the workspace holds an n x (n+1) f32 matrix
    const uint32_t by_lds = (uint32_t)((c->lds_bytes > 2048 ? c->lds_bytes - 2048 : 0) / 4);
    return by_lds < 65535u ? by_lds : 65535u;
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
                 o_cnt = up(o_best + sizeof(ThreeOptBest)), o_dt = up(o_cnt + 16), total = up(o_dt + (size_t)n * (n + 1) * 4);
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
    HIPCHK(c, launch_three_opt_pass(S.A, S.nblocks, S.dm, 0, c->stream));
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
    HIPCHK(c, hipEventRecord(c->ev0, c->stream));
    uint64_t passes = 0, moves = 0;
    const uint64_t cap = 64ull * n + 1024;  // safety cap, far above any observed pass count
    for (;;) {
        HIPCHK(c, launch_three_opt_pass(S.A, S.nblocks, S.dm, 1, c->stream));
        ThreeOptBest b{};
        HIPCHK(c, hipMemcpyAsync(&b, S.A.best, sizeof(b), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        ++passes;
        if (!b.found) break;  // three_opt.rs:36-45
        if (move_log && moves < log_cap) {
            uint32_t *w = move_log + 4 * moves;
            w[0] = b.ij >> 16;
            w[1] = b.ij & 0xFFFFu;
            w[2] = b.kc >> 3;
            w[3] = b.kc & 7u;
        }
        ++moves;
        if (passes > cap) return fail(c, TL_ERR_NO_CONVERGE, "three_opt: pass cap reached");
    }
    HIPCHK(c, hipEventRecord(c->ev1, c->stream));
    c->ev_valid = true;
    HIPCHK(c, hipMemcpyAsync(out_pos, S.A.perm, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
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
// host loop here already reads every move back (one word pair per pass), so the list costs nothing.
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
    HIPCHK(c, hipEventRecord(c->ev0, c->stream));
    uint64_t passes = 0, moves = 0;
    const uint64_t cap = 64ull * n + 1024;
    for (;;) {  // or_opt.rs:45 while let Some(best) = find_best_move(..)
        HIPCHK(c, launch_or_opt_pass(A, dm, 1, c->stream, c->lds_bytes));
        OrOptBest b{};
        HIPCHK(c, hipMemcpyAsync(&b, A.best, sizeof(b), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        ++passes;
        if (!b.found) break;
        if (move_log && moves < log_cap) {
            uint32_t *w = move_log + 4 * moves;
            w[0] = b.i;
            w[1] = b.j;
            w[2] = b.seg_len;
            w[3] = b.reversed;
        }
        ++moves;
        if (passes > cap) return fail(c, TL_ERR_NO_CONVERGE, "or_opt: pass cap reached");
    }
    HIPCHK(c, hipEventRecord(c->ev1, c->stream));
    c->ev_valid = true;
    HIPCHK(c, hipMemcpyAsync(out_pos, A.perm, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
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
// (or_opt.rs:40-42,62-67,70-72); the host loop here already reads every move back.
extern "C" int tl_or_opt_trace(tl_ctx *c, const float *xy, uint32_t n, const float *dm_packed, const uint32_t *init_pos,
                               uint32_t *out_pos, float *out_cost, tl_stats *stats, uint32_t *move_log, uint32_t log_cap, uint32_t *log_len)
{
    TL_ENTER(c);
    if (!move_log || !log_len) return fail(c, TL_ERR_BADARG, "tl_or_opt_trace: NULL argument");
    return or_opt_run(c, xy, n, dm_packed, init_pos, out_pos, out_cost, stats, move_log, log_cap, log_len);
}

static bool max_depth_ge2_split(uint32_t) { return true; }
// A tune-only create flag: always clear in the product build (tl_create refuses them), so the branches it selects fold away.
#ifdef TL_TUNE
static inline uint32_t tune_flags(const tl_ctx *c) { return c->flags; }
#else
static inline uint32_t tune_flags(const tl_ctx *) { return 0u; }
#endif
static int knn_form(const tl_ctx *c) { return (tune_flags(c) & TL_FLAG_KNN_1LANE) ? 1 : (tune_flags(c) & TL_FLAG_KNN_4LANES) ? 4 : 0; }

// lin_kernighan::build_candidates (lin_kernighan.rs:12-27) into d_cand (n x k): the reference's kd-tree k-NN — tree built and
// queried on the device (kdtree.hip) — or, under the TL_FLAG_KNN_* flags, the
// brute-force scan in (distance, position) order (identical lists wherever no two candidates of a city tie in f32 distance).
// d_xy must already hold xy (enqueued on c->stream).
static int build_candidates_dev(tl_ctx *c, const float *xy_host, const float2 *d_xy, uint32_t n, uint32_t k, uint32_t *d_cand)
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
    if (k > 16) return fail(c, TL_ERR_UNSUPPORTED, "tl_build_candidates: k=%u > 16", k);
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
static int nn_seed_dev(tl_ctx *c, const float2 *d_xy, uint32_t n, uint32_t n_nearest, uint32_t *d_path)
{
    uint32_t k = n_nearest > n - 1 ? n - 1 : n_nearest;
    if (k > 16) return fail(c, TL_ERR_UNSUPPORTED, "nearest_neighbor: n_nearest=%u > 16", n_nearest);
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
static int lk_run(tl_ctx *c, const float *xy, uint32_t n, const float *dm_packed, const uint32_t *init_pos, const tl_lk_opts *opts,
                  uint64_t seed, uint32_t *out_pos, float *out_cost, tl_stats *stats, uint32_t *snap_pos, float *snap_dist, uint32_t snap_cap,
                  uint32_t *snap_len)
{
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
    if (o.n_nearest > 16) return fail(c, TL_ERR_UNSUPPORTED, "tl_lk: n_nearest=%u > 16", o.n_nearest);
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
    const bool multi_cu = !(c->flags & TL_FLAG_LK_ONE_WORKGROUP) && !lk_small;
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
    const size_t total = o_sub + (keep_sub ? up(sub_b) : 0);
    // every mode / size check and every allocation comes before the first event record and the first enqueue: a rejected call
    // leaves the previous kernel sequence's event pair intact and nothing in flight
    // (the single-workgroup forms keep no snapshots on the device: a trace of theirs is the final best tour alone, below)
    const bool snap_dev = snap_pos && multi_cu;
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
    G.chip_step = (fused_pick && levels == 3u && n >= 1500u && !(tf & TL_FLAG_LK_SEPARATE_STEP)) ? 1u : 0u;
    if (snap_dev) {
        G.snap = (uint32_t *)c->out_pos.p;
        G.snap_dist = (float *)c->out_stats.p;
        G.snap_cap = snap_cap;
    }
    if (split_scan) HIPCHK(c, hipMemsetAsync(G.pairmin, 0xFF, (size_t)2 * n * 4, c->stream));
    uint64_t cnt[4] = {0, 0, 0, 0};
    if (!multi_cu) {
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
            if (hs.finished) break;
        }
        if (gexec) (void)hipGraphExecDestroy(gexec);
        if (graph) (void)hipGraphDestroy(graph);
        if (rc_loop != TL_OK) return rc_loop;
        cnt[0] = hs.scans;
        cnt[1] = hs.searches;
        cnt[2] = hs.moves;
        cnt[3] = hs.exchanged;
        if (snap_pos) {
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
    if (!multi_cu) HIPCHK(c, hipMemcpyAsync(cnt, G.counters, 32, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(&cost, c->out_cost.p, 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (out_cost) *out_cost = cost;
    if (snap_pos && !multi_cu) {
        // TL_FLAG_LK_ONE_WORKGROUP: the reference's last PathUpdate only — the final best tour with its best_dist, the Euclidean
        // tour_distance of lin_kernighan.rs:118-122 (edges in tour order, the closing edge last; f32, as KDPoint::distance)
        float bd = 0.0f;
        for (uint32_t q = 0; q < n; ++q) {
            const float *p0 = xy + 2 * (size_t)out_pos[q], *p1 = xy + 2 * (size_t)out_pos[(q + 1u) % n];
            const float dx = p0[0] - p1[0], dy = p0[1] - p1[1];
            const float sq = dx * dx + dy * dy;  // -ffp-contract=off: three roundings
            bd += sqrtf(sq);
        }
        if (snap_len) *snap_len = 1;
        if (snap_cap) {
            memcpy(snap_pos, out_pos, (size_t)n * 4);
            if (snap_dist) snap_dist[0] = bd;
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
