// tl_api.hip — host side of the C ABI declared in include/teeline_gpu.h: context, device matrix, tour length.
// (2-opt: tl_api_two_opt.hip; 3-opt / Or-opt: tl_api_scans.hip; candidate lists, NN seed, Lin-Kernighan: tl_api_lk.hip.)
//
// A tl_ctx owns one HIP stream, a grow-only device workspace and a pair of HIP events; every entry
// point validates its arguments on the host (shapes, permutation validity, size limits of the
// kernel it is about to launch) before anything reaches the GPU.  There is no CPU fallback.
#include "tl_api_common.h"

using namespace tl;
using namespace tlapi;

static thread_local std::string g_create_err;  // tl_last_error(NULL): the calling thread's last tl_create failure

namespace tlapi {
int fail(tl_ctx *c, int code, const char *fmt, ...)
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


int ensure(tl_ctx *c, DevBuf &b, size_t bytes)
{
    if (bytes <= b.cap) return TL_OK;
    if (b.p) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (c->ws_pending) HIPCHK(c, hipEventSynchronize(c->ev_ws));  // an asynchronous batch on a caller's stream may still read it
        HIPCHK(c, hipFree(b.p));
        b.p = nullptr;
        b.cap = 0;
    }
    size_t cap = bytes + bytes / 8 + 256;
    HIPCHK(c, hipMalloc(&b.p, cap));
    b.cap = cap;
    return TL_OK;
}


int ws_order(tl_ctx *c, hipStream_t s)
{
    if (c->ws_pending && c->ws_stream != s) HIPCHK(c, hipStreamWaitEvent(s, c->ev_ws, 0));
    return TL_OK;
}

int ws_mark(tl_ctx *c, hipStream_t s)
{
    HIPCHK(c, hipEventRecord(c->ev_ws, s));
    c->ws_stream = s;
    c->ws_pending = true;
    return TL_OK;
}


bool is_permutation(const uint32_t *p, uint32_t n)  // validate_tour, src/tsp/mod.rs:1620-1634
{
    std::vector<unsigned char> seen(n, 0);
    for (uint32_t i = 0; i < n; ++i) {
        if (p[i] >= n || seen[p[i]]) return false;
        seen[p[i]] = 1;
    }
    return true;
}


uint32_t lds_max_n(int lds_bytes)
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
}  // namespace tlapi

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
        (e = hipEventCreate(&c->ev0)) != hipSuccess || (e = hipEventCreate(&c->ev1)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&c->ev_ws, hipEventDisableTiming)) != hipSuccess) {
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
    if (c->ws_pending) (void)hipEventSynchronize(c->ev_ws);
    for (DevBuf *b : {&c->xy, &c->dm, &c->init, &c->out_pos, &c->out_cost, &c->out_stats, &c->misc, &c->work, &c->dmfull, &c->kd, &c->fx, &c->nl, &c->dmx})
        if (b->p) (void)hipFree(b->p);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->ev_ws) (void)hipEventDestroy(c->ev_ws);
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

