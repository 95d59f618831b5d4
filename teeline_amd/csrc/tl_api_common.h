// tl_api_common.h — what the host-side translation units of the C ABI share (internal to libteeline_gpu): the context, its
// single-thread guard, error reporting and the grow-only device buffers.  tl_api.hip (context, matrix, tour length),
// tl_api_two_opt.hip, tl_api_scans.hip (3-opt, Or-opt) and tl_api_lk.hip (candidate lists, NN seed, Lin-Kernighan) include it.
#pragma once
#include "../../include/teeline_gpu.h"
#include "tl_kernels.h"

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

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
    // The context's device workspace shared by asynchronous calls (neighbour lists `nl`, grid coordinates `fx`) belongs to the LAST
    // kernel sequence enqueued with it.  tl_two_opt_batch_dev returns before that sequence has run and takes the caller's stream, so a
    // later call on ANOTHER stream (or on the context's own) must be ordered behind it before it rebuilds that workspace:
    // ev_ws is recorded behind every such sequence on ws_stream; ws_order() makes a different stream wait on it.
    hipEvent_t ev_ws = nullptr;
    hipStream_t ws_stream = nullptr;
    bool ws_pending = false;
    bool in_callback = false;  // a tl_lk_live callback of this context is running: any entry into this context from inside it is refused
    int cus = 0, lds_bytes = 0;
    std::string arch;
    std::string err;
    DevBuf xy, dm, init, out_pos, out_cost, out_stats, misc, work, dmfull, kd, fx, nl, dmx;  // dmx: matrix-form 2-opt, the descents' per-city records of their late sweeps
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
            if (c->in_callback) {  // re-entry from inside a live-progress callback: the running search owns stream and workspace
                ok = false;
                return;
            }
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


#define HIPCHK(c, expr)                                                                           \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess) {                                                                   \
            (void)hipGetLastError(); /* the thread's sticky error: a later launch check must not report this one again */ \
            return fail((c), _e == hipErrorOutOfMemory ? TL_ERR_NOMEM : TL_ERR_HIP, "%s: %s", #expr, \
                        hipGetErrorString(_e));                                                   \
        }                                                                                         \
    } while (0)

namespace tlapi {
int fail(tl_ctx *c, int code, const char *fmt, ...);            // sets the context's (or the thread's create-) error string, returns code
int ensure(tl_ctx *c, DevBuf &b, size_t bytes);                // grow-only device buffer
bool is_permutation(const uint32_t *p, uint32_t n);            // validate_tour, src/tsp/mod.rs:1620-1634
uint32_t lds_max_n(int lds_bytes);                             // largest n of the LDS-resident 2-opt descent
int ws_order(tl_ctx *c, hipStream_t s);                        // before touching nl / fx on stream s: wait for their last user on another stream
int ws_mark(tl_ctx *c, hipStream_t s);                         // behind a kernel sequence that reads nl / fx on stream s
// A tune-only create flag: always clear in the product build (tl_create refuses them), so the branches it selects fold away.
#ifdef TL_TUNE
inline uint32_t tune_flags(const tl_ctx *c) { return c->flags; }
#else
inline uint32_t tune_flags(const tl_ctx *) { return 0u; }
#endif
inline int knn_form(const tl_ctx *c) { return (tune_flags(c) & TL_FLAG_KNN_1LANE) ? 1 : (tune_flags(c) & TL_FLAG_KNN_4LANES) ? 4 : 0; }
// tl_api_lk.hip: device-side candidate lists and NN seed, also used by tl_lk
int build_candidates_dev(tl_ctx *c, const float *xy_host, const float2 *d_xy, uint32_t n, uint32_t k, uint32_t *d_cand);
int nn_seed_dev(tl_ctx *c, const float2 *d_xy, uint32_t n, uint32_t n_nearest, uint32_t *d_path);
}  // namespace tlapi
