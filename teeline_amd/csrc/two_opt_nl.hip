// two_opt_nl.hip — neighbour lists of an instance for the late sweeps of the LDS 2-opt descent (two_opt_ref.hip, "NL rows").
//
// The reference decides candidate (i, j) of row (a, b) = (p[i], p[i+1]) by  d(a,c) + d(b,e) < d(a,b) + d(c,e)  with c = p[j],
// e = p[j+1] (src/tsp/two_opt.rs:35-49).  In f32 that implies  sq(a,c) < sq(a,b)  or  sq(b,e) < sq(c,e)  (the kernel's L1 bound).
// So the improving candidates of a row all lie in
//   A = { j : c = p[j] is strictly closer to a than b is }            -> c is among the KA nearest cities of a  whenever sq(a,b) <= the
//                                                                        KA-th smallest squared distance from a     (record words 0, 4..19)
//   B = { j : b is strictly closer to e = p[j+1] than c = p[j] is }   -> b is among the KB nearest cities of e, i.e. e is in b's REVERSE
//                                                                        list, whenever sq(c,e) <= the KB-th smallest squared distance
//                                                                        from e                          (record words 2, 20..55; dkb2)
// (a city strictly closer than the K-th nearest is in every K-nearest set, ties or not; squared distances are the kernel's own
// sqdist, which is symmetric bit for bit).  Cities e with a tour edge longer than their KB-th distance are kept by the descent in a
// short "long" list and offered to every row; rows with sq(a,b) beyond a's KA-th distance, or with a b that more than kNlRB cities
// count among their nearest, take the tile path.  The lists depend on the coordinates only: built once per call as one 128-byte
// record per city (tl_kernels.h), shared by every descent of a batch (L2-resident: 1.3 MB at n = 10^4).
#include "tl_kernels.h"

#pragma clang fp contract(off)

namespace tl {

namespace {

constexpr int kKnnThreads = 256;
constexpr int kHistBins = 2048;

__device__ __forceinline__ uint32_t block_exclusive_scan_256(uint32_t v, uint32_t *sh4, int tid, uint32_t *total)
{
    // inclusive scan inside the wave, then the four wave totals through LDS
    uint32_t x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t y = (uint32_t)__shfl_up((int)x, o);
        if ((tid & 63) >= o) x += y;
    }
    if ((tid & 63) == 63) sh4[tid >> 6] = x;
    __syncthreads();
    uint32_t off = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const uint32_t t = sh4[w];
        if (w < (tid >> 6)) off += t;
        tot += t;
    }
    __syncthreads();
    *total = tot;
    return off + x - v;
}

// K-th smallest (1-based) of d[0..n) by a three-digit radix select (11 + 11 + 10 bits).  Returns the value; *quota = how many
// elements EQUAL to it belong to the K smallest (the others among them are strictly smaller).
// `cap`: elements above it are known not to be among the K smallest and stay out of the histograms (0xFFFFFFFF: none known).
__device__ __forceinline__ uint32_t radix_select(const uint32_t *d, uint32_t n, uint32_t K, uint32_t *hist, uint32_t *misc, int tid, uint32_t *quota,
                                                 uint32_t cap = 0xFFFFFFFFu)
{
    uint32_t prefix = 0, pmask = 0, need = K;
    const int shifts[3] = {21, 10, 0}, bits[3] = {11, 11, 10};
    for (int pass = 0; pass < 3; ++pass) {
        const int sh = shifts[pass];
        const uint32_t nb = 1u << bits[pass];
        for (uint32_t b = (uint32_t)tid; b < (uint32_t)kHistBins; b += kKnnThreads) hist[b] = 0;
        __syncthreads();
        for (uint32_t v = (uint32_t)tid; v < n; v += kKnnThreads) {
            const uint32_t x = d[v];
            if (x <= cap && (x & pmask) == prefix) atomicAdd(&hist[(x >> sh) & (nb - 1u)], 1u);
        }
        __syncthreads();
        // thread t owns bins [8t, 8t+8): exclusive scan of the owners' sums, then the owner of the K-th walks its bins
        uint32_t own = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) own += hist[tid * 8 + q];
        uint32_t total;
        const uint32_t before = block_exclusive_scan_256(own, misc, tid, &total);
        if (before < need && need <= before + own) {
            uint32_t cum = before;
            for (int q = 0; q < 8; ++q) {
                const uint32_t h = hist[tid * 8 + q];
                if (need <= cum + h) {
                    misc[8] = (uint32_t)(tid * 8 + q);
                    misc[9] = need - cum;
                    break;
                }
                cum += h;
            }
        }
        __syncthreads();
        prefix |= misc[8] << sh;
        need = misc[9];
        pmask |= (nb - 1u) << sh;
        __syncthreads();
    }
    *quota = need;
    return prefix;
}

// the K smallest of d[] (threshold T, `quota` of the elements equal to T: the ones with the lowest indices), written in index order
// (rcnt / rec / u: the list is a KB list — city u is entered in the reverse list of every city of it)
__device__ __forceinline__ void emit_list(const uint32_t *d, uint32_t n, uint32_t T, uint32_t quota, uint16_t *out, uint32_t *rcnt, uint16_t *rec, uint32_t u,
                                          uint32_t *misc, int tid)
{
    const uint32_t chunk = ((n + kKnnThreads - 1) / kKnnThreads) | 1u;  // (odd: a thread's run of elements starts in its own LDS bank)
    const uint32_t v0 = (uint32_t)tid * chunk < n ? (uint32_t)tid * chunk : n, v1 = v0 + chunk < n ? v0 + chunk : n;
    uint32_t less = 0, tie = 0;
    for (uint32_t v = v0; v < v1; ++v) {
        const uint32_t x = d[v];
        less += x < T ? 1u : 0u;
        tie += x == T ? 1u : 0u;
    }
    uint32_t tot;
    const uint32_t before = block_exclusive_scan_256(less | (tie << 16), misc, tid, &tot);
    uint32_t lb = before & 0xFFFFu, tb = before >> 16;
    for (uint32_t v = v0; v < v1; ++v) {
        const uint32_t x = d[v];
        if (x < T || (x == T && tb < quota)) {
            out[lb + (tb < quota ? tb : quota)] = (uint16_t)v;
            if (rcnt) {
                const uint32_t slot = atomicAdd(&rcnt[v], 1u);
                if (slot < (uint32_t)kNlRB) rec[(size_t)v * 64u + (uint32_t)kNlRecB0 + slot] = (uint16_t)u;
            }
        }
        lb += x < T ? 1u : 0u;
        tb += x == T ? 1u : 0u;
    }
}

// one workgroup per city u: squared distances to every other city as bit patterns (squares are >= +0: their bits order like
// unsigned ints, NaNs last), the KB-th and KA-th smallest, the two lists
// The lists are kept in the context's workspace together with the coordinates they were built from.  A later call compares its
// coordinates with that copy, bit for bit, on the device (k_nl_check: state[0] = 1 where everything is equal) and every kernel
// of the build returns at once where the lists are still the instance's — a multi-start loop builds them once, not per batch.
// state[0]: lists valid for this call's xy; state[1]: the n they were built for.
__global__ __launch_bounds__(256) void k_nl_check(const float2 *__restrict__ xy, uint32_t n, const uint2 *__restrict__ kept, uint32_t *__restrict__ state)
{
    // (launched with state[0] preset to 1 by k_nl_begin; any difference clears it)
    const uint32_t v = blockIdx.x * 256u + threadIdx.x;
    if (v >= n) return;
    const float2 p = xy[v];
    const uint2 q = kept[v];
    if (__builtin_bit_cast(uint32_t, p.x) != q.x || __builtin_bit_cast(uint32_t, p.y) != q.y) state[0] = 0u;
}
__global__ void k_nl_begin(uint32_t n, uint32_t *__restrict__ state)
{
    state[0] = state[1] == n ? 1u : 0u;
}
__global__ __launch_bounds__(256) void k_nl_clear(uint32_t n, uint32_t *__restrict__ rcnt, uint32_t *__restrict__ rec32, const uint32_t *__restrict__ state)
{
    if (state[0]) return;
    const uint32_t g = blockIdx.x * 256u + threadIdx.x;
    if (g < n) rcnt[g] = 0u;
    if (g < n * 32u) rec32[g] = 0xFFFFFFFFu;  // (0xFFFF: an empty slot)
}
__global__ __launch_bounds__(256) void k_nl_commit(const float2 *__restrict__ xy, uint32_t n, uint2 *__restrict__ kept, uint32_t *__restrict__ state)
{
    if (state[0]) return;
    const uint32_t v = blockIdx.x * 256u + threadIdx.x;
    if (v < n) {
        const float2 p = xy[v];
        kept[v] = make_uint2(__builtin_bit_cast(uint32_t, p.x), __builtin_bit_cast(uint32_t, p.y));
    }
    if (v == 0) state[1] = n;  // (state[0] stays 0 until the next call's k_nl_begin: the kernels behind this one still see "rebuilt")
}

__global__ __launch_bounds__(kKnnThreads) void k_nl_knn(const float2 *__restrict__ xy, uint32_t n, uint32_t ka, uint32_t kb, uint16_t *__restrict__ rec,
                                                        uint16_t *__restrict__ knn_b, uint32_t *__restrict__ dkb2, uint32_t *__restrict__ rcnt,
                                                        const uint32_t *__restrict__ state)
{
    if (state[0]) return;
    extern __shared__ __attribute__((aligned(16))) uint32_t sm[];
    uint32_t *d = sm;
    uint32_t *hist = sm + ((n + 63u) & ~63u);
    uint32_t *misc = hist + kHistBins;
    const int tid = threadIdx.x;
    const uint32_t u = blockIdx.x;
    const float2 pu = xy[u];
    uint32_t mine = 0xFFFFFFFFu;
    for (uint32_t v = (uint32_t)tid; v < n; v += kKnnThreads) {
        const uint32_t x = v == u ? 0xFFFFFFFFu : __builtin_bit_cast(uint32_t, sqdist(pu, xy[v]));
        d[v] = x;
        mine = x < mine ? x : mine;
    }
    // An upper bound of the KB-th smallest distance, so that the selects below histogram a few dozen elements instead of n (10^4
    // LDS atomics on a handful of exponent bins were most of this kernel): each thread's smallest element is one of n's, so the
    // KB-th smallest of the 256 thread minima has at least KB elements at or below it.
    uint32_t *mins = misc + 16;
    mins[tid] = mine;
    __syncthreads();
    uint32_t qb, qa;
    const uint32_t cap = kb <= (uint32_t)kKnnThreads ? radix_select(mins, kKnnThreads, kb, hist, misc, tid, &qb) : 0xFFFFFFFFu;
    const uint32_t tb = radix_select(d, n, kb, hist, misc, tid, &qb, cap);
    emit_list(d, n, tb, qb, knn_b + (size_t)u * kb, rcnt, rec, u, misc, tid);
    const uint32_t ta = radix_select(d, n, ka, hist, misc, tid, &qa, cap);
    emit_list(d, n, ta, qa, rec + (size_t)u * 64u + (uint32_t)kNlRecA0, nullptr, nullptr, u, misc, tid);
    if (tid == 0) {
        dkb2[u] = tb;
        rec[(size_t)u * 64u + 0u] = (uint16_t)(ta >> 16);  // (rounded down: a row is listed only if sq(a, b)'s high half is BELOW it)
    }
}

// "reverse list incomplete" into the records of the cities that more than kNlRB others count among their nearest
__global__ __launch_bounds__(256) void k_nl_counts(const uint32_t *__restrict__ rcnt, uint32_t n, uint16_t *__restrict__ rec, const uint32_t *__restrict__ state)
{
    if (state[0]) return;
    const uint32_t v = blockIdx.x * 256u + threadIdx.x;
    if (v >= n) return;
    rec[(size_t)v * 64u + 2u] = rcnt[v] > (uint32_t)kNlRB ? 1u : 0u;
}


// ---- the same lists, one WAVE per city (the form that runs; k_nl_knn above is its cross-check, tl_two_opt_neighbour_lists(form = 1)).
// No block barrier, no distance array: (1) every lane keeps the smallest distance it meets; the KB-th smallest of the 64 lane minima
// is an upper bound `cap` of the KB-th smallest of all (each lane's minimum is one of the n - 1 distances); (2) the elements at or
// below cap — a few dozen — are collected in index order; (3) their ranks by (distance, index) give both lists and both K-th
// distances.  Where more than kNlBuf elements lie at or below cap (masses of equal distances) the K-th distances come from a
// bisection on the bit pattern instead (32 counting passes each), and the lists from one more pass each.
constexpr uint32_t kNlBuf = 256;

__device__ __forceinline__ uint32_t nl_dist_bits(const float2 pu, const float2 *__restrict__ xy, uint32_t u, uint32_t v, uint32_t n)
{
    return (v < n && v != u) ? __builtin_bit_cast(uint32_t, sqdist(pu, xy[v < n ? v : 0u])) : 0xFFFFFFFFu;
}

// K-th smallest distance of city u by bisection on the bits: the smallest T with #{x <= T} >= K; *quota = K - #{x < T}
__device__ uint32_t nl_bisect(const float2 pu, const float2 *__restrict__ xy, uint32_t u, uint32_t n, uint32_t K, int lane, uint32_t *quota)
{
    uint32_t lo = 0u, hi = 0xFFFFFFFEu;  // invariant: count(x <= hi) >= K (n - 1 >= K real elements), count(x <= lo - 1) < K
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        uint32_t c = 0;
        for (uint32_t base = 0; base < n; base += 64u)
            c += (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(nl_dist_bits(pu, xy, u, base + (uint32_t)lane, n) <= mid));
        if (c >= K) hi = mid;
        else lo = mid + 1u;
    }
    uint32_t less = 0;
    for (uint32_t base = 0; base < n; base += 64u)
        less += (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(nl_dist_bits(pu, xy, u, base + (uint32_t)lane, n) < lo));
    *quota = K - less;
    return lo;
}

// the list "x < T, or x == T among the first `quota` such in index order", written in index order (one pass over the cities)
__device__ void nl_emit_pass(const float2 pu, const float2 *__restrict__ xy, uint32_t u, uint32_t n, uint32_t T, uint32_t quota, uint16_t *__restrict__ out,
                             uint32_t *__restrict__ rcnt, uint16_t *__restrict__ rec, int lane)
{
    uint32_t at = 0, ties = 0;
    for (uint32_t base = 0; base < n; base += 64u) {
        const uint32_t v = base + (uint32_t)lane;
        const uint32_t x = nl_dist_bits(pu, xy, u, v, n);
        const uint64_t mt = __builtin_amdgcn_ballot_w64(x == T);
        const uint32_t trank = ties + __builtin_amdgcn_mbcnt_hi((uint32_t)(mt >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mt, 0u));
        const bool sel = x < T || (x == T && trank < quota);
        const uint64_t ms = __builtin_amdgcn_ballot_w64(sel);
        if (sel) {
            out[at + __builtin_amdgcn_mbcnt_hi((uint32_t)(ms >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ms, 0u))] = (uint16_t)v;
            if (rcnt) {
                const uint32_t slot = atomicAdd(&rcnt[v], 1u);
                if (slot < (uint32_t)kNlRB) rec[(size_t)v * 64u + (uint32_t)kNlRecB0 + slot] = (uint16_t)u;
            }
        }
        at += (uint32_t)__builtin_popcountll(ms);
        ties += (uint32_t)__builtin_popcountll(mt);
    }
}

__global__ __launch_bounds__(256) void k_nl_knn_wave(const float2 *__restrict__ xy, uint32_t n, uint32_t ka, uint32_t kb, uint16_t *__restrict__ rec,
                                                     uint16_t *__restrict__ knn_b, uint32_t *__restrict__ dkb2, uint32_t *__restrict__ rcnt,
                                                     const uint32_t *__restrict__ state)
{
    if (state[0]) return;
    __shared__ uint32_t s_bits[4][kNlBuf];
    __shared__ uint16_t s_idx[4][kNlBuf];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t u = blockIdx.x * 4u + (uint32_t)wave;
    if (u >= n) return;
    const float2 pu = xy[u];
    uint16_t *__restrict__ out_a = rec + (size_t)u * 64u + (uint32_t)kNlRecA0, *__restrict__ out_b = knn_b + (size_t)u * kb;
    // (1) lane minima, their KB-th smallest
    uint32_t mine = 0xFFFFFFFFu;
    for (uint32_t base = 0; base < n; base += 64u) {
        const uint32_t x = nl_dist_bits(pu, xy, u, base + (uint32_t)lane, n);
        mine = x < mine ? x : mine;
    }
    uint32_t rank = 0;
    for (int m = 0; m < 64; ++m) {
        const uint32_t y = (uint32_t)__builtin_amdgcn_readlane((int)mine, m);
        rank += (y < mine || (y == mine && m < lane)) ? 1u : 0u;
    }
    const uint64_t mk = __builtin_amdgcn_ballot_w64(rank == kb - 1u);  // (kb <= 64; exactly one lane)
    const uint32_t cap = (uint32_t)__builtin_amdgcn_readlane((int)mine, __builtin_ffsll((long long)mk) - 1);
    // (2) the elements at or below cap, in index order
    uint32_t cnt = 0;
    for (uint32_t base = 0; base < n; base += 64u) {
        const uint32_t v = base + (uint32_t)lane;
        const uint32_t x = nl_dist_bits(pu, xy, u, v, n);
        const bool sel = x <= cap && x != 0xFFFFFFFFu;
        const uint64_t m = __builtin_amdgcn_ballot_w64(sel);
        const uint32_t pos = cnt + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        if (sel && pos < kNlBuf) {
            s_bits[wave][pos] = x;
            s_idx[wave][pos] = (uint16_t)v;
        }
        cnt += (uint32_t)__builtin_popcountll(m);
    }
    uint32_t tb, ta;
    if (cnt <= kNlBuf && cnt >= kb) {
        // (3) ranks by (distance, position in the buffer = index order); a lane holds entries lane, lane + 64, ...
        uint32_t mb[kNlBuf / 64], mr[kNlBuf / 64];
#pragma unroll
        for (int q = 0; q < (int)(kNlBuf / 64); ++q) {
            const uint32_t j = (uint32_t)q * 64u + (uint32_t)lane;
            mb[q] = j < cnt ? s_bits[wave][j] : 0xFFFFFFFFu;
            mr[q] = 0;
        }
        for (uint32_t i = 0; i < cnt; ++i) {
            const uint32_t b = s_bits[wave][i];  // (one address for the whole wave: a broadcast read)
#pragma unroll
            for (int q = 0; q < (int)(kNlBuf / 64); ++q) {
                const uint32_t j = (uint32_t)q * 64u + (uint32_t)lane;
                mr[q] += (b < mb[q] || (b == mb[q] && i < j)) ? 1u : 0u;
            }
        }
        uint32_t at_a = 0, at_b = 0;
        tb = 0;
        ta = 0;
#pragma unroll
        for (int q = 0; q < (int)(kNlBuf / 64); ++q) {
            const uint32_t j = (uint32_t)q * 64u + (uint32_t)lane;
            const bool in = j < cnt;
            const bool sb = in && mr[q] < kb, sa = in && mr[q] < ka;
            const uint64_t m_b = __builtin_amdgcn_ballot_w64(sb), m_a = __builtin_amdgcn_ballot_w64(sa);
            const uint32_t v = in ? s_idx[wave][j] : 0u;
            if (sb) {
                out_b[at_b + __builtin_amdgcn_mbcnt_hi((uint32_t)(m_b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m_b, 0u))] = (uint16_t)v;
                const uint32_t slot = atomicAdd(&rcnt[v], 1u);
                if (slot < (uint32_t)kNlRB) rec[(size_t)v * 64u + (uint32_t)kNlRecB0 + slot] = (uint16_t)u;
            }
            if (sa) out_a[at_a + __builtin_amdgcn_mbcnt_hi((uint32_t)(m_a >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m_a, 0u))] = (uint16_t)v;
            at_b += (uint32_t)__builtin_popcountll(m_b);
            at_a += (uint32_t)__builtin_popcountll(m_a);
            // the K-th distances: the entries of rank kb - 1 and ka - 1
            const uint64_t kbm = __builtin_amdgcn_ballot_w64(in && mr[q] == kb - 1u), kam = __builtin_amdgcn_ballot_w64(in && mr[q] == ka - 1u);
            if (kbm) tb = (uint32_t)__builtin_amdgcn_readlane((int)mb[q], __builtin_ffsll((long long)kbm) - 1);
            if (kam) ta = (uint32_t)__builtin_amdgcn_readlane((int)mb[q], __builtin_ffsll((long long)kam) - 1);
        }
    } else {
        uint32_t qb, qa;
        tb = nl_bisect(pu, xy, u, n, kb, lane, &qb);
        nl_emit_pass(pu, xy, u, n, tb, qb, out_b, rcnt, rec, lane);
        ta = nl_bisect(pu, xy, u, n, ka, lane, &qa);
        nl_emit_pass(pu, xy, u, n, ta, qa, out_a, nullptr, nullptr, lane);
    }
    if (lane == 0) {
        dkb2[u] = tb;
        rec[(size_t)u * 64u + 0u] = (uint16_t)(ta >> 16);  // (rounded down: a row is listed only if sq(a, b)'s high half is BELOW it)
    }
}

constexpr size_t al256(size_t b) { return (b + 255) & ~(size_t)255; }

}  // namespace

size_t two_opt_nl_ws_bytes(uint32_t n)
{
    return al256((size_t)n * 128) + al256((size_t)n * kNlKB * 2) + 2 * al256((size_t)n * 4) + al256((size_t)n * 8) + 256;
}

// Lays the lists out in `ws` (two_opt_nl_ws_bytes; `fresh`: the buffer has just been (re)allocated and holds nothing), builds them on
// `s` unless they are already there for these very coordinates, returns the pointers the descent kernel reads.
hipError_t launch_two_opt_nl_build(const float2 *xy, uint32_t n, void *ws, bool fresh, TwoOptNl *out, hipStream_t s, int form)
{
    static_assert(kNlRecA0 + kNlKA == kNlRecB0 && kNlRecB0 + kNlRB == kNlSurv0 && kNlSurv0 + kNlSurvSlots == 64, "record layout");
    const uint32_t ka = kNlKA, kb = kNlKB;
    if (n <= kb + 1u || n > 65535u) return hipErrorInvalidValue;
    // (the state block first: its place does not depend on n)
    unsigned char *p = (unsigned char *)ws;
    uint32_t *state = (uint32_t *)p;
    p += 256;
    uint16_t *rec = (uint16_t *)p;
    p += al256((size_t)n * 128);
    uint16_t *knn_b = (uint16_t *)p;
    p += al256((size_t)n * kb * 2);
    uint32_t *dkb2 = (uint32_t *)p;
    p += al256((size_t)n * 4);
    uint32_t *rcnt = (uint32_t *)p;
    p += al256((size_t)n * 4);
    uint2 *kept = (uint2 *)p;
    hipError_t e;
    if (fresh && (e = hipMemsetAsync(state, 0, 256, s)) != hipSuccess) return e;  // state[1] = 0: no n matches
    const size_t lds = ((size_t)((n + 63u) & ~63u) + kHistBins + 16 + kKnnThreads) * 4;
    e = allow_max_lds(reinterpret_cast<const void *>(k_nl_knn));
    if (e != hipSuccess) return e;
    const dim3 per_city((n + 255u) / 256u);
    hipLaunchKernelGGL(k_nl_begin, dim3(1), dim3(1), 0, s, n, state);
    hipLaunchKernelGGL(k_nl_check, per_city, dim3(256), 0, s, xy, n, kept, state);
    hipLaunchKernelGGL(k_nl_clear, dim3((n * 32u + 255u) / 256u), dim3(256), 0, s, n, rcnt, (uint32_t *)rec, state);
    if (form == 1) hipLaunchKernelGGL(k_nl_knn, dim3(n), dim3(kKnnThreads), lds, s, xy, n, ka, kb, rec, knn_b, dkb2, rcnt, state);  // (the cross-check: one workgroup per city)
    else hipLaunchKernelGGL(k_nl_knn_wave, dim3((n + 3u) / 4u), dim3(256), 0, s, xy, n, ka, kb, rec, knn_b, dkb2, rcnt, state);
    hipLaunchKernelGGL(k_nl_counts, per_city, dim3(256), 0, s, rcnt, n, rec, state);
    hipLaunchKernelGGL(k_nl_commit, per_city, dim3(256), 0, s, xy, n, kept, state);
    out->rec = rec;
    out->dkb2 = dkb2;
    out->knn_b = knn_b;
    out->rcnt = rcnt;
    return hipGetLastError();
}

}  // namespace tl
