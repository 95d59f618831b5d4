// two_opt_nl.hip — neighbour lists of an instance for the late sweeps of the LDS 2-opt descent (two_opt_ref.hip, "NL rows").
//
// The reference decides candidate (i, j) of row (a, b) = (p[i], p[i+1]) by  d(a,c) + d(b,e) < d(a,b) + d(c,e)  with c = p[j],
// e = p[j+1] (src/tsp/two_opt.rs:35-49).  In f32 that implies  sq(a,c) < sq(a,b)  or  sq(b,e) < sq(c,e)  (the kernel's L1 bound).
// So the improving candidates of a row all lie in
//   A = { j : c = p[j] is strictly closer to a than b is }            -> c is among the KA nearest cities of a  whenever sq(a,b) <= the
//                                                                        KA-th smallest squared distance from a            (knn_a, dka2)
//   B = { j : b is strictly closer to e = p[j+1] than c = p[j] is }   -> b is among the KB nearest cities of e, i.e. e is in b's REVERSE
//                                                                        list, whenever sq(c,e) <= the KB-th smallest squared distance
//                                                                        from e                                            (rl, rknn, dkb2)
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
__device__ __forceinline__ uint32_t radix_select(const uint32_t *d, uint32_t n, uint32_t K, uint32_t *hist, uint32_t *misc, int tid, uint32_t *quota)
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
            if ((x & pmask) == prefix) atomicAdd(&hist[(x >> sh) & (nb - 1u)], 1u);
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
    const uint32_t chunk = (n + kKnnThreads - 1) / kKnnThreads;
    const uint32_t v0 = (uint32_t)tid * chunk, v1 = v0 + chunk < n ? v0 + chunk : n;
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
__global__ __launch_bounds__(kKnnThreads) void k_nl_knn(const float2 *__restrict__ xy, uint32_t n, uint32_t ka, uint32_t kb, uint16_t *__restrict__ rec,
                                                        uint16_t *__restrict__ knn_b, uint32_t *__restrict__ dkb2, uint32_t *__restrict__ rcnt)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t sm[];
    uint32_t *d = sm;
    uint32_t *hist = sm + ((n + 63u) & ~63u);
    uint32_t *misc = hist + kHistBins;
    const int tid = threadIdx.x;
    const uint32_t u = blockIdx.x;
    const float2 pu = xy[u];
    for (uint32_t v = (uint32_t)tid; v < n; v += kKnnThreads) d[v] = v == u ? 0xFFFFFFFFu : __builtin_bit_cast(uint32_t, sqdist(pu, xy[v]));
    __syncthreads();
    uint32_t qb, qa;
    const uint32_t tb = radix_select(d, n, kb, hist, misc, tid, &qb);
    emit_list(d, n, tb, qb, knn_b + (size_t)u * kb, rcnt, rec, u, misc, tid);
    const uint32_t ta = radix_select(d, n, ka, hist, misc, tid, &qa);
    emit_list(d, n, ta, qa, rec + (size_t)u * 64u + (uint32_t)kNlRecA0, nullptr, nullptr, u, misc, tid);
    if (tid == 0) {
        dkb2[u] = tb;
        rec[(size_t)u * 64u + 0u] = (uint16_t)(ta >> 16);  // (rounded down: a row is listed only if sq(a, b)'s high half is BELOW it)
    }
}

// "reverse list incomplete" into the records of the cities that more than kNlRB others count among their nearest
__global__ __launch_bounds__(256) void k_nl_counts(const uint32_t *__restrict__ rcnt, uint32_t n, uint16_t *__restrict__ rec)
{
    const uint32_t v = blockIdx.x * 256u + threadIdx.x;
    if (v >= n) return;
    rec[(size_t)v * 64u + 2u] = rcnt[v] > (uint32_t)kNlRB ? 1u : 0u;
}

constexpr size_t al256(size_t b) { return (b + 255) & ~(size_t)255; }

}  // namespace

size_t two_opt_nl_ws_bytes(uint32_t n)
{
    return al256((size_t)n * 128) + al256((size_t)n * kNlKB * 2) + 2 * al256((size_t)n * 4);
}

// lays the lists out in `ws` (two_opt_nl_ws_bytes), builds them on `s`, returns the pointers the descent kernel reads
hipError_t launch_two_opt_nl_build(const float2 *xy, uint32_t n, void *ws, TwoOptNl *out, hipStream_t s)
{
    static_assert(kNlRecA0 + kNlKA == kNlRecB0 && kNlRecB0 + kNlRB == kNlSurv0 && kNlSurv0 + kNlSurvSlots == 64, "record layout");
    const uint32_t ka = kNlKA, kb = kNlKB;
    if (n <= kb + 1u || n > 65535u) return hipErrorInvalidValue;
    unsigned char *p = (unsigned char *)ws;
    uint16_t *rec = (uint16_t *)p;
    p += al256((size_t)n * 128);
    uint16_t *knn_b = (uint16_t *)p;
    p += al256((size_t)n * kb * 2);
    uint32_t *dkb2 = (uint32_t *)p;
    p += al256((size_t)n * 4);
    uint32_t *rcnt = (uint32_t *)p;
    hipError_t e = hipMemsetAsync(rcnt, 0, (size_t)n * 4, s);
    if (e != hipSuccess) return e;
    if ((e = hipMemsetAsync(rec, 0xFF, (size_t)n * 128, s)) != hipSuccess) return e;  // (0xFFFF: an empty slot)
    const size_t lds = ((size_t)((n + 63u) & ~63u) + kHistBins + 16) * 4;
    e = allow_max_lds(reinterpret_cast<const void *>(k_nl_knn));
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_nl_knn, dim3(n), dim3(kKnnThreads), lds, s, xy, n, ka, kb, rec, knn_b, dkb2, rcnt);
    hipLaunchKernelGGL(k_nl_counts, dim3((n + 255u) / 256u), dim3(256), 0, s, rcnt, n, rec);
    out->rec = rec;
    out->dkb2 = dkb2;
    out->knn_b = knn_b;
    out->rcnt = rcnt;
    return hipGetLastError();
}

}  // namespace tl
