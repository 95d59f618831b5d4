// or_opt.hip — Or-opt: best-improvement relocation of 1-, 2- or 3-city segments (reference: src/tsp/or_opt.rs;
// SURVEY.md §8(f) "next" row 3).
//
//   find_best_move (or_opt.rs:80-164): for seg_len in 1..=3, i in 0..n (segments that wrap are skipped), j in 0..n
//   outside {prev, i..i+seg_len-1}:  fwd_delta = -remove_gain + d(x,first) + d(last,y) - d(x,y) and, for seg_len > 1,
//   rev_delta with first/last swapped; the move with the lowest delta below -1e-3 wins, first in loop order
//   (seg_len, i, j, forward-before-reversed) on ties (strict `<`, :141,153).  apply_relocation (:170-184).
//
// Best-improvement, so one pass is a whole-chip scan: one wave per (seg_len, i) row, lanes along j, every f32
// expression associated exactly as the reference writes it; argmin by a packed 64-bit key
// (~delta_bits << 32 | loop-order index) reduced per wave, per workgroup, then by k_or_pick, which also applies the
// relocation in place.  Exact distances (correctly rounded sqrt) — no pruning yet; roofline: VALU.
#include "tl_kernels.h"

#pragma clang fp contract(off)

namespace tl {

namespace {

constexpr int kOrWaves = 4;
constexpr unsigned long long kNoKey64 = ~0ULL;

__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v)
{
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = __shfl_down(v, off);
        v = o < v ? o : v;
    }
    return __shfl(v, 0);
}

template <bool DM>
struct Dist {
    const float2 *Pt;
    const float *dm;
    const uint32_t *perm;
    __device__ __forceinline__ float operator()(uint32_t kp, uint32_t kq) const  // tour positions
    {
        if (DM) return dm_lookup(dm, perm[kp], perm[kq]);
        return dist(Pt[kp], Pt[kq]);
    }
};

}  // namespace

template <bool DM>
__global__ __launch_bounds__(kOrWaves * 64) void k_or_scan(OrOptArgs A)
{
    __shared__ unsigned long long s_key[kOrWaves];
    const uint32_t n = A.n;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t row = blockIdx.x * kOrWaves + (uint32_t)wave;  // (seg_len - 1) * n + i
    unsigned long long best = kNoKey64;
    const uint32_t seg_len = row / n + 1u, i = row % n;
    if (row < 3u * n && n > seg_len + 1u && i + seg_len <= n) {  // :90-92, :98-100
        const Dist<DM> D{A.Pt, A.dm, A.perm};
        const uint32_t prev = i == 0u ? n - 1u : i - 1u;
        const uint32_t after = (i + seg_len) % n;
        const uint32_t pf = i, pl = i + seg_len - 1u;
        const float remove_gain = D(prev, pf) + D(pl, after) - D(prev, after);  // :114-116
        const float neg_rg = -remove_gain;
        for (uint32_t j = (uint32_t)lane; j < n; j += 64u) {
            if (j == prev || (j >= i && j < i + seg_len)) continue;  // :123-125
            const uint32_t jy = j + 1u == n ? 0u : j + 1u;
            const float edge_xy = A.E[j];
            const float fwd = neg_rg + D(j, pf) + D(pl, jy) - edge_xy;  // :136-139
            const unsigned long long order = ((unsigned long long)row * n + j) * 2ull;
            if (fwd < -1e-3f) {
                const unsigned long long key = ((unsigned long long)(~__builtin_bit_cast(uint32_t, fwd)) << 32) | order;
                best = key < best ? key : best;
            }
            if (seg_len > 1u) {
                const float rev = neg_rg + D(j, pl) + D(pf, jy) - edge_xy;  // :148-151
                if (rev < -1e-3f) {
                    const unsigned long long key = ((unsigned long long)(~__builtin_bit_cast(uint32_t, rev)) << 32) | (order + 1ull);
                    best = key < best ? key : best;
                }
            }
        }
    }
    best = wave_min_u64(best);
    if (lane == 0) s_key[wave] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long k = s_key[0];
        for (int w = 1; w < kOrWaves; ++w) k = s_key[w] < k ? s_key[w] : k;
        A.partials[blockIdx.x] = k;
    }
}

template <bool DM>
__global__ __launch_bounds__(256) void k_or_prepare(OrOptArgs A)
{
    const uint32_t n = A.n;
    for (uint32_t k = blockIdx.x * 256u + threadIdx.x; k < n; k += gridDim.x * 256u) {
        const uint32_t p = A.perm[k], q = A.perm[k + 1u == n ? 0u : k + 1u];
        if (!DM) A.Pt[k] = A.xy[p];
        A.E[k] = DM ? dm_lookup(A.dm, p, q) : dist(A.xy[p], A.xy[q]);
    }
}

// reduce, publish the move and (optionally) apply_relocation (or_opt.rs:170-184)
__global__ __launch_bounds__(1024) void k_or_pick(OrOptArgs A, uint32_t nblocks, int apply)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *old = reinterpret_cast<uint32_t *>(smem);  // n entries
    __shared__ unsigned long long s_key[16];
    const uint32_t tid = threadIdx.x, n = A.n;
    const int lane = tid & 63, wave = tid >> 6;
    unsigned long long best = kNoKey64;
    for (uint32_t b = tid; b < nblocks; b += 1024u) {
        const unsigned long long k = A.partials[b];
        best = k < best ? k : best;
    }
    best = wave_min_u64(best);
    if (lane == 0) s_key[wave] = best;
    __syncthreads();
    best = s_key[0];
    for (int w = 1; w < 16; ++w) best = s_key[w] < best ? s_key[w] : best;
    const bool found = best != kNoKey64;
    const uint32_t order = (uint32_t)(best & 0xFFFFFFFFu);
    const uint32_t reversed = order & 1u, j = (order >> 1) % n, row = (order >> 1) / n;
    const uint32_t seg_len = row / n + 1u, i = row % n;
    if (tid == 0) {
        A.best->found = found ? 1u : 0u;
        A.best->delta_bits = ~(uint32_t)(best >> 32);
        A.best->i = i;
        A.best->j = j;
        A.best->seg_len = seg_len;
        A.best->reversed = reversed;
    }
    if (!found || !apply) return;
    uint32_t *path = A.perm;
    for (uint32_t t = tid; t < n; t += 1024u) old[t] = path[t];
    __syncthreads();
    const uint32_t insert_at = (j >= i + seg_len) ? (j - seg_len + 1u) : (j + 1u);  // index in the drained tour
    for (uint32_t t = tid; t < n; t += 1024u) {
        uint32_t src;
        if (t >= insert_at && t < insert_at + seg_len) {
            const uint32_t s = t - insert_at;
            src = i + (reversed ? (seg_len - 1u - s) : s);
        } else {
            const uint32_t u = t < insert_at ? t : t - seg_len;  // index in the drained tour
            src = u < i ? u : u + seg_len;
        }
        path[t] = old[src];
    }
}

hipError_t launch_or_opt_pass(const OrOptArgs &A, bool dm, int apply, hipStream_t s)
{
    const uint32_t rows = 3u * A.n, nblocks = (rows + kOrWaves - 1) / kOrWaves;
    const uint32_t pg = (A.n + 255u) / 256u;
    if (dm) {
        hipLaunchKernelGGL(k_or_prepare<true>, dim3(pg), dim3(256), 0, s, A);
        hipLaunchKernelGGL(k_or_scan<true>, dim3(nblocks), dim3(kOrWaves * 64), 0, s, A);
    } else {
        hipLaunchKernelGGL(k_or_prepare<false>, dim3(pg), dim3(256), 0, s, A);
        hipLaunchKernelGGL(k_or_scan<false>, dim3(nblocks), dim3(kOrWaves * 64), 0, s, A);
    }
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_or_pick), hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)A.n * 4));
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_or_pick, dim3(1), dim3(1024), (size_t)A.n * 4, s, A, nblocks, apply);
    return hipGetLastError();
}

uint32_t or_opt_scan_blocks(uint32_t n) { return (3u * n + kOrWaves - 1) / kOrWaves; }

}  // namespace tl
