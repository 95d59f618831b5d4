// two_opt_large.hip — REF_ORDER 2-opt (src/tsp/two_opt.rs:26-61) for tours that do not fit one CU's LDS
// (n > ~14.7 K; no upper limit but the u32 positions of the C ABI): the same first-improvement order and the same exact decision cascade as
// two_opt_ref.hip, with the tour-ordered coordinates, the tour and the L0 tile boxes in HBM (L2-resident) and the
// speculative block of rows spread over the whole chip.
//   k_rl_scan    rows [i0, i0+R) of the current cursor, one wave per row: L0 (lanes = tiles) -> live tiles ->
//                L1/L2/L3 inline (dense_tile, shared with the LDS kernel); the lexicographically first improving
//                (i, j) is reduced with a global 64-bit atomicMin on (i << 32 | j) (round 4: was i << 16 | j, n <= 65 535).
//   k_rl_apply   one workgroup, device-side cursor state machine: no hit -> advance R rows (R doubles up to 1024
//                while nothing is found, restarts at 16 after a move); hit -> swap_2opt(path, i+1, j) on P and perm,
//                rebuild the touched tile boxes, resume at (i, j+1); end of sweep -> next sweep or done.
// The host enqueues (scan, apply) pairs in batches and polls the done flag; results are bit-identical to the LDS
// kernel's and the oracle's by construction (same candidate order, same exact tests).
#include "tl_kernels.h"
#include "two_opt_common.h"

#pragma clang fp contract(off)

namespace tl {

namespace {
constexpr unsigned long long kNoKey = ~0ull;
constexpr int kRlWaves = 4;
constexpr uint32_t kRlMinRows = 16, kRlMaxRows = 1024;
}  // namespace

__global__ __launch_bounds__(kRlWaves * 64) void k_rl_scan(LargeTwoOptArgs A)
{
    LargeTwoOptState *S = A.state;
    if (S->done) return;
    const uint32_t n = A.n, i0 = S->i0, j0 = S->j0, R = S->rows;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t r = blockIdx.x * kRlWaves + (uint32_t)wave;
    const uint32_t i = i0 + r;
    if (r >= R || i + 3u >= n) return;  // rows i in [0, n-3)
    const float2 *P = A.P;
    const float2 a = P[i], b = P[i + 1u];
    const float sqab = sqdist(a, b);
    const uint32_t jmin = (r == 0u) ? j0 : (i + 2u), tmin = jmin >> 6;
    const uint32_t ngroups = ((A.n_pad >> 6) + 63u) >> 6;
    for (uint32_t g = tmin >> 6; g < ngroups; ++g) {
        // an earlier row already improves (the high word alone decides; kNoKey reads as row 0xFFFFFFFF)
        const unsigned long long kb = __hip_atomic_load(&S->key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // a fresh read per tile group
        if ((uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(kb >> 32)) < i) return;
        const uint32_t tl = (g << 6) + (uint32_t)lane;
        const float4 box = A.tbox[tl];
        const float msq = A.tmsq[tl];
        const bool live = (tl >= tmin) && ((box_lb(a.x, a.y, box) < sqab) || (box_lb(b.x, b.y, box) < msq));  // L0
        uint64_t m = __builtin_amdgcn_ballot_w64(live);
        while (m) {
            const uint32_t t = (g << 6) + (uint32_t)(__builtin_ffsll((long long)m) - 1);
            m &= m - 1;
            NoCounts tc;
            const uint64_t hm = tile_improving_mask<true>(P, n, t << 6, jmin, a.x, a.y, b.x, b.y, sqab, lane, tc);
            if (hm) {
                if (lane == 0) atomicMin(&S->key, ((unsigned long long)i << 32) | ((t << 6) + (uint32_t)(__builtin_ffsll((long long)hm) - 1)));
                return;
            }
        }
    }
}

__global__ __launch_bounds__(1024) void k_rl_apply(LargeTwoOptArgs A)
{
    LargeTwoOptState *S = A.state;
    if (S->done) return;
    const uint32_t tid = threadIdx.x, n = A.n, nrows = n - 3u;
    const int lane = tid & 63, wave = tid >> 6;
    const unsigned long long key = S->key;
    const uint32_t i0 = S->i0, R = S->rows;
    TL_SYNC();
    uint32_t ni0, nj0, nrowsstep = R;
    bool improved = S->improved != 0u;
    if (key == kNoKey) {
        ni0 = i0 + R;
        nj0 = ni0 + 2u;
        nrowsstep = R * 2u > kRlMaxRows ? kRlMaxRows : R * 2u;
    } else {
        const uint32_t is = (uint32_t)(key >> 32), js = (uint32_t)key;
        const uint32_t lo = is + 1u, hi = js, half = (hi - lo + 1u) >> 1;
        float2 *P = A.P;
        uint32_t *perm = A.perm;
        for (uint32_t t = tid; t < half; t += 1024u) {  // swap_2opt(path, i+1, j), two_opt.rs:69-79
            const float2 x = P[lo + t], y = P[hi - t];
            P[lo + t] = y;
            P[hi - t] = x;
            const uint32_t u = perm[lo + t], v = perm[hi - t];
            perm[lo + t] = v;
            perm[hi - t] = u;
        }
        TL_SYNC();
        for (uint32_t t = ((lo - 1u) >> 6) + (uint32_t)wave; t <= (hi >> 6); t += 16u) build_tile_meta(P, n, t, lane, A.tbox, A.tmsq);
        improved = true;
        ni0 = is;
        nj0 = js + 1u;
        if (nj0 > n - 2u) {
            ++ni0;
            nj0 = ni0 + 2u;
        }
        nrowsstep = kRlMinRows;
        if (tid == 0) {
            S->moves += 1;
            S->reversed += (uint64_t)(js - is);
        }
    }
    if (tid == 0) {
        if (ni0 >= nrows) {  // sweep finished (two_opt.rs:26-28)
            if (!improved) S->done = 1;
            else if (S->sweeps >= A.max_sweeps) {
                S->done = 1;
                S->status = 1;
            } else {
                S->sweeps += 1;
                improved = false;
                ni0 = 0;
                nj0 = 2;
            }
        }
        S->i0 = ni0;
        S->j0 = nj0;
        S->rows = nrowsstep;
        S->improved = improved ? 1u : 0u;
        S->key = kNoKey;
    }
}

__global__ __launch_bounds__(256) void k_rl_init(LargeTwoOptArgs A, int phase)
{
    const uint32_t n = A.n, npad = A.n_pad, ntile = npad >> 6;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (phase == 0) {
        for (uint32_t k = blockIdx.x * 256u + threadIdx.x; k <= npad; k += gridDim.x * 256u)
            A.P[k] = k < n ? A.xy[A.perm[k]] : make_float2(0.f, 0.f);
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            LargeTwoOptState *S = A.state;
            S->key = kNoKey;
            S->i0 = 0;
            S->j0 = 2;
            S->rows = kRlMinRows;
            S->improved = 0;
            S->sweeps = 1;
            S->done = n < 4u ? 1u : 0u;
            S->status = 0;
            S->moves = 0;
            S->reversed = 0;
        }
        return;
    }
    for (uint32_t t = blockIdx.x * 4u + wave; t < A.ntile_cap; t += gridDim.x * 4u) {
        if (t < ntile) build_tile_meta(A.P, n, t, (int)lane, A.tbox, A.tmsq);
        else if (lane == 0) {
            const float inf = __builtin_inff();
            A.tbox[t] = make_float4(inf, inf, -inf, -inf);
            A.tmsq[t] = -1.0f;
        }
    }
}

hipError_t launch_large_two_opt_init(const LargeTwoOptArgs &A, hipStream_t s)
{
    hipLaunchKernelGGL(k_rl_init, dim3((A.n_pad + 256u) / 256u), dim3(256), 0, s, A, 0);
    hipLaunchKernelGGL(k_rl_init, dim3((A.ntile_cap + 3u) / 4u), dim3(256), 0, s, A, 1);
    return hipGetLastError();
}

hipError_t launch_large_two_opt_round(const LargeTwoOptArgs &A, hipStream_t s)
{
    hipLaunchKernelGGL(k_rl_scan, dim3(kRlMaxRows / kRlWaves), dim3(kRlWaves * 64), 0, s, A);
    hipLaunchKernelGGL(k_rl_apply, dim3(1), dim3(1024), 0, s, A);
    return hipGetLastError();
}

}  // namespace tl
