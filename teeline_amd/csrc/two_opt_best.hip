// two_opt_best.hip — TL_MODE_BEST_SWEEP: best-improvement 2-opt (this build's own throughput mode; NOT the
// reference's algorithm — the reference is first-improvement, src/tsp/two_opt.rs:26-61).  Specification =
// oracle/tlo_two_opt_best: per sweep every (i,j) of the same open-path candidate set is decided, the move with the
// lowest f32 delta (new - cur) < 0 wins, lowest (i,j) on ties; apply; repeat until a sweep finds none.
//
// Unlike REF_ORDER a sweep has no sequential dependency, so the whole chip works on it:
//   k_bs_init    tour-ordered coordinates P[k] = xy[perm[k]] and per-tile L0 metadata, in HBM (L2-resident)
//   k_bs_scan    one wave per row i.  L0 (lanes = tiles) -> live-tile mask, L1 (lanes = j) on live tiles, L2 hardware
//                sqrt to discard the clearly non-improving, exact delta for the rest.  Argmin by wavefront DPP/shuffle
//                reduction of a packed 64-bit key (~delta bits << 32 | i << 16 | j), then per workgroup.  Every row's best key is
//                cached; a sweep decides only the rows (and, for the rows in front of the move, the columns) the previous move touched.
//   k_bs_apply   one workgroup: reduces the per-workgroup keys, applies swap_2opt(path, i+1, j) on P and perm,
//                rebuilds the L0 metadata of the touched tiles, bumps the counters / sets the done flag, files the move for the scan.
// The host replays 64 sweeps (scan + apply) as one hipGraph per poll of the done flag.
#include "tl_kernels.h"
#include "two_opt_common.h"

#pragma clang fp contract(off)

namespace tl {

namespace {

constexpr unsigned long long kNoKey64 = ~0ULL;

__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v)
{
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = __shfl_down(v, off);
        v = o < v ? o : v;
    }
    return __shfl(v, 0);
}

}  // namespace

__global__ __launch_bounds__(256) void k_bs_init(BestSweepArgs A)
{
    const uint32_t n = A.n, npad = A.n_pad, ntile = npad >> 6;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (A.phase == 0) {
        for (uint32_t k = blockIdx.x * 256u + threadIdx.x; k <= npad; k += gridDim.x * 256u)
            A.P[k] = k < n ? A.xy[A.perm[k]] : make_float2(0.f, 0.f);
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

// A row's best candidate among the columns [jlo, jhi] (clipped to the row's own [i+2, n-2]) as a packed key, or kNoKey64.
__device__ __forceinline__ unsigned long long bs_row_best(const BestSweepArgs &A, const float2 *__restrict__ P, uint32_t i, uint32_t jlo, uint32_t jhi, int lane)
{
    const uint32_t n = A.n;
    unsigned long long best = kNoKey64;
    const float2 a = P[i], b = P[i + 1u];
    const float sqab = sqdist(a, b);
    const float dab_a = __builtin_amdgcn_sqrtf(sqab);
    const uint32_t jmin = i + 2u > jlo ? i + 2u : jlo, jmax = n - 2u < jhi ? n - 2u : jhi;
    if (jmin > jmax) return best;
    const uint32_t tmin = jmin >> 6, tmax = jmax >> 6;
    for (uint32_t g = tmin >> 6; g <= (tmax >> 6); ++g) {
        const uint32_t tl = (g << 6) + (uint32_t)lane;
        const float4 box = A.tbox[tl];
        const float msq = A.tmsq[tl];
        const bool live = (tl >= tmin) && (tl <= tmax) && ((box_lb(a.x, a.y, box) < sqab) || (box_lb(b.x, b.y, box) < msq));  // L0
        uint64_t m = __builtin_amdgcn_ballot_w64(live);
        while (m) {
            const uint32_t t = (g << 6) + (uint32_t)(__builtin_ffsll((long long)m) - 1);
            m &= m - 1;
            const uint32_t j = (t << 6) + (uint32_t)lane;
            const float2 c = P[j], e = P[j + 1u];
            const float sqce = sqdist(c, e), s1 = sqdist(a, c), s2 = sqdist(b, e);
            bool test = (j >= jmin) & (j <= jmax) & ((s1 < sqab) | (s2 < sqce));  // L1
            if (!__builtin_amdgcn_ballot_w64(test)) continue;
            // L2: discard what the hardware sqrt already shows to be non-improving by a safe margin
            const float neu_a = __builtin_amdgcn_sqrtf(s1) + __builtin_amdgcn_sqrtf(s2);
            const float cur_a = dab_a + __builtin_amdgcn_sqrtf(sqce);
            test = test & !(neu_a > cur_a + cur_a * 1.9073486e-6f);
            if (!__builtin_amdgcn_ballot_w64(test)) continue;
            const float neu = sqrt_rn(s1) + sqrt_rn(s2);
            const float cur = sqrt_rn(sqab) + sqrt_rn(sqce);
            if (test & (neu < cur)) {
                const float delta = neu - cur;
                const unsigned long long key = ((unsigned long long)(~__builtin_bit_cast(uint32_t, delta)) << 32) | (i << 16) | j;
                best = key < best ? key : best;
            }
        }
    }
    return wave_min_u64(best);
}

// One wave per row.  Round 5: a row's best candidate is CACHED (A.rowkey) and a sweep decides only what the previous move can have changed.
// The move reversed positions [is+1 .. js]: a row r > js reads none of them — its cached key stands; a row is <= r <= js has a new a or b —
// decided afresh; a row r < is keeps its a and b, and of its columns only j in [is .. js] have a new c = p[j] or e = p[j+1]: its new best
// is the smaller of the cached key and the best over those columns — unless the cached key's own column lies in that range (then afresh).
// Same keys, hence the same argmin with the same tie rule (lowest delta, then lowest (i, j)), as deciding every row afresh.
// (Scan and apply as ONE launch — 256 workgroups of 16 waves, rows strided over all waves, the workgroup that draws the sweep's last
//  ticket applies the move — was built and measured: 19.0 us per sweep against 10.2 for the two launches at n = 10^4, NOTEBOOK r5.4.)
constexpr int kBsWaves = 4;  // rows per scan workgroup
__global__ __launch_bounds__(kBsWaves * 64) void k_bs_scan(BestSweepArgs A)
{
    __shared__ unsigned long long s_key[kBsWaves];
    const uint32_t n = A.n;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t i = blockIdx.x * kBsWaves + (uint32_t)wave;  // row
    unsigned long long best = kNoKey64;
    if (A.counters[2] == 0 && i + 3u < n) {  // not done; rows i in [0, n-3)
        const float2 *__restrict__ P = A.P;
        const uint32_t mv = A.move[0], is = A.move[1], js = A.move[2];
        if (mv == 0u || (i >= is && i <= js)) {
            best = bs_row_best(A, P, i, 0u, n, lane);
            if (lane == 0) A.rowkey[i] = best;
        } else if (i > js) {
            best = A.rowkey[i];
        } else {
            const unsigned long long ck = A.rowkey[i];
            const uint32_t cj = (uint32_t)(ck & 0xFFFFu);
            if (ck != kNoKey64 && cj >= is && cj <= js) {
                best = bs_row_best(A, P, i, 0u, n, lane);
            } else {
                const unsigned long long pk = bs_row_best(A, P, i, is, js, lane);
                best = pk < ck ? pk : ck;
            }
            if (best != ck && lane == 0) A.rowkey[i] = best;
        }
    }
    if (lane == 0) s_key[wave] = best;
    TL_SYNC();
    if (threadIdx.x == 0) {
        unsigned long long k = s_key[0];
        for (int w = 1; w < kBsWaves; ++w) k = s_key[w] < k ? s_key[w] : k;
        A.partials[blockIdx.x] = k;
    }
}

__global__ __launch_bounds__(1024) void k_bs_apply(BestSweepArgs A, uint32_t nblocks)
{
    __shared__ unsigned long long s_key[16];
    const uint32_t tid = threadIdx.x, n = A.n;
    const int lane = tid & 63, wave = tid >> 6;
    if (A.counters[2] != 0) return;  // done in an earlier sweep of this batch
    unsigned long long best = kNoKey64;
    for (uint32_t b = tid; b < nblocks; b += 1024u) {
        const unsigned long long k = A.partials[b];
        best = k < best ? k : best;
    }
    best = wave_min_u64(best);
    if (lane == 0) s_key[wave] = best;
    TL_SYNC();
    best = s_key[0];
    for (int w = 1; w < 16; ++w) best = s_key[w] < best ? s_key[w] : best;
    if (best == kNoKey64) {
        if (tid == 0) {
            A.counters[0] += 1;  // the final, move-less sweep
            A.counters[2] = 1;   // done
        }
        return;
    }
    const uint32_t is = (uint32_t)((best >> 16) & 0xFFFFu), js = (uint32_t)(best & 0xFFFFu);
    const uint32_t lo = is + 1u, hi = js, half = (hi - lo + 1u) >> 1;
    float2 *P = A.P;
    uint32_t *perm = A.perm;
    for (uint32_t t = tid; t < half; t += 1024u) {  // swap_2opt(path, i+1, j)
        const float2 x = P[lo + t], y = P[hi - t];
        P[lo + t] = y;
        P[hi - t] = x;
        const uint32_t u = perm[lo + t], v = perm[hi - t];
        perm[lo + t] = v;
        perm[hi - t] = u;
    }
    TL_SYNC();
    for (uint32_t t = ((lo - 1u) >> 6) + (uint32_t)wave; t <= (hi >> 6); t += 16u) build_tile_meta(P, n, t, lane, A.tbox, A.tmsq);
    if (tid == 0) {
        A.counters[0] += 1;                       // sweeps
        A.counters[1] += 1;                       // moves
        A.counters[3] += (uint64_t)(js - is);     // reversed elements
        A.move[1] = is;                           // what the next sweep's scan needs to know
        A.move[2] = js;
        A.move[0] = 1u;
    }
}

hipError_t launch_best_sweep_init(const BestSweepArgs &A, hipStream_t s)
{
    BestSweepArgs B = A;
    B.phase = 0;
    hipLaunchKernelGGL(k_bs_init, dim3((A.n_pad + 256u) / 256u), dim3(256), 0, s, B);
    B.phase = 1;
    hipLaunchKernelGGL(k_bs_init, dim3((A.ntile_cap + 3u) / 4u), dim3(256), 0, s, B);
    return hipGetLastError();
}

hipError_t launch_best_sweep_round(const BestSweepArgs &A, hipStream_t s)
{
    const uint32_t nblocks = best_sweep_scan_blocks(A.n);
    hipLaunchKernelGGL(k_bs_scan, dim3(nblocks), dim3(kBsWaves * 64), 0, s, A);
    hipLaunchKernelGGL(k_bs_apply, dim3(1), dim3(1024), 0, s, A, nblocks);
    return hipGetLastError();
}

uint32_t best_sweep_scan_blocks(uint32_t n) { return (n - 3u + kBsWaves - 1) / kBsWaves; }

}  // namespace tl
