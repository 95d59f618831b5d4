// two_opt_ref.hip — REF_ORDER 2-opt: the reference's first-improvement descent
// (src/tsp/two_opt.rs:26-61) as ONE persistent workgroup per descent, whole tour resident in LDS.
//
// Reference loop, per sweep:  for i in 0..n-3 { for j in i+2..n-1 {
//     if d(p[i],p[j]) + d(p[i+1],p[j+1]) < d(p[i],p[i+1]) + d(p[j],p[j+1]) { reverse p[i+1..=j] } } }
// repeated until a sweep applies no move.  Move k+1 sees the path after move k, so the scan is
// sequential *between* moves and parallel only *within* the stretch of candidates up to the next
// improving one.  MI355X mapping:
//   - one descent = one workgroup = one CU; tour-ordered coordinates P[k] = xy[perm[k]] (float2) and
//     perm (u16) live in that CU's LDS for the whole descent (10 B/city: n <= ~16 K) — zero HBM
//     traffic inside the loop; 256 CUs = 256 concurrent restarts (north-star config 4);
//   - a step speculatively evaluates a block of R rows (i0..i0+R) x all j under "no move yet":
//     lanes run along j (stride-1 LDS reads), the row endpoints a=P[i], b=P[i+1] are wave-uniform
//     (v_readlane -> SGPR operands); the lexicographically first improving (i,j) is reduced with
//     ballot/ffs per wave and one ds_min_u32 per wave on a packed key (i<<16 | j);
//   - the workgroup applies that reversal cooperatively in LDS and resumes at (i, j+1), exactly
//     where the reference's inner loop continues; R adapts to the observed gap between moves.
// Exact pruning: d_ac + d_be < d_ab + d_ce can only hold if sq_ac < sq_ab or
// sq_be < sq_ce (f32 sqrt and add are monotone), so a candidate is decided "not improving" from
// squared distances alone; only the rare survivors pay the four correctly rounded sqrt.  The
// decision for every candidate is identical to the reference's; TL_FLAG_NO_PRUNE disables it.
#include "tl_kernels.h"

#pragma clang fp contract(off)

namespace tl {

namespace {

constexpr uint32_t kNoKey = 0xFFFFFFFFu;
#ifndef TL_RMAX
#define TL_RMAX 32
#endif
#ifndef TL_STEP_COST
#define TL_STEP_COST 3000.0f
#endif
constexpr int kRMax = TL_RMAX;             // rows per speculative block (<= 63: row table is lane-resident)
constexpr float kStepCost = TL_STEP_COST;  // per-step fixed overhead in candidate-equivalents (tuning only)

#ifdef TL_PROFILE
#define TL_STAMP(var) do { if (tid == 0) { const uint64_t _t = __builtin_amdgcn_s_memtime(); prof[var] += _t - tlast; tlast = _t; } } while (0)
#else
#define TL_STAMP(var) do { } while (0)
#endif
constexpr uint32_t kQCap = 128;       // per-wave survivor queue entries (power of two, >= 2 * 64)

struct Ctl {
    uint32_t keys[4];
};

// Exact decision for up to 64 queued survivors of the squared-distance pre-test, one per lane.
// Level 2: the hardware v_sqrt_f32 (<= 1 ulp) decides every candidate whose |new - cur| exceeds a
// margin ~3x the worst-case accumulated error; level 3: the few near-ties (and degenerate tiny
// operands) take four correctly rounded sqrt — the reference's arithmetic (two_opt.rs:35-49).
__device__ __forceinline__ void flush_survivors(const float2 *P, const uint32_t *q, uint32_t head,
                                                uint32_t count, uint32_t *keyslot, int lane)
{
    const bool act = (uint32_t)lane < count;
    const uint32_t key = act ? q[(head + (uint32_t)lane) & (kQCap - 1u)] : 0u;
    const uint32_t i = key >> 16, j = key & 0xFFFFu;
    const float2 a = P[i], b = P[i + 1u], c = P[j], e = P[j + 1u];
    const float s1 = sqdist(a, c), s2 = sqdist(b, e), sab = sqdist(a, b), sce = sqdist(c, e);
    const float neu_a = __builtin_amdgcn_sqrtf(s1) + __builtin_amdgcn_sqrtf(s2);
    const float cur_a = __builtin_amdgcn_sqrtf(sab) + __builtin_amdgcn_sqrtf(sce);
    const float margin = cur_a * 1.9073486e-6f;  // 2^-19
    const float smin = fminf(fminf(s1, s2), fminf(sab, sce));
    bool imp = act & (neu_a < cur_a - margin);
    const bool tie = act & !imp & ((neu_a <= cur_a + margin) | (smin < 1e-30f) | !(cur_a < 3.0e38f));
    if (__builtin_amdgcn_ballot_w64(tie)) {
        const float neu = sqrt_rn(s1) + sqrt_rn(s2);
        const float cur = sqrt_rn(sab) + sqrt_rn(sce);
        imp = act & (tie ? (neu < cur) : imp);
    }
    if (imp) atomicMin(keyslot, key);
}

// Scans rows [0, rhi) of the block against one 64-wide j tile held in registers.
// PRUNE: survivors of the exact squared-distance pre-test are queued (per-wave ring in LDS) and
// decided 64 at a time by flush_survivors.  !PRUNE: every in-range candidate is decided inline with
// four correctly rounded sqrt (TL_FLAG_NO_PRUNE).
template <bool PRUNE, bool MASKED>
__device__ __forceinline__ void scan_rows(const float2 *P, const float2 c, const float2 e, const float sqce,
                                          const float rowx, const float rowy, const float rowsq, int rhi,
                                          const uint32_t i0, const uint32_t tb, const uint32_t j,
                                          const uint32_t j0, uint32_t *q, uint32_t &head, uint32_t &tail,
                                          uint32_t *keyslot, const int lane)
{
    float ax = readlane_f(rowx, 0), ay = readlane_f(rowy, 0);
    for (int r = 0; r < rhi; ++r) {
        const float bx = readlane_f(rowx, r + 1), by = readlane_f(rowy, r + 1);
        const float sqab = readlane_f(rowsq, r);
        float dx = ax - c.x, dy = ay - c.y;
        const float s1 = dx * dx + dy * dy;
        dx = bx - e.x;
        dy = by - e.y;
        const float s2 = dx * dx + dy * dy;
        ax = bx;
        ay = by;
        bool test;
        if (PRUNE) test = (s1 < sqab) | (s2 < sqce);
        else test = (sqce >= 0.0f);  // every in-range lane
        if (MASKED) {
            const uint32_t jmin = (r == 0) ? j0 : (i0 + (uint32_t)r + 2u);
            test = test & (j >= jmin);
        }
        const uint64_t m = __builtin_amdgcn_ballot_w64(test);
        if (m == 0) continue;
        const uint32_t key = ((i0 + (uint32_t)r) << 16) | j;
        if (PRUNE) {
            const uint32_t off = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            if (test) q[(tail + off) & (kQCap - 1u)] = key;
            tail += (uint32_t)__builtin_popcountll(m);
            if (tail - head >= 64u) {
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                flush_survivors(P, q, head, 64u, keyslot, lane);
                head += 64u;
                // a hit bounds the rows this tile still has to look at
                const uint32_t kb = (uint32_t)__builtin_amdgcn_readfirstlane((int)*keyslot);
                if (kb != kNoKey) {
                    const int lim = (int)((kb >> 16) - i0) + (tb <= (kb & 0xFFFFu) ? 1 : 0);
                    rhi = rhi < lim ? rhi : lim;
                }
            }
        } else {
            const float neu = sqrt_rn(s1) + sqrt_rn(s2);
            const float cur = sqrt_rn(sqab) + sqrt_rn(sqce);
            const bool imp = test & (neu < cur);  // two_opt.rs:35-49
            const uint64_t mi = __builtin_amdgcn_ballot_w64(imp);
            if (mi) {
                if (lane == 0) atomicMin(keyslot, (key & 0xFFFF0000u) | (tb + (uint32_t)(__builtin_ffsll((long long)mi) - 1)));
                return;  // later rows of this tile are lexicographically later
            }
        }
    }
}

}  // namespace

template <int NT, bool PRUNE>
__global__ __launch_bounds__(NT) void k_two_opt_ref_lds(TwoOptBatchArgs A)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NW = NT / 64;
    const uint32_t n = A.n, npad = A.n_pad;
    float2 *P = reinterpret_cast<float2 *>(smem);
    uint16_t *perm = reinterpret_cast<uint16_t *>(smem + (size_t)npad * 8);
    Ctl *ctl = reinterpret_cast<Ctl *>(smem + (size_t)npad * 10);
    // survivor queues (NW x kQCap u32) during the descent; reused as NT floats for the cost sum
    uint32_t *queues = reinterpret_cast<uint32_t *>(smem + (size_t)npad * 10 + 64);
    float *scratch = reinterpret_cast<float *>(queues);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform -> SGPR loop control
    const uint32_t d = blockIdx.x;
    const float2 *__restrict__ xy = A.xy;

    // ---------------------------------------------------------------- initial tour
    if (A.init_mode == TL_INIT_ARRAY) {
        const uint32_t *__restrict__ src = A.init + (size_t)d * n;
        for (uint32_t k = tid; k < n; k += NT) perm[k] = (uint16_t)src[k];
    } else {
        for (uint32_t k = tid; k < n; k += NT) perm[k] = (uint16_t)k;  // two_opt.rs:18-20
    }
    if (A.init_mode == TL_INIT_SEEDED) {
        // Fisher-Yates `for i in (1..n).rev(): j = rng % (i+1); swap` from splitmix64(seed + r):
        // the draws are counter-based, so compute them in parallel, then one lane applies the swaps.
        uint16_t *draws = reinterpret_cast<uint16_t *>(P);  // P is not live yet
        const uint64_t s = A.seed + (uint64_t)(A.first + d);
        for (uint32_t i = 1 + tid; i < n; i += NT) {
            const uint64_t kth = (uint64_t)(n - 1 - i);  // i = n-1 is draw 0
            draws[i] = (uint16_t)(splitmix64_at(s, kth) % ((uint64_t)i + 1));
        }
        __syncthreads();
        if (tid == 0) {
            for (uint32_t i = n - 1; i >= 1; --i) {
                const uint32_t j = draws[i];
                const uint16_t t = perm[i];
                perm[i] = perm[j];
                perm[j] = t;
            }
        }
    }
    if (tid < 4) ctl->keys[tid] = kNoKey;
    __syncthreads();
    for (uint32_t k = tid; k < npad; k += NT) P[k] = (k < n) ? xy[perm[k]] : make_float2(0.f, 0.f);
    __syncthreads();

    // ---------------------------------------------------------------- descent
    const uint32_t nrows = n - 3;  // rows i in [0, n-3); j in [i+2, n-2]
    uint32_t i0 = 0, j0 = 2;
    bool improved = false;
    uint32_t sweeps = 1, step = 0, status = 0;
    uint64_t moves = 0, reversed = 0;
    float gap_est = 0.0f, since = 0.0f;
#ifdef TL_PROFILE
    uint64_t prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t tlast = __builtin_amdgcn_s_memtime();
    const uint64_t t_begin = tlast, rt_begin = __builtin_amdgcn_s_memrealtime();
    uint64_t rows_total = 0;
#endif

    while (n >= 4) {
        const uint32_t slot = step % 3u;
        if (tid == 0) ctl->keys[(step + 1u) % 3u] = kNoKey;  // slot of the next step
        ++step;

        // rows of this block
        const float rowlen = (float)(n - 2u - i0);
        const float g = fmaxf(gap_est, since);
        int R = (int)(sqrtf(2.0f * g * kStepCost) / rowlen);
        R = R < 1 ? 1 : (R > kRMax ? kRMax : R);
        if ((uint32_t)R > nrows - i0) R = (int)(nrows - i0);

        // lane-resident row table: lane l holds P[i0+l] and sq(P[i0+l], P[i0+l+1])
        const float2 rp = P[i0 + (uint32_t)lane];
        const float2 rq = P[i0 + (uint32_t)lane + 1u];
        const float rowsq = sqdist(rp, rq);

        TL_STAMP(0);
        // row 0 resumes at j0; rows r >= 1 start at their own diagonal i0+r+2 (<= j0 possible)
        const uint32_t tile_lo = ((R == 1) ? j0 : (i0 + 2u)) >> 6;
        uint32_t *q = queues + (uint32_t)wave * kQCap;
        uint32_t qhead = 0, qtail = 0;
        uint32_t *keyslot = &ctl->keys[slot];
        for (uint32_t tb = (tile_lo + (uint32_t)wave) << 6; tb <= n - 2u; tb += (uint32_t)NW << 6) {
            const uint32_t j = tb + (uint32_t)lane;
            float2 c = P[j];
            const float2 e = P[j + 1u];
            float sqce = sqdist(c, e);
            if (j > n - 2u) {  // out-of-range lane: can never pass either test
                c.x = 1e30f;
                sqce = -1.0f;
            }
            // rows that can still matter for this tile
            int rhi = R;
            {
                const int lim = (int)(tb + 62u - i0);  // need i + 2 <= tb + 63
                rhi = rhi < lim ? rhi : lim;
                const uint32_t kb = (uint32_t)__builtin_amdgcn_readfirstlane((int)*keyslot);
                if (kb != kNoKey) {
                    const int kr = (int)((kb >> 16) - i0), kj = (int)(kb & 0xFFFFu);
                    const int lim2 = kr + ((int)tb <= kj ? 1 : 0);
                    rhi = rhi < lim2 ? rhi : lim2;
                }
            }
            if (rhi <= 0) continue;
            // first row of the block resumes at j0; rows below the diagonal need j >= i+2
            const bool masked = (tb < j0) | (tb < i0 + (uint32_t)rhi + 1u);
            if (masked) scan_rows<PRUNE, true>(P, c, e, sqce, rp.x, rp.y, rowsq, rhi, i0, tb, j, j0, q, qhead, qtail, keyslot, lane);
            else scan_rows<PRUNE, false>(P, c, e, sqce, rp.x, rp.y, rowsq, rhi, i0, tb, j, j0, q, qhead, qtail, keyslot, lane);
            // decide this tile's survivors before moving on: a hit found now stops the other tiles/waves early
            if (PRUNE && qtail != qhead) {
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                flush_survivors(P, q, qhead, qtail - qhead, keyslot, lane);
                qhead = qtail;
            }
        }
        TL_STAMP(1);
        __syncthreads();
        TL_STAMP(2);
        const uint32_t key = (uint32_t)__builtin_amdgcn_readfirstlane((int)ctl->keys[slot]);
#ifdef TL_PROFILE
        rows_total += (uint64_t)R;
#endif

        if (key == kNoKey) {
            since += (float)R * rowlen;
            i0 += (uint32_t)R;
            j0 = i0 + 2u;
        } else {
            const uint32_t is = key >> 16, js = key & 0xFFFFu;
            // two_opt.rs:50,69-79  swap_2opt(path, i+1, j)
            const uint32_t lo = is + 1u, hi = js;
            const uint32_t half = (hi - lo + 1u) >> 1;
            for (uint32_t t = tid; t < half; t += NT) {
                const float2 x = P[lo + t], y = P[hi - t];
                P[lo + t] = y;
                P[hi - t] = x;
                const uint16_t u = perm[lo + t], v = perm[hi - t];
                perm[lo + t] = v;
                perm[hi - t] = u;
            }
            TL_STAMP(3);
            __syncthreads();
            TL_STAMP(4);
            improved = true;
            ++moves;
            reversed += (uint64_t)(js - is);
            since += (float)(is - i0) * rowlen + (float)(js - j0);
            gap_est = 0.5f * (gap_est + since);
            since = 0.0f;
            i0 = is;
            j0 = js + 1u;
            if (j0 > n - 2u) {
                ++i0;
                j0 = i0 + 2u;
            }
        }
        if (i0 >= nrows) {  // sweep finished (two_opt.rs:26-28)
            if (!improved) break;
            if (sweeps >= A.max_sweeps) {
                status = 1;
                break;
            }
            improved = false;
            ++sweeps;
            i0 = 0;
            j0 = 2;
        }
    }

    // ---------------------------------------------------------------- results
    uint32_t *__restrict__ out = A.out_pos + (size_t)d * n;
    for (uint32_t k = tid; k < n; k += NT) out[k] = perm[k];

    // Solution::from_parts -> tour_length_by_pos (distance_matrix.rs:235-245): sequential f32 sum,
    // closing edge first.  Edge lengths in parallel, the sum by one lane in tour order.
    float total = 0.0f;
    if (n >= 2) total = dist(P[n - 1], P[0]);
    for (uint32_t base = 0; base + 1 < n; base += NT) {
        const uint32_t k = base + tid;
        scratch[tid] = (k + 1 < n) ? dist(P[k], P[k + 1]) : 0.0f;
        __syncthreads();
        if (tid == 0) {
            const uint32_t cnt = (n - 1 - base) < (uint32_t)NT ? (n - 1 - base) : (uint32_t)NT;
            const float4 *s4 = reinterpret_cast<const float4 *>(scratch);
            uint32_t q = 0;
            for (; q + 4 <= cnt; q += 4) {
                const float4 v = s4[q >> 2];
                total += v.x;
                total += v.y;
                total += v.z;
                total += v.w;
            }
            for (; q < cnt; ++q) total += scratch[q];
        }
        __syncthreads();
    }
    if (tid == 0) {
        A.out_cost[d] = total;
        uint64_t *st = A.out_stats + (size_t)d * TL_STATS_STRIDE;
        st[0] = sweeps;
        st[1] = moves;
        st[2] = reversed;
        st[3] = status;
        st[4] = step;
#ifdef TL_PROFILE
        for (int q = 0; q < 5; ++q) st[5 + q] = prof[q];
        st[10] = __builtin_amdgcn_s_memtime() - t_begin;
        st[11] = __builtin_amdgcn_s_memrealtime() - rt_begin;
        st[12] = rows_total;
#endif
    }
}

// ------------------------------------------------------------------------------------------------
// launch
// ------------------------------------------------------------------------------------------------
size_t two_opt_ref_lds_bytes(uint32_t n, uint32_t *n_pad_out, int nt)
{
    const uint32_t n_pad = ((n + 64u + 63u) / 64u) * 64u;  // P[j+1] of any lane of the last tile is in range
    if (n_pad_out) *n_pad_out = n_pad;
    const size_t tail = (size_t)(nt / 64) * kQCap * 4;  // survivor queues, >= nt floats of scratch
    return (size_t)n_pad * 10 + 64 + (tail > (size_t)nt * 4 ? tail : (size_t)nt * 4);
}

template <int NT, bool PRUNE>
static hipError_t launch_one(const TwoOptBatchArgs &A, uint32_t count, size_t lds, hipStream_t s)
{
    auto kern = k_two_opt_ref_lds<NT, PRUNE>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(count), dim3(NT), lds, s, A);
    return hipGetLastError();
}

hipError_t launch_two_opt_ref_lds(const TwoOptBatchArgs &A, uint32_t count, bool prune, hipStream_t s)
{
    constexpr int NT = TL_TWO_OPT_NT;
    uint32_t n_pad = 0;
    const size_t lds = two_opt_ref_lds_bytes(A.n, &n_pad, NT);
    TwoOptBatchArgs B = A;
    B.n_pad = n_pad;
    return prune ? launch_one<NT, true>(B, count, lds, s) : launch_one<NT, false>(B, count, lds, s);
}

}  // namespace tl
