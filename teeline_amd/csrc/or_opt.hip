// or_opt.hip — Or-opt: best-improvement relocation of 1-, 2- or 3-city segments (reference: src/tsp/or_opt.rs;
// SURVEY.md §8(f) "next" row 3).
//
//   find_best_move (or_opt.rs:80-164): for seg_len in 1..=3, i in 0..n (segments that wrap are skipped), j in 0..n
//   outside {prev, i..i+seg_len-1}:  fwd_delta = -remove_gain + d(x,first) + d(last,y) - d(x,y) and, for seg_len > 1,
//   rev_delta with first/last swapped; the move with the lowest delta below -1e-3 wins, first in loop order
//   (seg_len, i, j, forward-before-reversed) on ties (strict `<`, :141,153).  apply_relocation (:170-184).
//
// Best-improvement, so one pass is a whole-chip scan: a wave takes 8 consecutive segment starts x a slab of insertion
// points (lanes along j) and shares the distances between the five placement kinds and the 8 starts (k_or_scan), every
// f32 expression associated exactly as the reference writes it; argmin by a packed 96-bit key
// (~delta_bits << 64 | loop-order index; the index 6 n^2 needs more than 32 bits from n = 26 755 on — round 4) reduced per wave,
// per workgroup, then by k_or_pick, which also applies the relocation in place.  Exact distances (correctly rounded sqrt); roofline: VALU.
#include "tl_kernels.h"

#pragma clang fp contract(off)

namespace tl {

namespace {

constexpr int kOrWaves = 4;
constexpr int kOrIR = 8;          // consecutive segment starts served from one set of distance registers
constexpr uint32_t kOrTargetWaves = 4096;  // waves a scan should at least consist of: small tours take fewer 63-wide chunks of
                                           // insertion points per wave (or_opt_chunks)
typedef unsigned __int128 key_t;  // (~delta bits) << 64 | loop-order index
constexpr unsigned long long kNoKey64 = ~0ULL;
__device__ __forceinline__ key_t make_key(unsigned long long hi, unsigned long long lo) { return ((key_t)hi << 64) | (key_t)lo; }
__device__ __forceinline__ key_t no_key() { return make_key(kNoKey64, kNoKey64); }

__device__ __forceinline__ key_t wave_min_key(key_t v)
{
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long oh = __shfl_down((unsigned long long)(v >> 64), off), ol = __shfl_down((unsigned long long)v, off);
        const key_t o = make_key(oh, ol);
        v = o < v ? o : v;
    }
    return make_key(__shfl((unsigned long long)(v >> 64), 0), __shfl((unsigned long long)v, 0));
}

static uint32_t or_opt_grid_x(uint32_t n) { return ((n + kOrIR - 1) / kOrIR + kOrWaves - 1) / kOrWaves; }
static uint32_t or_opt_chunks(uint32_t n)  // 63-wide chunks of insertion points per wave
{
    const uint32_t total = (n + 62u) / 63u, groups = (n + kOrIR - 1) / kOrIR;
    uint32_t slabs = (kOrTargetWaves + groups - 1u) / groups;
    if (slabs > total) slabs = total;
    if (slabs == 0u) slabs = 1u;
    return (total + slabs - 1u) / slabs;
}
static uint32_t or_opt_grid_y(uint32_t n) { return ((n + 62u) / 63u + or_opt_chunks(n) - 1u) / or_opt_chunks(n); }

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

// One wave = kOrIR consecutive segment starts i0..i0+kOrIR-1 x a slab of insertion points (kOrChunks chunks of 63).
// Every placement of the five kinds (len 1 fwd; len 2, 3 fwd and reversed) of a pair (i, j) is a sum of the row constant,
// the tour edge (x_j, y_j) and two distances out of { d(x_j, P[i+m]), d(y_j, P[i+m]) : m = 0, 1, 2 } — and y_j = x_{j+1}.
// So a chunk computes d(x_j, P[i0+m]) once for m = 0..kOrIR+1 (lane = j, lane 63 only supplies x of the next j),
// gets the y-distances from the neighbouring lane, and serves all kOrIR starts from those registers: 1.25 correctly
// rounded distances per (i, j) instead of the 10 a row-per-wave scan evaluates.  The f32 expressions keep the
// reference's association (or_opt.rs:136-139, :148-151); distances are symmetric bit for bit.
template <bool DM>
__global__ __launch_bounds__(kOrWaves * 64) void k_or_scan(OrOptArgs A, uint32_t chunks)
{
    __shared__ unsigned long long s_key[kOrWaves][2];
    if (A.run && A.run->done) return;  // a later pass of a batch whose descent is over
    const uint32_t n = A.n;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t i0 = (blockIdx.x * kOrWaves + (uint32_t)wave) * kOrIR;
    const uint32_t jlo = blockIdx.y * (chunks * 63u);
    key_t best = no_key();
    float bestd = __builtin_inff();
    if (i0 < n) {
        const Dist<DM> D{A.Pt, A.dm, A.perm};
        // row constants: lane (len-1)*kOrIR + r holds -remove_gain of (seg_len, i0 + r) (:114-116), rowmask its validity (:90-92, :98-100)
        float nrg = 0.0f;
        bool rv = false;
        if (lane < 3 * kOrIR) {
            const uint32_t len = (uint32_t)lane / kOrIR + 1u, i = i0 + (uint32_t)lane % kOrIR;
            rv = i < n && n > len + 1u && i + len <= n;
            if (rv) {
                const uint32_t prev = i == 0u ? n - 1u : i - 1u, after = (i + len) % n, pl = i + len - 1u;
                const float remove_gain = D(prev, i) + D(pl, after) - D(prev, after);
                nrg = -remove_gain;
            }
        }
        const uint64_t rowmask = __builtin_amdgcn_ballot_w64(rv);
        for (uint32_t c = 0; c < chunks; ++c) {
            const uint32_t jb = jlo + c * 63u;
            if (jb >= n) break;
            const uint32_t j = jb + (uint32_t)lane;
            const bool real = lane < 63 && j < n;
            const uint32_t jj = j < n ? j : 0u;  // position of x_j; j == n is the wrap (y of j = n-1 is P[0]), lanes beyond are unused
            const float e = real ? A.E[j] : 0.0f;
            float dX[kOrIR + 2], dY[kOrIR + 2];
#pragma unroll
            for (int m = 0; m < kOrIR + 2; ++m) {
                const uint32_t pm = i0 + (uint32_t)m < n ? i0 + (uint32_t)m : n - 1u;  // beyond the tour: unused by any valid row
                dX[m] = D(jj, pm);
                // d(y_j, P[i0+m]) = d(x_{j+1}, P[i0+m]): the lane above, one DPP wave shift (lane 63, the helper lane, gets 0)
                dY[m] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, dX[m]), 0x130 /* wave_shl:1 */, 0xf, 0xf, false));
            }
#pragma unroll
            for (int r = 0; r < kOrIR; ++r) {
                const uint32_t i = i0 + (uint32_t)r;
                if (i >= n) break;
                const uint32_t prev = i == 0u ? n - 1u : i - 1u;
                // the five placements of (i, j); an invalid one (row or column excluded, :90-92, :98-100, :123-125) reads +inf.
                // Only when the smallest of them can still beat this lane's best are the 64-bit keys looked at.
                float val[5];
                const bool okj = real & (j != prev);
                const float inf = __builtin_inff();
#pragma unroll
                for (int len = 1; len <= 3; ++len) {
                    const int rl = (len - 1) * kOrIR + r;
                    const bool ok = okj & !((j - i) < (uint32_t)len) & (bool)((rowmask >> rl) & 1ull);
                    const float nr = readlane_f(nrg, rl);
                    const float fwd = nr + dX[r] + dY[r + len - 1] - e;  // :136-139  -rg + d(x,first) + d(last,y) - d(x,y)
                    val[len == 1 ? 0 : 2 * len - 3] = ok ? fwd : inf;
                    if (len > 1) {
                        const float rev = nr + dX[r + len - 1] + dY[r] - e;  // :148-151  -rg + d(x,last) + d(first,y) - d(x,y)
                        val[2 * len - 2] = ok ? rev : inf;
                    }
                }
                const float vmin = fminf(fminf(fminf(val[0], val[1]), fminf(val[2], val[3])), val[4]);  // NaN deltas drop out like in `<`
                if (__builtin_amdgcn_ballot_w64((vmin < -1e-3f) & (vmin <= bestd))) {
#pragma unroll
                    for (int q = 0; q < 5; ++q) {  // loop order within (i, j): len 1 fwd; len 2 fwd, rev; len 3 fwd, rev — the order index decides ties
                        const int len = q == 0 ? 1 : (q + 3) / 2;
                        const unsigned long long order = ((unsigned long long)((uint32_t)(len - 1) * n + i) * n + j) * 2ull + (unsigned long long)(q != 0 && (q & 1) == 0);
                        const float v = val[q];
                        if ((v < -1e-3f) & (v <= bestd)) {
                            const key_t key = make_key((unsigned long long)(~__builtin_bit_cast(uint32_t, v)), order);
                            if (key < best) {
                                best = key;
                                bestd = v;
                            }
                        }
                    }
                }
            }
        }
    }
    best = wave_min_key(best);
    if (lane == 0) {
        s_key[wave][0] = (unsigned long long)(best >> 64);
        s_key[wave][1] = (unsigned long long)best;
    }
    TL_SYNC();
    if (threadIdx.x == 0) {
        key_t k = make_key(s_key[0][0], s_key[0][1]);
        for (int w = 1; w < kOrWaves; ++w) {
            const key_t o = make_key(s_key[w][0], s_key[w][1]);
            k = o < k ? o : k;
        }
        unsigned long long *out = A.partials + 2u * (size_t)(blockIdx.y * gridDim.x + blockIdx.x);
        out[0] = (unsigned long long)(k >> 64);
        out[1] = (unsigned long long)k;
    }
}

template <bool DM>
__global__ __launch_bounds__(256) void k_or_prepare(OrOptArgs A)
{
    if (A.run && A.run->done) return;
    const uint32_t n = A.n;
    for (uint32_t k = blockIdx.x * 256u + threadIdx.x; k < n; k += gridDim.x * 256u) {
        const uint32_t p = A.perm[k], q = A.perm[k + 1u == n ? 0u : k + 1u];
        if (!DM) A.Pt[k] = A.xy[p];
        A.E[k] = DM ? dm_lookup(A.dm, p, q) : dist(A.xy[p], A.xy[q]);
    }
}

// reduce, publish the move and (optionally) apply_relocation (or_opt.rs:170-184)
__global__ __launch_bounds__(1024) void k_or_pick(OrOptArgs A, uint32_t nblocks, int apply, int stage_lds)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // the pre-move tour: in LDS where n entries fit, else in the workspace (one workgroup: its own barrier orders the copy)
    uint32_t *old = stage_lds ? reinterpret_cast<uint32_t *>(smem) : A.scratch;
    __shared__ unsigned long long s_key[16][2];
    const uint32_t tid = threadIdx.x, n = A.n;
    const int lane = tid & 63, wave = tid >> 6;
    if (A.run && A.run->done) return;
    key_t best = no_key();
    for (uint32_t b = tid; b < nblocks; b += 1024u) {
        const key_t k = make_key(A.partials[2u * (size_t)b], A.partials[2u * (size_t)b + 1u]);
        best = k < best ? k : best;
    }
    best = wave_min_key(best);
    if (lane == 0) {
        s_key[wave][0] = (unsigned long long)(best >> 64);
        s_key[wave][1] = (unsigned long long)best;
    }
    TL_SYNC();
    best = make_key(s_key[0][0], s_key[0][1]);
    for (int w = 1; w < 16; ++w) {
        const key_t o = make_key(s_key[w][0], s_key[w][1]);
        best = o < best ? o : best;
    }
    const bool found = best != no_key();
    const unsigned long long order = (unsigned long long)best;
    const uint32_t reversed = (uint32_t)(order & 1ull), j = (uint32_t)((order >> 1) % n);
    const unsigned long long row = (order >> 1) / n;
    const uint32_t seg_len = (uint32_t)(row / n) + 1u, i = (uint32_t)(row % n);
    if (tid == 0) {
        A.best->found = found ? 1u : 0u;
        A.best->delta_bits = ~(uint32_t)(unsigned long long)(best >> 64);
        A.best->i = i;
        A.best->j = j;
        A.best->seg_len = seg_len;
        A.best->reversed = reversed;
        if (A.run) {  // or_opt.rs:45 `while let Some(best) = find_best_move(..)`: count the pass, file the move or end the descent
            A.run->passes += 1u;
            if (!found) {
                A.run->done = 1u;
            } else {
                const uint32_t m = A.run->moves;
                if (m < A.run->log_cap) {
                    A.log[4u * m + 0u] = i;
                    A.log[4u * m + 1u] = j;
                    A.log[4u * m + 2u] = seg_len;
                    A.log[4u * m + 3u] = reversed;
                }
                A.run->moves = m + 1u;
            }
        }
    }
    if (!found || !apply) return;
    uint32_t *path = A.perm;
    for (uint32_t t = tid; t < n; t += 1024u) old[t] = path[t];
    TL_SYNC();
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

hipError_t launch_or_opt_pass(const OrOptArgs &A, bool dm, int apply, hipStream_t s, int lds_budget)
{
    const uint32_t gx = or_opt_grid_x(A.n), gy = or_opt_grid_y(A.n), nblocks = gx * gy;
    const uint32_t pg = (A.n + 255u) / 256u;
    if (dm) {
        hipLaunchKernelGGL(k_or_prepare<true>, dim3(pg), dim3(256), 0, s, A);
        hipLaunchKernelGGL(k_or_scan<true>, dim3(gx, gy), dim3(kOrWaves * 64), 0, s, A, or_opt_chunks(A.n));
    } else {
        hipLaunchKernelGGL(k_or_prepare<false>, dim3(pg), dim3(256), 0, s, A);
        hipLaunchKernelGGL(k_or_scan<false>, dim3(gx, gy), dim3(kOrWaves * 64), 0, s, A, or_opt_chunks(A.n));
    }
    hipError_t e = allow_max_lds(reinterpret_cast<const void *>(k_or_pick));
    if (e != hipSuccess) return e;
    // LDS up to 256 cities, the workspace beyond: both forms run in the parity tests at sizes the oracle affords
    const bool stage_lds = A.n <= 256u && (size_t)A.n * 4 + 1024 <= (size_t)lds_budget;
    hipLaunchKernelGGL(k_or_pick, dim3(1), dim3(1024), stage_lds ? (size_t)A.n * 4 : 0, s, A, nblocks, apply, stage_lds ? 1 : 0);
    return hipGetLastError();
}

uint32_t or_opt_scan_blocks(uint32_t n) { return or_opt_grid_x(n) * or_opt_grid_y(n); }

}  // namespace tl
