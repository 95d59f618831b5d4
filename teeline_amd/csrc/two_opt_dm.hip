// two_opt_dm.hip — REF_ORDER 2-opt (src/tsp/two_opt.rs:26-61) with every distance taken from the caller's matrix
// (the reference's packed lower triangle, distance_matrix.rs:177-191) resident in HBM/L2.
//
// Used when the caller supplies `dm_packed` (EXPLICIT / GEO problems, and north-star config 2:
// "fp32 distance matrix in HBM").  MI355X layout: the packed triangle is expanded once per call into a full
// row-major n x n matrix in the context's workspace (k_dm_expand_full; 4 MB at n = 1002, sized for 288 GB of HBM),
// because a row scan gathers D[a][perm[j]] for one a and all j: in the full matrix those 64 gathers of a wave fall
// into the 4n bytes of row a (L1/L2-resident after the first touch), in the packed triangle every column > a sits in
// a different cache line.  One persistent workgroup per descent; in LDS: the tour (u32 positions) and the lengths of
// its edges, edge[j] = D[perm[j]][perm[j+1]] (the D[c][e] term of every candidate; a reversal reverses that array
// too and its two new boundary edges are the D[a][c], D[b][e] the winning lane already holds).
// A step speculatively decides 16 rows (one per wave, lanes along j, four 64-column tiles of gathers in flight) under
// "no move yet", reduces the lexicographically first improving (i,j) (ballot/ffs per wave, ds_min_u32 on i<<16|j),
// applies the reversal cooperatively and resumes at (i, j+1) — exactly the reference's loop order.
// Algorithmic bytes per candidate: perm[j+1] 4 B + D[a][c] 4 B + D[b][e] 4 B + D[c][e] 4 B = 16 B
// (SURVEY.md §8(d)); the row terms a, b, D[a][b] are amortised over the row.
#include "tl_kernels.h"

#pragma clang fp contract(off)

namespace tl {

namespace {
constexpr uint32_t kNoKey = 0xFFFFFFFFu;
constexpr int kDmNT = 1024;
#ifndef TL_DM_TILES
#define TL_DM_TILES 4
#endif
constexpr int kDmTiles = TL_DM_TILES;  // 64-column tiles whose gathers are in flight together (wide mode)
#ifndef TL_DM_CHAIN
#define TL_DM_CHAIN 16
#endif
constexpr uint32_t kDmChain = TL_DM_CHAIN;  // improving moves one wave may chain inside its tile of a dense row
#ifndef TL_DM_WARM_MB
#define TL_DM_WARM_MB 32
#endif
constexpr size_t kDmWarmBytes = (size_t)TL_DM_WARM_MB << 20;  // matrices up to this size are read once at the start of a descent (L2 / MALL warm-up)
}

// packed strict lower triangle (idx(r > c) = r(r-1)/2 + c) -> full symmetric row-major n x n, zero diagonal
// (distance_by_pos returns 0.0 for equal positions, distance_matrix.rs:181-183).  One workgroup per 64 x 64 tile of the
// lower triangle: coalesced reads along a packed row, the mirrored tile through an LDS transpose.
__global__ __launch_bounds__(256) void k_dm_expand_full(const float *__restrict__ packed, uint32_t n, float *__restrict__ full)
{
    __shared__ float tile[64][65];
    const uint32_t tr = blockIdx.y, tc = blockIdx.x;
    if (tc > tr) return;
    const uint32_t lx = threadIdx.x & 63u, ly = threadIdx.x >> 6;  // 64 x 4
    for (uint32_t rr = ly; rr < 64u; rr += 4u) {
        const uint32_t r = tr * 64u + rr, c = tc * 64u + lx;
        float v = 0.0f;
        if (r < n && c < r) v = packed[(size_t)r * (r - 1u) / 2u + c];
        tile[rr][lx] = v;
        if (r < n && c < n && c <= r) full[(size_t)r * n + c] = v;
    }
    TL_SYNC();
    for (uint32_t rr = ly; rr < 64u; rr += 4u) {
        // mirrored element: full[c'][r'] with c' = tc*64 + rr (row of the upper part), r' = tr*64 + lx
        const uint32_t cu = tc * 64u + rr, ru = tr * 64u + lx;
        if (cu < n && ru < n && cu < ru) full[(size_t)cu * n + ru] = tile[lx][rr];
    }
}

// STAGE (small tours: 32 matrix rows fit the LDS next to the tour, n <= ~1170 — pr1002 does): in a wide block every wave
// copies its two matrix rows a and b into LDS with coalesced loads and gathers D[a][c], D[b][e] from there; a fully
// divergent global gather costs the CU's address unit ~64 cycles per wave instruction, an LDS gather a few.
template <bool STAGE>
__global__ __launch_bounds__(kDmNT) void k_two_opt_ref_dm(TwoOptBatchArgs A)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t n = A.n;
    const uint32_t nq = (n + 1u + 3u) & ~3u;
    uint32_t *perm = reinterpret_cast<uint32_t *>(smem);            // n + 1 (pad)
    float *edge = reinterpret_cast<float *>(perm + nq);             // edge[j] = D[perm[j]][perm[j+1]], j < n-1
    uint32_t *keys = reinterpret_cast<uint32_t *>(edge + nq);       // 4 slots
    // per wave: the improving moves it chained in its tile — column, D[a][c] and D[b][e] of the hit lane (the two new boundary
    // edges of that reversal) — their count and the column at which the scan resumes
    uint32_t *hl_j = keys + 4;                                      // [NW][kDmChain]
    float *hl_a = reinterpret_cast<float *>(hl_j + (kDmNT / 64) * kDmChain);
    float *hl_b = hl_a + (kDmNT / 64) * kDmChain;
    uint32_t *hcnt = reinterpret_cast<uint32_t *>(hl_b + (kDmNT / 64) * kDmChain);  // [NW]
    uint32_t *hres = hcnt + (kDmNT / 64);                           // [NW]
    float *rowbuf = reinterpret_cast<float *>(hres + (kDmNT / 64)); // STAGE: per wave two matrix rows of nq floats
    const float *__restrict__ dm = A.dm_full;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t d = blockIdx.x;

    if (A.init_mode == TL_INIT_ARRAY) {
        const uint32_t *__restrict__ src = A.init + (size_t)d * n;
        for (uint32_t k = tid; k < n; k += kDmNT) perm[k] = src[k];
    } else {
        for (uint32_t k = tid; k < n; k += kDmNT) perm[k] = k;  // two_opt.rs:18-20
    }
    if (tid == 0) perm[n] = 0;
    if (tid < 4) keys[tid] = kNoKey;
    TL_SYNC();
    for (uint32_t k = tid; k + 1u < n; k += kDmNT) edge[k] = dm[(size_t)perm[k] * n + perm[k + 1u]];
    if ((size_t)n * n * 4u <= kDmWarmBytes) {
        // The descent touches a matrix row for the first time almost every step of its first sweep (row a of every i, the row of
        // every move's new b), and a first touch is a miss of this XCD's L2 — k_dm_expand_full ran on all of them — i.e. a round
        // trip to the Infinity Cache / HBM in front of a step.  So stream the matrix through once (4 MB at n = 1002: ~30 us).
        const float4 *__restrict__ m4 = reinterpret_cast<const float4 *>(dm);
        const size_t n4 = (size_t)n * n / 4u;
        float acc = 0.0f;
        for (size_t k = tid; k < n4; k += kDmNT) {
            const float4 v = m4[k];
            acc += (v.x + v.y) + (v.z + v.w);
        }
        if (acc == -1.0f) keys[3] = 0u;  // distances are >= 0: never taken, keeps the loads
    }
    TL_SYNC();

    const uint32_t nrows = n - 3;
    uint32_t i0 = 0, j0 = 2, step = 0, sweeps = 1, status = 0;
    bool improved = false;
    uint64_t moves = 0, reversed = 0;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    constexpr uint32_t NWv = kDmNT / 64;
    // optional move log (tl_two_opt_trace): (i << 16) | j per applied move in the reference's order, 0xFFFFFFFF where a new sweep begins
    uint32_t *mlog = A.move_log ? A.move_log + (size_t)d * A.log_cap : nullptr;
    uint32_t log_n = 0;
    uint32_t since_rows = 0;  // rows scanned since the last move
    uint32_t gap_rows = 0;    // ... and its running average over the recent moves: the block shape follows the larger of the two

#ifdef TL_DM_PROFILE
    uint64_t qd[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // wave 0's cycles in dense steps: row set-up, first decision, chain, barrier wait, boundary + reversals, [5] steps; wide steps from [8]: staging, -, scan, ...
    uint64_t tq = __builtin_amdgcn_s_memtime();
#define TL_DSTAMP(k) do { const uint64_t t_ = __builtin_amdgcn_s_memtime(); qd[(wide ? 8 : 0) + (k)] += t_ - tq; tq = t_; } while (0)
#else
#define TL_DSTAMP(k) do { } while (0)
#endif
    while (n >= 4) {
        const uint32_t slot = step % 3u;
        if (tid == 0) keys[(step + 1u) % 3u] = kNoKey;
        ++step;
        // Two block shapes, chosen from the observed gap between moves (like the coordinate kernel):
        //  wide  — moves are rare: 16 rows per step, one per wave, lanes along j;
        //  dense — moves every few rows: one row per step, its columns dealt to the 16 waves.
        // Either way the lexicographically first improving (i, j) wins (ds_min_u32 on i << 16 | j) and the scan
        // resumes at (i, j+1) like the reference.
        const bool wide = (since_rows > gap_rows ? since_rows : gap_rows) >= 4u;
        const uint32_t R = wide ? NWv : 1u;
        const uint32_t jbase = j0 - (j0 & 63u);
        if (wide) {
            const uint32_t i = i0 + wave;
            if (i < nrows) {
                const uint32_t a = perm[i], b = perm[i + 1u];
                const float dab = edge[i];
                const float *__restrict__ rowa = dm + (size_t)a * n;
                const float *__restrict__ rowb = dm + (size_t)b * n;
                float *la = rowbuf + (size_t)(2u * wave) * nq, *lb = la + nq;
                if (STAGE) {
                    for (uint32_t c0 = lane; c0 < n; c0 += 256u) {  // four coalesced loads of each row in flight
                        float va[4], vb[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const uint32_t c = c0 + 64u * (uint32_t)u;
                            va[u] = c < n ? rowa[c] : 0.0f;
                            vb[u] = c < n ? rowb[c] : 0.0f;
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const uint32_t c = c0 + 64u * (uint32_t)u;
                            if (c < n) {
                                la[c] = va[u];
                                lb[c] = vb[u];
                            }
                        }
                    }
                }
                TL_DSTAMP(0);
                const uint32_t jmin = wave == 0u ? j0 : i + 2u;
                bool done = false;
                for (uint32_t jb = jmin - (jmin & 63u); jb <= n - 2u && !done; jb += 64u * kDmTiles) {
                    const uint32_t kb = (uint32_t)__builtin_amdgcn_readfirstlane((int)keys[slot]);
                    if (kb != kNoKey && ((kb >> 16) < i || ((kb >> 16) == i && (kb & 0xFFFFu) < jb))) break;  // an earlier hit exists
                    float dac[kDmTiles], dbe[kDmTiles], dce[kDmTiles];
#pragma unroll
                    for (int u = 0; u < kDmTiles; ++u) {
                        const uint32_t j = jb + 64u * (uint32_t)u + lane;
                        const uint32_t jj = j <= n - 2u ? j : n - 2u;  // lanes beyond the row read a valid column and are masked below
                        const uint32_t c = perm[jj], e = perm[jj + 1u];
                        dac[u] = STAGE ? la[c] : rowa[c];
                        dbe[u] = STAGE ? lb[e] : rowb[e];
                        dce[u] = edge[jj];
                    }
#pragma unroll
                    for (int u = 0; u < kDmTiles; ++u) {
                        const uint32_t j = jb + 64u * (uint32_t)u + lane;
                        const float cur = dab + dce[u];                    // two_opt.rs:35-40
                        const float neu = dac[u] + dbe[u];                 // two_opt.rs:42-47
                        const bool imp = (j >= jmin) & (j <= n - 2u) & (neu < cur);  // :49
                        const uint64_t m = __builtin_amdgcn_ballot_w64(imp);
                        if (m && !done) {
                            const uint32_t l = (uint32_t)(__builtin_ffsll((long long)m) - 1);
                            if (lane == l) {
                                lds_min_u32(&keys[slot], (i << 16) | j);  // (one lane; not through the atomic optimiser's lane scan, tl_device.h)
                                hl_j[wave * kDmChain] = j;
                                hl_a[wave * kDmChain] = dac[u];
                                hl_b[wave * kDmChain] = dbe[u];
                                hcnt[wave] = 1u;
                                hres[wave] = j + 1u;
                            }
                            done = true;
                        }
                    }
                }
            }
        } else {
            const uint32_t i = i0;
            const uint32_t a = perm[i], b = perm[i + 1u];
            const float dab = edge[i];
            const float *__restrict__ rowa = dm + (size_t)a * n;
            const float *__restrict__ rowb = dm + (size_t)b * n;
            TL_DSTAMP(0);
            for (uint32_t jb = jbase + (wave << 6); jb <= n - 2u; jb += kDmNT) {
                const uint32_t kb = (uint32_t)__builtin_amdgcn_readfirstlane((int)keys[slot]);
                if (kb != kNoKey && (kb & 0xFFFFu) < jb) break;
                const uint32_t j = jb + lane;
                const uint32_t jj = j <= n - 2u ? j : n - 2u;
                const uint32_t c = perm[jj], e = perm[jj + 1u];
                const float dac = rowa[c], dce = edge[jj];
                float dbe = rowb[e];
                float dabc = dab;
                bool imp = (j >= j0) & (j <= n - 2u) & (dac + dbe < dabc + dce);
                uint64_t m = __builtin_amdgcn_ballot_w64(imp);
                TL_DSTAMP(1);
                if (m) {
                    // Chain every improving move of the reference's scan inside this tile: after a hit at lane l the row's b is
                    // the old perm[j] (two_opt.rs:50 reverses p[i+1..=j]), positions > j are untouched, so the lanes > l are
                    // decided again with D[b'][e] gathered from the new b's matrix row and D[a][b'] = the hit lane's D[a][c].
                    // The reversals themselves wait until the step's barrier (they all start at i+1).
                    uint32_t nh = 0, jh = 0;
                    for (;;) {
                        const uint32_t l = (uint32_t)(__builtin_ffsll((long long)m) - 1);
                        jh = jb + l;
                        if (lane == l) {
                            if (nh == 0) lds_min_u32(&keys[slot], (i << 16) | jh);
                            hl_j[wave * kDmChain + nh] = jh;
                            hl_a[wave * kDmChain + nh] = dac;
                            hl_b[wave * kDmChain + nh] = dbe;
                        }
                        ++nh;
                        if (nh >= kDmChain || l == 63u || jh + 1u > n - 2u) break;
                        const uint32_t bn = (uint32_t)__builtin_amdgcn_readlane((int)c, (int)l);
                        dabc = readlane_f(dac, (int)l);
                        dbe = dm[(size_t)bn * n + e];
                        imp = (j > jh) & (j <= n - 2u) & (dac + dbe < dabc + dce);
                        m = __builtin_amdgcn_ballot_w64(imp);
                        if (!m) break;
                    }
                    if (lane == 0) {
                        hcnt[wave] = nh;
                        hres[wave] = nh >= kDmChain ? jh + 1u : jb + 64u;  // the tile is exhausted unless the chain was cut short
                    }
                    break;
                }
            }
        }
        TL_DSTAMP(2);
        TL_SYNC();
        TL_DSTAMP(3);
        const uint32_t key = (uint32_t)__builtin_amdgcn_readfirstlane((int)keys[slot]);
        if (key == kNoKey) {
            i0 += R;
            j0 = i0 + 2u;
            since_rows += R;
        } else {
            const uint32_t is = key >> 16, js = key & 0xFFFFu;
            gap_rows = (gap_rows + since_rows + (is - i0) + 1u) >> 1;  // running estimate of the rows between moves
            since_rows = 0;
            // the wave that posted the winning key: its row in a wide block, its tile of the row in a dense one
            const uint32_t ww = wide ? is - i0 : ((js - jbase) >> 6) & (NWv - 1u);
            const uint32_t nh = (uint32_t)__builtin_amdgcn_readfirstlane((int)hcnt[ww]);
            const uint32_t resume = (uint32_t)__builtin_amdgcn_readfirstlane((int)hres[ww]);
            const uint32_t lo = is + 1u;
            for (uint32_t h = 0; h < nh; ++h) {  // swap_2opt(path, i+1, j), two_opt.rs:69-79, in the order the reference applies them
                const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)hl_j[ww * kDmChain + h]);
                const uint32_t half = (hi - lo + 1u) >> 1;
                for (uint32_t t = tid; t < half; t += kDmNT) {
                    const uint32_t u = perm[lo + t], v = perm[hi - t];
                    perm[lo + t] = v;
                    perm[hi - t] = u;
                }
                // edges inside the segment keep their lengths in reversed order (the matrix is symmetric); the two boundary
                // edges become (a, c) and (b, e)
                const uint32_t ehalf = (hi - lo) >> 1;
                for (uint32_t t = tid; t < ehalf; t += kDmNT) {
                    const float x = edge[lo + t], y = edge[hi - 1u - t];
                    edge[lo + t] = y;
                    edge[hi - 1u - t] = x;
                }
                if (tid == 0) {
                    edge[lo - 1u] = hl_a[ww * kDmChain + h];
                    edge[hi] = hl_b[ww * kDmChain + h];
                }
                reversed += (uint64_t)(hi - is);
                if (mlog && tid == 0 && log_n + h < A.log_cap) mlog[log_n + h] = (is << 16) | hi;
                TL_SYNC();
            }
            log_n += nh;
            improved = true;
            moves += nh;
            i0 = is;
            j0 = resume;
            if (j0 > n - 2u) {
                ++i0;
                j0 = i0 + 2u;
            }
        }
        TL_DSTAMP(4);
#ifdef TL_DM_PROFILE
        qd[(wide ? 8 : 0) + 5] += 1;
#endif
        if (i0 >= nrows) {
            if (!improved) break;
            if (sweeps >= A.max_sweeps) {
                status = 1;
                break;
            }
            improved = false;
            ++sweeps;
            i0 = 0;
            j0 = 2;
            if (mlog && tid == 0 && log_n < A.log_cap) mlog[log_n] = 0xFFFFFFFFu;
            log_n += 1u;
        }
    }

    uint32_t *__restrict__ out = A.out_pos + (size_t)d * n;
    for (uint32_t k = tid; k < n; k += kDmNT) out[k] = perm[k];

    // tour_length_by_pos (distance_matrix.rs:235-245): closing edge first, then the windows, sequential f32 sum
    if (tid == 0) {
        float total = 0.0f;
        if (n >= 2) total = dm[(size_t)perm[n - 1] * n + perm[0]];
        for (uint32_t k = 0; k + 1u < n; ++k) total += edge[k];
        A.out_cost[d] = total;
        uint64_t *st = A.out_stats + (size_t)d * TL_STATS_STRIDE;
        st[0] = sweeps;
        st[1] = moves;
#ifdef TL_DM_REPORT_STEPS
        st[2] = step;  // tuning builds: steps instead of reversed elements
#else
        st[2] = reversed;
#endif
        st[3] = status;
        st[4] = step;
        st[15] = log_n;  // words offered to the move log (moves + sweep marks)
#ifdef TL_DM_PROFILE
        printf("dmprof dense: setup %lu first %lu chain %lu wait %lu boundary %lu steps %lu | wide: staging %lu scan %lu wait %lu boundary %lu steps %lu\n", qd[0], qd[1], qd[2], qd[3], qd[4], qd[5], qd[8], qd[10], qd[11], qd[12], qd[13]);
#endif
    }
}

size_t two_opt_ref_dm_lds_bytes(uint32_t n)
{
    return (size_t)((n + 1u + 3u) & ~3u) * 8 + 16 + (size_t)(kDmNT / 64) * (kDmChain * 12 + 8);
}
static size_t two_opt_ref_dm_stage_bytes(uint32_t n) { return (size_t)((n + 1u + 3u) & ~3u) * 4 * 2 * (kDmNT / 64); }

hipError_t launch_dm_expand_full(const float *packed, uint32_t n, float *full, hipStream_t s)
{
    const uint32_t nt = (n + 63u) / 64u;
    hipLaunchKernelGGL(k_dm_expand_full, dim3(nt, nt), dim3(256), 0, s, packed, n, full);
    return hipGetLastError();
}

hipError_t launch_two_opt_ref_dm(const TwoOptBatchArgs &A, uint32_t count, int lds_budget, hipStream_t s)
{
    const size_t base = two_opt_ref_dm_lds_bytes(A.n), staged = base + two_opt_ref_dm_stage_bytes(A.n);
    const bool stage = staged <= (size_t)lds_budget;
    const size_t lds = stage ? staged : base;
    auto kern = stage ? k_two_opt_ref_dm<true> : k_two_opt_ref_dm<false>;
    hipError_t e = allow_max_lds(reinterpret_cast<const void *>(kern));
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(count), dim3(kDmNT), lds, s, A);
    return hipGetLastError();
}

}  // namespace tl
