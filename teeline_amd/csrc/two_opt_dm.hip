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
// applies the reversal cooperatively and resumes at (i, j+1) — exactly the reference's loop order.  Where moves are frequent
// a step is ONE row dealt over the 16 waves; its hits are chained inside the owning wave's tile and their reversals deferred to
// the row's end and composed (round 4: the coordinate kernel's row structure — the next step needs only the new b and D[a][b],
// which come with the owner's hit list).
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

// ---- deferred reversals of a dense row (round 4: the coordinate kernel's row structure, two_opt_ref.hip flush_deferred, for a tour
// of u32 positions and its edge-length array).  The hits (i, g_0 < g_1 < ... < g_{k-1}) of ONE row all reverse a prefix that
// starts at lo = i + 1 (two_opt.rs:50 swap_2opt(path, i+1, j)); the rest of the row's scan reads only positions > g and the
// row's new b and D[a][b] — which the owner of the hits hands on in its list — so nothing inside [lo..g_last] is looked at before
// the row ends and the k reversals are applied at once.  With S_0 = [lo..g_0], S_m = [g_{m-1}+1..g_m] their composition is
//     rev(S_{k-1}) rev(S_{k-3}) ... | ... S_{k-4} S_{k-2}
// every element moves once.  The edge array moves with it: an edge inside a segment keeps its length and lands beside its
// left end's new place (one lower if the segment is reversed); the edge behind hit column g_m is the one that hit removed; and
// the k + 1 new edges are the D[b][e] each hit's lane held — hit m's joins source positions (m == 0 ? lo : g_{m-1}) and
// g_m + 1, adjacent in the result — and the last hit's D[a][c] in front of lo.  (Checked against hit-by-hit reversals over
// random hit sets before it was written: NOTEBOOK.md round 4.)
namespace {
constexpr uint32_t kDmPend = 64;   // deferred hits of one row (lane m of a flush holds hit m)
constexpr int kDmFlushSlots = 8;   // elements per thread a composed flush holds in registers: regions up to 8 x 1024 positions;
                                   // longer ones (n > 8 K and a hit that far out) are reversed hit by hit

__device__ __forceinline__ uint32_t rdl(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); }
template <int CTRL>
__device__ __forceinline__ uint32_t dm_dpp(uint32_t v)  // a row shift inside rows of 16 lanes; lanes without a source read 0
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);
}

// every wave calls this; the caller has put a barrier behind the writes of pend_j / pend_b.  Ends with a barrier.
__device__ __forceinline__ void dm_flush(uint32_t *perm, float *edge, const uint32_t *pend_j, const float *pend_b, float dac_last, uint32_t lo,
                                         uint32_t k, uint32_t tid, uint32_t lane, uint32_t wave)
{
    const bool have = lane < k;
    const uint32_t g = have ? pend_j[lane] : 0xFFFFFFFFu;
    const float pb = have ? pend_b[lane] : 0.0f;
    const uint32_t ghi = rdl(g, k - 1u);
    if (ghi - lo + 1u > (uint32_t)kDmFlushSlots * kDmNT) {
        // a region beyond the register budget: hit by hit, in the order the reference applies them
        for (uint32_t h = 0; h < k; ++h) {
            const uint32_t hi = rdl(g, h);
            const uint32_t half = (hi - lo + 1u) >> 1;
            for (uint32_t t = tid; t < half; t += kDmNT) {
                const uint32_t u = perm[lo + t], v = perm[hi - t];
                perm[lo + t] = v;
                perm[hi - t] = u;
            }
            const uint32_t ehalf = (hi - lo) >> 1;
            for (uint32_t t = tid; t < ehalf; t += kDmNT) {
                const float x = edge[lo + t], y = edge[hi - 1u - t];
                edge[lo + t] = y;
                edge[hi - 1u - t] = x;
            }
            if (tid == 0) edge[hi] = __builtin_bit_cast(float, rdl(__builtin_bit_cast(uint32_t, pb), h));
            TL_SYNC();
        }
        if (tid == 0) edge[lo - 1u] = dac_last;
        TL_SYNC();
        return;
    }
    // segment table, lane m = segment m (DPP shifts inside rows of 16 lanes, the row seams patched: a ds_bpermute chain here
    // was ~1 k cycles of every flush)
    uint32_t gprev = dm_dpp<0x111>(g);  // row_shr:1
    gprev = lane == 16u ? rdl(g, 15) : gprev;
    gprev = lane == 32u ? rdl(g, 31) : gprev;
    gprev = lane == 48u ? rdl(g, 47) : gprev;
    gprev = lane == 0u ? lo - 1u : gprev;
    const uint32_t start = gprev + 1u;
    const uint32_t len = have ? g - gprev : 0u;
    uint32_t pre = len;  // inclusive prefix over the lanes of the same parity: stride-2 scan inside each row of 16, then the rows below
    pre += dm_dpp<0x112>(pre);
    pre += dm_dpp<0x114>(pre);
    pre += dm_dpp<0x118>(pre);
    if (k > 16u) {
        const uint32_t e0 = rdl(pre, 14), o0 = rdl(pre, 15);
        const uint32_t e1 = e0 + rdl(pre, 30), o1 = o0 + rdl(pre, 31);
        const uint32_t e2 = e1 + rdl(pre, 46), o2 = o1 + rdl(pre, 47);
        const bool odd = (lane & 1u) != 0u;
        uint32_t off = 0u;
        off = lane >= 16u ? (odd ? o0 : e0) : off;
        off = lane >= 32u ? (odd ? o1 : e1) : off;
        off = lane >= 48u ? (odd ? o2 : e2) : off;
        pre += off;
    }
    const uint32_t t_rev = rdl(pre, k - 1u);  // total length of the reversed group
    const bool isrev = ((k - 1u - lane) & 1u) == 0u;
    const uint32_t base = lo + (isrev ? t_rev - pre : t_rev + pre - len);
    const uint32_t cst = isrev ? base + g : base - start;  // target = cst - p (reversed) or cst + p (kept)
    const uint32_t wfirst = lo + (wave << 6);
    uint32_t val[kDmFlushSlots], pk[kDmFlushSlots];  // position id; target | edge moves << 31 | segment reversed << 30
    float ev[kDmFlushSlots];
#pragma unroll
    for (int q = 0; q < kDmFlushSlots; ++q) {
        val[q] = 0u;
        pk[q] = 0u;
        ev[q] = 0.0f;
    }
#pragma unroll
    for (int q = 0; q < kDmFlushSlots; ++q) {
        const uint32_t w0 = wfirst + (uint32_t)(q * kDmNT);
        if (w0 > ghi) break;
        const uint32_t p = w0 + lane;
        const uint32_t w1 = (w0 + 63u < ghi) ? w0 + 63u : ghi;
        uint32_t m = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(g < w0));
        const uint32_t mhi = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(g < w1));
        bool rv = ((k - 1u - m) & 1u) == 0u;
        const uint32_t c0 = rdl(cst, m);
        uint32_t dst = rv ? c0 - p : c0 + p, gend = rdl(g, m);
        bool rvl = rv;
        for (++m; m <= mhi; ++m) {
            const uint32_t sm = rdl(start, m), cm = rdl(cst, m), gm = rdl(g, m);
            rv = ((k - 1u - m) & 1u) == 0u;
            const bool in = p >= sm;
            dst = in ? (rv ? cm - p : cm + p) : dst;
            gend = in ? gm : gend;
            rvl = in ? rv : rvl;
        }
        if (p <= ghi) {
            val[q] = perm[p];
            const bool moves = p != gend;  // the edge behind a hit column is the one that hit removed
            if (moves) ev[q] = edge[p];
            pk[q] = dst | (moves ? 0x80000000u : 0u) | (rvl ? 0x40000000u : 0u);
        } else {
            pk[q] = 0xFFFFFFFFu;
        }
    }
    // the k + 1 new edges (wave 0, lane m = hit m)
    uint32_t cst_dn = dm_dpp<0x111>(cst), cst_up = dm_dpp<0x101>(cst);  // row_shr:1 / row_shl:1, seams patched
    cst_dn = lane == 16u ? rdl(cst, 15) : cst_dn;
    cst_dn = lane == 32u ? rdl(cst, 31) : cst_dn;
    cst_dn = lane == 48u ? rdl(cst, 47) : cst_dn;
    cst_up = lane == 15u ? rdl(cst, 16) : cst_up;
    cst_up = lane == 31u ? rdl(cst, 32) : cst_up;
    cst_up = lane == 47u ? rdl(cst, 48) : cst_up;
    TL_SYNC();
#pragma unroll
    for (int q = 0; q < kDmFlushSlots; ++q) {
        const uint32_t w0 = wfirst + (uint32_t)(q * kDmNT);
        if (w0 > ghi) break;
        if (pk[q] != 0xFFFFFFFFu) {
            const uint32_t dst = pk[q] & 0x3FFFFFFFu;
            perm[dst] = val[q];
            if (pk[q] & 0x80000000u) edge[(pk[q] & 0x40000000u) ? dst - 1u : dst] = ev[q];
        }
    }
    if (wave == 0u && have) {
        if (lane + 1u < k) {
            const uint32_t bsrc = lane == 0u ? lo : gprev, sb = lane == 0u ? 0u : lane - 1u;
            const uint32_t cb = lane == 0u ? cst : cst_dn;
            const uint32_t db = (((k - 1u - sb) & 1u) == 0u) ? cb - bsrc : cb + bsrc;
            const uint32_t de = (((k - 2u - lane) & 1u) == 0u) ? cst_up - (g + 1u) : cst_up + (g + 1u);
            edge[db < de ? db : de] = pb;
        } else {
            edge[ghi] = pb;          // (b_{k-1}, e_{k-1})
            edge[lo - 1u] = dac_last;  // (a, c_{k-1})
        }
    }
    TL_SYNC();
}
}  // namespace

// STAGE (small tours: 32 matrix rows fit the LDS next to the tour, n <= ~1170 — pr1002 does): in a wide block every wave
// copies its two matrix rows a and b into LDS with coalesced loads and gathers D[a][c], D[b][e] from there; a fully
// divergent global gather costs the CU's address unit ~64 cycles per wave instruction, an LDS gather a few.
template <bool STAGE>
__global__ __launch_bounds__(kDmNT) void k_two_opt_ref_dm(TwoOptBatchArgs A)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t n = A.n;
    const uint32_t nq = (n + 1u + 3u) & ~3u;
    constexpr uint32_t NWv = kDmNT / 64;
    uint32_t *perm = reinterpret_cast<uint32_t *>(smem);            // n + 1 (pad)
    float *edge = reinterpret_cast<float *>(perm + nq);             // edge[j] = D[perm[j]][perm[j+1]], j < n-1
    uint32_t *keys = reinterpret_cast<uint32_t *>(edge + nq);       // 4 slots (3 rotate with the step)
    // per (step parity, wave): the improving moves it chained in its tile — header {count, resume column, the row's new b (a city),
    // D[a][new b] = the last hit lane's D[a][c]}, the hit columns and each hit's D[b][e] (the new edge behind that hit).  A list is
    // read behind its step's barrier while its owner may already write the next step's: the parities alternate.
    uint4 *hdr = reinterpret_cast<uint4 *>(keys + 4);               // [2][NW]
    uint32_t *hl_j = reinterpret_cast<uint32_t *>(hdr + 2 * NWv);   // [2][NW][kDmChain]
    float *hl_b = reinterpret_cast<float *>(hl_j + 2 * NWv * kDmChain);
    uint32_t *pend_j = reinterpret_cast<uint32_t *>(hl_b + 2 * NWv * kDmChain);  // [kDmPend] deferred hit columns of the current dense row
    float *pend_b = reinterpret_cast<float *>(pend_j + kDmPend);                 // ... and their D[b][e]
    float *rowbuf = pend_b + kDmPend;                               // STAGE: per wave two matrix rows of nq floats
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
    uint64_t rev_lane = 0;  // wave 0: per-lane share of `reversed` from the dense rows' hits (summed at the end)
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    // optional move log (tl_two_opt_trace): (i << 16) | j per applied move in the reference's order, 0xFFFFFFFF where a new sweep begins
    uint32_t *mlog = A.move_log ? A.move_log + (size_t)d * A.log_cap : nullptr;
    uint32_t log_n = 0;
    uint32_t since_rows = 0;  // rows scanned since the last move
    uint32_t gap_rows = 0;    // ... and its running average over the recent moves: the block shape follows the larger of the two
    // dense rows: deferred hits (columns in pend_j), the row's a and current b (cities) and D[a][b]; row_valid: they are loaded
    uint32_t np = 0, ra = 0, rb = 0;
    float rdab = 0.0f, dac_last = 0.0f;
    bool row_valid = false;
#ifdef TL_DM_PROFILE
    // wave 0's shader cycles by phase, dense / wide steps apart: [0] step top -> scan done, [1] wait at the barrier, [2] boundary without
    // flushes, [3] flushes, [4] steps, [5] flushes; wide from [8]
    uint64_t qd[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t tq = __builtin_amdgcn_s_memtime();
    bool pw = false;
#define TL_DSTAMP(k) do { const uint64_t t_ = __builtin_amdgcn_s_memtime(); qd[(pw ? 8 : 0) + (k)] += t_ - tq; tq = t_; } while (0)
#define TL_DCOUNT(k) do { qd[(pw ? 8 : 0) + (k)] += 1; } while (0)
#else
#define TL_DSTAMP(k) do { } while (0)
#define TL_DCOUNT(k) do { } while (0)
#endif

    while (n >= 4) {
        const uint32_t slot = step % 3u, par = step & 1u;
        if (tid == 0) keys[(step + 1u) % 3u] = kNoKey;
        ++step;
        // Two block shapes, chosen from the observed gap between moves (like the coordinate kernel):
        //  wide  — moves are rare: 16 rows per step, one per wave, lanes along j;
        //  dense — moves every few rows: one row per step, its columns dealt to the 16 waves, hits deferred to the row's end.
        // Either way the lexicographically first improving (i, j) wins (ds_min_u32 on i << 16 | j) and the scan
        // resumes at (i, j+1) like the reference.
        const bool wide = (since_rows > gap_rows ? since_rows : gap_rows) >= 4u;
#ifdef TL_DM_PROFILE
        pw = wide;
        tq = __builtin_amdgcn_s_memtime();
#endif
        if (wide && np) {  // the shape changes in the middle of a row: its deferred reversals first
            TL_SYNC();     // (the hits filed at the last boundary)
            dm_flush(perm, edge, pend_j, pend_b, dac_last, i0 + 1u, np, tid, lane, wave);
            np = 0;
        }
        if (wide) row_valid = false;
        const uint32_t R = wide ? NWv : 1u;
        const uint32_t jbase = j0 - (j0 & 63u);
        uint4 *my_hdr = hdr + par * NWv + wave;
        uint32_t *my_j = hl_j + (par * NWv + wave) * kDmChain;
        float *my_b = hl_b + (par * NWv + wave) * kDmChain;
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
                                lds_min_u32(&keys[slot], (i << 16) | j);  // (one lane; not through the atomic optimiser's lane scan, tl_device.h;
                                my_j[0] = j;                              //  the tracked LDS writes behind it carry the wait in front of the barrier)
                                my_b[0] = dbe[u];
                                *my_hdr = make_uint4(1u, j + 1u, 0u, __builtin_bit_cast(uint32_t, dac[u]));
                            }
                            done = true;
                        }
                    }
                }
            }
        } else {
            const uint32_t i = i0;
            if (!row_valid) {  // a = p[i], b = p[i+1]; behind a step with hits both the new b and D[a][b] came with the owner's list
                ra = (uint32_t)__builtin_amdgcn_readfirstlane((int)perm[i]);
                rb = (uint32_t)__builtin_amdgcn_readfirstlane((int)perm[i + 1u]);
                rdab = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, edge[i])));
                row_valid = true;
            }
            const float dab = rdab;
            const float *__restrict__ rowa = dm + (size_t)ra * n;
            const float *__restrict__ rowb = dm + (size_t)rb * n;
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
                if (m) {
                    // Chain every improving move of the reference's scan inside this tile: after a hit at lane l the row's b is
                    // the old perm[j] (two_opt.rs:50 reverses p[i+1..=j]), positions > j are untouched, so the lanes > l are
                    // decided again with D[b'][e] gathered from the new b's matrix row and D[a][b'] = the hit lane's D[a][c].
                    // The reversals themselves wait until the row ends (they all start at i+1).
                    uint32_t nh = 0, jh = 0, bn = 0;
                    for (;;) {
                        const uint32_t l = (uint32_t)(__builtin_ffsll((long long)m) - 1);
                        jh = jb + l;
                        if (lane == l) {
                            if (nh == 0) lds_min_u32(&keys[slot], (i << 16) | jh);
                            my_j[nh] = jh;
                            my_b[nh] = dbe;
                        }
                        ++nh;
                        bn = (uint32_t)__builtin_amdgcn_readlane((int)c, (int)l);
                        dabc = readlane_f(dac, (int)l);
                        if (nh >= kDmChain || l == 63u || jh + 1u > n - 2u) break;
                        dbe = dm[(size_t)bn * n + e];
                        imp = (j > jh) & (j <= n - 2u) & (dac + dbe < dabc + dce);
                        m = __builtin_amdgcn_ballot_w64(imp);
                        if (!m) break;
                    }
                    // count, the column at which the scan resumes (the tile is exhausted unless the chain was cut short), the
                    // row's new b and D[a][b] = the last hit lane's D[a][c]
                    if (lane == 0) *my_hdr = make_uint4(nh, nh >= kDmChain ? jh + 1u : jb + 64u, bn, __builtin_bit_cast(uint32_t, dabc));
                    break;
                }
            }
        }
        TL_DSTAMP(0);
        TL_SYNC();
        TL_DSTAMP(1);
        TL_DCOUNT(4);
        const uint32_t key = (uint32_t)__builtin_amdgcn_readfirstlane((int)keys[slot]);
        if (key == kNoKey) {
            if (np) {  // the dense row is finished: its reversals are due, composed
                TL_DSTAMP(2);
                dm_flush(perm, edge, pend_j, pend_b, dac_last, i0 + 1u, np, tid, lane, wave);
                TL_DSTAMP(3);
                TL_DCOUNT(5);
                np = 0;
            }
            i0 += R;
            j0 = i0 + 2u;
            since_rows += R;
            row_valid = false;
        } else {
            const uint32_t is = key >> 16, js = key & 0xFFFFu;
            gap_rows = (gap_rows + since_rows + (is - i0) + 1u) >> 1;  // running estimate of the rows between moves
            since_rows = 0;
            // the wave that posted the winning key: its row in a wide block, its tile of the row in a dense one
            const uint32_t ww = wide ? is - i0 : ((js - jbase) >> 6) & (NWv - 1u);
            const uint4 hv = hdr[par * NWv + ww];
            const uint32_t nh = (uint32_t)__builtin_amdgcn_readfirstlane((int)hv.x);
            const uint32_t resume = (uint32_t)__builtin_amdgcn_readfirstlane((int)hv.y);
            const uint32_t *wj = hl_j + (par * NWv + ww) * kDmChain;
            const float *wb = hl_b + (par * NWv + ww) * kDmChain;
            const uint32_t lo = is + 1u;
            if (wide) {
                // one move, applied at once: swap_2opt(path, i+1, j), two_opt.rs:69-79
                const uint32_t hi = js;
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
                    edge[lo - 1u] = __builtin_bit_cast(float, hv.w);
                    edge[hi] = wb[0];
                    if (mlog && log_n < A.log_cap) mlog[log_n] = (is << 16) | hi;
                }
                reversed += (uint64_t)(hi - is);
                TL_SYNC();
            } else {
                // dense: the hits wait in pend_* until the row ends; the scan goes on at `resume` with the new b and D[a][b]
                if (wave == 0u && lane < nh) {
                    const uint32_t g = wj[lane];
                    pend_j[np + lane] = g;
                    pend_b[np + lane] = wb[lane];
                    rev_lane += (uint64_t)(g - is);
                    if (mlog && log_n + lane < A.log_cap) mlog[log_n + lane] = (is << 16) | g;
                }
                np += nh;
                rb = (uint32_t)__builtin_amdgcn_readfirstlane((int)hv.z);
                dac_last = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane((int)hv.w));
                rdab = dac_last;
            }
            log_n += nh;
            improved = true;
            moves += nh;
            i0 = is;
            j0 = resume;
            const bool next_row = j0 > n - 2u;
            if (!wide && (next_row || np + kDmChain > kDmPend)) {
                TL_DSTAMP(2);
                TL_SYNC();  // the hits filed just now
                dm_flush(perm, edge, pend_j, pend_b, dac_last, i0 + 1u, np, tid, lane, wave);
                TL_DSTAMP(3);
                TL_DCOUNT(5);
                np = 0;
            }
            if (next_row) {
                ++i0;
                j0 = i0 + 2u;
                row_valid = false;
            }
        }
        TL_DSTAMP(2);
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
            row_valid = false;
            if (mlog && tid == 0 && log_n < A.log_cap) mlog[log_n] = 0xFFFFFFFFu;
            log_n += 1u;
        }
    }

    uint32_t *__restrict__ out = A.out_pos + (size_t)d * n;
    for (uint32_t k = tid; k < n; k += kDmNT) out[k] = perm[k];

    // `reversed` of the dense rows: per-lane shares of wave 0, summed
    if (wave == 0u) {
#pragma unroll
        for (int sft = 32; sft >= 1; sft >>= 1) {
            const uint32_t lo32 = (uint32_t)__shfl_xor((int)(uint32_t)rev_lane, sft), hi32 = (uint32_t)__shfl_xor((int)(uint32_t)(rev_lane >> 32), sft);
            rev_lane += ((uint64_t)hi32 << 32) | lo32;
        }
        reversed += rev_lane;
    }
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
        printf("dmprof dense: scan %lu wait %lu boundary %lu flush %lu | steps %lu flushes %lu || wide: scan %lu wait %lu boundary %lu | steps %lu\n", qd[0], qd[1], qd[2], qd[3],
               qd[4], qd[5], qd[8], qd[9], qd[10], qd[12]);
#endif
    }
}

size_t two_opt_ref_dm_lds_bytes(uint32_t n)
{
    // tour + edge lengths, key slots, hit lists of both step parities (header, columns, D[b][e]), the row's deferred hits
    return (size_t)((n + 1u + 3u) & ~3u) * 8 + 16 + (size_t)2 * (kDmNT / 64) * (16 + kDmChain * 8) + (size_t)kDmPend * 8;
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
