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
#include "two_opt_common.h"

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
constexpr uint32_t kDmXl = 32;        // ... and entries of a city's cached "nearer than its successor" record
#ifndef TL_DM_LONG_CAP
#define TL_DM_LONG_CAP 1024  // (256 / 512 / 1024 measured: n = 1 002 alike, n = 5 000 population 148 / 134 / 131 ms)
#endif
constexpr uint32_t kDmLongCap = TL_DM_LONG_CAP;  // cities with a tour edge beyond their kDmK-th distance a descent can hold (late sweeps)
#ifndef TL_DM_LATE_ROWS
#define TL_DM_LATE_ROWS 2
#endif
constexpr uint32_t kDmLateRows = TL_DM_LATE_ROWS;  // rows of a late sweep a wave decides together (their look-ups in flight at once)
__device__ __forceinline__ uint32_t rl_u(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); }
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

// The lists of the late sweeps, cut from the full matrix: one wave per city walks its row kDmK times, each time taking the
// smallest (distance, id) beyond the last one taken — the row is 4n bytes, L1-resident after the first walk — and offers
// itself to the reverse list of every city it takes.  The order inside a reverse list depends on the order of the atomics;
// the lists are read as SETS (a row's first improving column is a minimum over them), so results do not.
// improving (two_opt.rs:49: D[a][c] + D[b][e] < D[a][b] + D[c][e]) implies D[a][c] < D[a][b] or D[b][e] < D[c][e], since an f32
// sum is monotone in both terms; with D[a][b] <= dk[a] the first puts c among a's kDmK nearest, with D[c][e] <= dk[e] the
// second puts b among e's (the matrix is symmetric: it was expanded from the packed triangle), i.e. e in b's reverse list.
// PER > 0: the row's n <= 64 * PER distances stay in registers over the 16 walks; PER = 0: they are read again each time.
template <int PER>
__global__ __launch_bounds__(256) void k_dm_lists(const float *__restrict__ full, uint32_t n, uint16_t *__restrict__ id, float *__restrict__ dl,
                                                  float *__restrict__ dk, uint16_t *__restrict__ inv_id, float *__restrict__ inv_d,
                                                  uint32_t *__restrict__ inv_cnt)
{
    const uint32_t a = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (a >= n) return;
    const float *__restrict__ row = full + (size_t)a * n;
    constexpr int NR = PER > 0 ? PER : 1;
    uint32_t kv[NR];  // order-preserving keys of this lane's distances (0xFFFFFFFF: no city — a itself, or beyond n)
    bool nan = false;
    if (PER > 0) {
#pragma unroll
        for (int t = 0; t < NR; ++t) {
            const uint32_t c = lane + 64u * (uint32_t)t;
            const float v = row[c < n ? c : 0u];
            nan = nan || (c < n && c != a && v != v);
            kv[t] = (c < n && c != a) ? fkey(v) : 0xFFFFFFFFu;  // (float order; -0 before +0, which compare equal: harmless)
        }
    }
    // each walk takes the smallest (key, city) beyond the last one taken: the smallest key first (a DPP maximum of the complement), then
    // the smallest city among the lanes that hold it
    uint32_t pk = 0u, pc = 0u;
    float last = 0.0f;
    uint32_t taken = 0;
    for (uint32_t r = 0; r < (uint32_t)kDmK; ++r) {
        uint32_t bk = 0xFFFFFFFFu, bc = 0xFFFFFFFFu;
        if (PER > 0) {
#pragma unroll
            for (int t = 0; t < NR; ++t) {
                const uint32_t c = lane + 64u * (uint32_t)t, k = kv[t];
                const bool beyond = r == 0u || k > pk || (k == pk && c > pc);
                const bool better = k < bk;  // (a lane's cities come in ascending order: the first of equal keys stays)
                if (k != 0xFFFFFFFFu && beyond && better) {
                    bk = k;
                    bc = c;
                }
            }
        } else {
            for (uint32_t c = lane; c < n; c += 64u) {
                if (c == a) continue;
                const float v = row[c];
                if (r == 0u && v != v) nan = true;
                const uint32_t k = fkey(v);
                const bool beyond = r == 0u || k > pk || (k == pk && c > pc);
                if (beyond && k < bk) {
                    bk = k;
                    bc = c;
                }
            }
        }
        const uint32_t mk = ~rl_u(wave_max_key_lane63(~bk), 63u);
        // (a lane without a candidate holds bc = 0xFFFFFFFF; a candidate's key can be 0xFFFFFFFF only for a NaN pattern, whose rows are never listed)
        const uint32_t mc = ~rl_u(wave_max_key_lane63((bk == mk && bc != 0xFFFFFFFFu) ? ~bc : 0u), 63u);
        if (mc == 0xFFFFFFFFu) break;  // fewer than kDmK other cities
        const float v = row[mc];
        if (lane == 0) {
            id[(size_t)a * kDmK + r] = (uint16_t)mc;
            dl[(size_t)a * kDmK + r] = v;
            const uint32_t slot = atomicAdd(&inv_cnt[mc], 1u);
            if (slot < (uint32_t)kDmInv) {
                inv_id[(size_t)mc * kDmInv + slot] = (uint16_t)a;
                inv_d[(size_t)mc * kDmInv + slot] = v;
            }
        }
        pk = mk;
        pc = mc;
        last = v;
        ++taken;
    }
    const bool any_nan = __builtin_amdgcn_ballot_w64(nan) != 0ull;
    if (lane == 0) {
        for (uint32_t r = taken; r < (uint32_t)kDmK; ++r) {
            id[(size_t)a * kDmK + r] = 0xFFFFu;
            dl[(size_t)a * kDmK + r] = 0.0f;
        }
        // a row with a NaN is never listed (every comparison with its dk fails); a row that lists every other city needs no bound
        dk[a] = any_nan ? __builtin_nanf("") : taken < (uint32_t)kDmK ? __builtin_inff() : last;
    }
}

// STAGE (small tours: 32 matrix rows fit the LDS next to the tour, n <= ~1170 — pr1002 does): in a wide block every wave
// copies its two matrix rows a and b into LDS with coalesced loads and gathers D[a][c], D[b][e] from there; a fully
// divergent global gather costs the CU's address unit ~64 cycles per wave instruction, an LDS gather a few.
// LATE: sweeps in which few cities have a tour edge beyond their kDmK-th distance run on the lists (k_dm_lists): a row is one
// pass over its <= 64 listed candidates + the long cities, kDmLateRows rows per wave at a time, a step scans the rest of the sweep.
template <bool STAGE, bool LATE>
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
    uint32_t *hkey = hres + (kDmNT / 64);                           // [NW] late sweeps: the key a wave posted in this step
    uint32_t *lctl = hkey + (kDmNT / 64);                           // [4] late sweeps: long cities, row counter
    // late sweeps: every city's kDmK-th distance, city -> position, "is in the long list", the long list
    float *dkl = reinterpret_cast<float *>(lctl + 4);
    uint16_t *pos = reinterpret_cast<uint16_t *>(dkl + (LATE ? nq : 0u));
    uint16_t *longl = pos + (LATE ? nq : 0u);
    uint8_t *lflag = reinterpret_cast<uint8_t *>(longl + (LATE ? kDmLongCap : 0u));
    float *rowbuf = reinterpret_cast<float *>(lflag + (LATE ? nq : 0u));  // STAGE: per wave two matrix rows of nq floats
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
    TL_SYNC();

    const uint32_t nrows = n - 3;
    uint32_t i0 = 0, j0 = 2, step = 0, sweeps = 1, status = 0;
    bool improved = false;
    uint64_t moves = 0, reversed = 0, sweep_m0 = 0;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    constexpr uint32_t NWv = kDmNT / 64;
    // optional move log (tl_two_opt_trace): (i << 16) | j per applied move in the reference's order, 0xFFFFFFFF where a new sweep begins
    uint32_t *mlog = A.move_log ? A.move_log + (size_t)d * A.log_cap : nullptr;
    uint32_t log_n = 0;
    uint32_t since_rows = 0;  // rows scanned since the last move
    uint32_t gap_rows = 0;    // ... and its running average over the recent moves: the block shape follows the larger of the two

    // ---- late sweeps (LATE): state and the decision taken where a sweep begins
    bool late = false;
    uint32_t n_late_steps = 0, n_late_sweeps = 0, n_full_rows = 0, n_xl_rows = 0;
    // per descent, in HBM: for a city whose row has been walked as `a`, the cities nearer than its tour successor was then
    uint2 *__restrict__ xlist = LATE ? reinterpret_cast<uint2 *>(A.work) + (size_t)d * ((size_t)n * (kDmXl + 1u)) : nullptr;
    uint2 *__restrict__ xmeta = LATE ? xlist + (size_t)n * kDmXl : nullptr;  // {entries (0xFFFFFFFF: none), bits of the D[a][b] they were cut for}
    const uint16_t *__restrict__ nlid = A.dml.id;
    const float *__restrict__ nld = A.dml.d;
    const uint16_t *__restrict__ invid = A.dml.inv_id;
    const float *__restrict__ invd = A.dml.inv_d;
    const uint32_t *__restrict__ invcnt = A.dml.inv_cnt;
    // a city is long when one of its (at most two) tour edges inside the path exceeds its kDmK-th distance — a reversal swaps the
    // two edges of the cities inside it, so only a move's four end points ever change
    auto is_long = [&](uint32_t u, uint32_t k) -> bool {
        const float dku = dkl[u];
        const bool pe_ok = k < 1u || edge[k - 1u] <= dku;
        const bool se_ok = k + 1u >= n || edge[k] <= dku;
        return !(pe_ok && se_ok);
    };
    auto sweep_begin = [&](uint32_t prev_moves) {  // every thread, at (i0, j0) = (0, 2): positions, long list, and whether the sweep runs on the lists
        if (!LATE) return;
        for (uint32_t k = tid; k < n; k += kDmNT) {
            pos[perm[k]] = (uint16_t)k;
            lflag[k] = 0;
        }
        if (tid == 0) {
            lctl[0] = 0u;
            lctl[1] = 0u;
            lctl[2] = 0u;
            lctl[3] = 0u;
        }
        TL_SYNC();
        for (uint32_t k = tid; k < n; k += kDmNT) {
            const uint32_t u = perm[k];
            if (is_long(u, k)) {
                const uint32_t idx = atomicAdd(&lctl[0], 1u);
                if (idx < kDmLongCap) longl[idx] = (uint16_t)u;
                lflag[u] = 1;
            }
        }
        TL_SYNC();
        // on the lists a move is a whole step (two dependent look-ups and a barrier), in the other block shapes a hit-free sweep is
        // n / 16 steps: the lists take the sweeps that follow one with few moves (the first sweep: the long cities stand in for them)
        const uint32_t nl0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)lctl[0]);
        late = nl0 <= A.dml.long_max && (prev_moves == 0xFFFFFFFFu ? nl0 : prev_moves) <= A.dml.moves_max;
        if (late) ++n_late_sweeps;
    };
    if (LATE) {
        for (uint32_t k = tid; k < n; k += kDmNT) {
            dkl[k] = A.dml.dk[k];
            xmeta[k] = make_uint2(0xFFFFFFFFu, 0u);
        }
        if (tid < kDmNT / 64) hkey[tid] = kNoKey;
        TL_SYNC();
        if (n >= 4) sweep_begin(0xFFFFFFFFu);
    }
    if ((size_t)n * n * 4u <= kDmWarmBytes) {
        // The descent touches a matrix row for the first time almost every step of its first sweep (row a of every i, the row of
        // every move's new b), and a first touch is a miss of this XCD's L2 — k_dm_expand_full ran on all of them — i.e. a round
        // trip to the Infinity Cache / HBM in front of a step.  So stream the matrix through once (4 MB at n = 1002: ~30 us).
        // (A descent whose first sweep already runs on the lists — an NN tour — touches few matrix entries: it warms the lists only.)
        const float4 *__restrict__ m4 = reinterpret_cast<const float4 *>(dm);
        const size_t n4 = late ? 0u : (size_t)n * n / 4u;
        float acc = 0.0f;
        for (size_t k = tid; k < n4; k += kDmNT) {
            const float4 v = m4[k];
            acc += (v.x + v.y) + (v.z + v.w);
        }
        if (LATE) {  // ... and the lists behind it (one buffer: ids, distances, bounds, reverse lists, counts)
            const float4 *__restrict__ l4 = reinterpret_cast<const float4 *>(A.dml.id);
            const size_t nl4 = (size_t)(reinterpret_cast<const unsigned char *>(A.dml.inv_cnt + n) - reinterpret_cast<const unsigned char *>(A.dml.id)) / 16u;
            for (size_t k = tid; k < nl4; k += kDmNT) {
                const float4 v = l4[k];
                acc += (v.x + v.y) + (v.z + v.w);
            }
        }
        if (acc == -1.0f) keys[3] = 0u;  // (never taken for the distances, which are >= 0; either way it only keeps the loads)
    }
    TL_SYNC();

#ifdef TL_DM_PROFILE
    uint64_t qd[24] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // wave 0's cycles in dense steps: row set-up, first decision, chain, barrier wait, boundary + reversals, [5] steps; wide steps from [8]: staging, -, scan, ...
    uint64_t tq = __builtin_amdgcn_s_memtime();
#define TL_DSTAMP(k) do { const uint64_t t_ = __builtin_amdgcn_s_memtime(); qd[(late_step ? 16 : wide ? 8 : 0) + (k)] += t_ - tq; tq = t_; } while (0)
#else
#define TL_DSTAMP(k) do { } while (0)
#endif
    bool done = n < 4;
    auto sweep_end = [&]() {  // (two_opt.rs:26-28) the sweep is over: stop, or begin the next one
        if (!improved) {
            done = true;
            return;
        }
        if (sweeps >= A.max_sweeps) {
            status = 1;
            done = true;
            return;
        }
        improved = false;
        ++sweeps;
        i0 = 0;
        j0 = 2;
        if (mlog && tid == 0 && log_n < A.log_cap) mlog[log_n] = 0xFFFFFFFFu;
        log_n += 1u;
        sweep_begin((uint32_t)(moves - sweep_m0));
        sweep_m0 = moves;
    };
    // 32-bit byte offsets from a uniform base (global_load with an SGPR base and a VGPR offset, no 64-bit address arithmetic per lane):
    // the late sweeps' state limits n to ~10^4, so the matrix stays below 2^29 bytes
    auto ld_f = [](const float *base, uint32_t idx) -> float { return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(base) + (idx << 2)); };
    auto ld_h = [](const uint16_t *base, uint32_t idx) -> uint32_t { return *reinterpret_cast<const uint16_t *>(reinterpret_cast<const char *>(base) + (idx << 1)); };
    // ---- the sweeps that run on the lists: a loop of their own, so that the other block shapes' loop carries none of their state
    auto late_sweeps = [&]() {
        while (late && !done) {
            const uint32_t slot = step % 3u;
            if (tid == 0) {
                keys[(step + 1u) % 3u] = kNoKey;
                lctl[1u + (step + 1u) % 3u] = 0u;  // the next step's row counter (its last readers are two barriers behind)
            }
            ++step;
            // A step scans rows i0 .. iend under "no move yet".  While moves come every few rows only the next `span` rows are looked at
            // (four times the running gap between moves): waves beyond them would only contend for the address unit and hold the
            // barrier; a hit-free stretch quadruples the span step by step up to the rest of the sweep.
            const uint32_t span = (since_rows > gap_rows ? since_rows : gap_rows) * 4u + 8u;
            const uint32_t iend = nrows - i0 > span ? i0 + span : nrows;
            uint32_t *rowctr = &lctl[1u + slot];
            constexpr bool late_step = true, wide = false;
            (void)late_step;
            (void)wide;
            {
                // A step scans the REST of the sweep under "no move yet": rows are taken kDmLateRows at a time from a counter, in order, and
                // decided together — their list entries, position look-ups and matrix gathers in flight at once.
                // improving => D[a][c] < D[a][b] or D[b][e] < D[c][e] (k_dm_lists), so a row's candidates are the union of
                //   A  c among a's nearest with D[a][c] < D[a][b] (lanes 0..15; the distance comes with the list, D[b][e] is gathered);
                //      when D[a][b] exceeds a's bound the list may miss some: then the matrix row of a is read instead (coalesced, by city);
                //   B  e in b's reverse list with D[b][e] < D[c][e] (lanes 16..63; D[a][c] gathered); an overfull reverse list: the matrix
                //      row of b, read the same way;
                //   C  the long cities e (their own lists say nothing about them) with D[b][e] < D[c][e]: D[b][e] gathered, D[a][c] only
                //      behind it.
                // Lanes that fail their list-side test issue no gather: a divergent gather costs the CU's address unit per distinct line.
                ++n_late_steps;
                const uint32_t nlong = (uint32_t)__builtin_amdgcn_readfirstlane((int)lctl[0]);  // (<= kDmLongCap while late)
                if (lane == 0) hkey[wave] = kNoKey;
                const bool aside = lane < (uint32_t)kDmK;
                for (;;) {
                    uint32_t g = 0;
                    if (lane == 0) g = atomicAdd(rowctr, kDmLateRows);
                    g = i0 + (uint32_t)__builtin_amdgcn_readfirstlane((int)g);
                    if (g >= iend) break;
                    const uint32_t kb = (uint32_t)__builtin_amdgcn_readfirstlane((int)keys[slot]);
                    if (kb != kNoKey && (kb >> 16) < g) break;  // a hit in an earlier row
                    uint32_t ra[kDmLateRows], rb[kDmLateRows], rjmin[kDmLateRows];
                    float rdab[kDmLateRows];
    #pragma unroll
                    for (uint32_t q = 0; q < kDmLateRows; ++q) {
                        const uint32_t i = g + q < nrows ? g + q : nrows - 1u;  // (rows beyond the sweep: the last row again, dropped below)
                        ra[q] = perm[i];
                        rb[q] = perm[i + 1u];
                        rdab[q] = edge[i];
                        rjmin[q] = i == i0 ? j0 : i + 2u;
                    }
                    uint32_t lu[kDmLateRows], cntb[kDmLateRows];
                    float ld[kDmLateRows];
    #pragma unroll
                    for (uint32_t q = 0; q < kDmLateRows; ++q) {
                        const uint32_t at = aside ? ra[q] * (uint32_t)kDmK + lane : rb[q] * (uint32_t)kDmInv + (lane - (uint32_t)kDmK);
                        lu[q] = ld_h(aside ? nlid : invid, at);
                        ld[q] = ld_f(aside ? nld : invd, at);
                        cntb[q] = invcnt[rb[q]];
                    }
                    uint32_t bj[kDmLateRows];       // per lane: its best (smallest) improving column of row q, and that candidate's two new edges
                    float bdac[kDmLateRows], bdbe[kDmLateRows];
                    // C, first block of 64 long cities: everything it needs is in LDS, so its D[b][e] gathers go out while the list entries
                    // of A and B are still on their way (the further blocks, rare, run behind)
                    uint32_t c_j = 0, c_c = 0;
                    float c_dce = 0.0f, c_gbe[kDmLateRows];
                    bool c_v[kDmLateRows];
                    {
                        const uint32_t e = longl[lane < nlong ? lane : 0u];
                        const uint32_t j = (uint32_t)pos[e] - 1u;
                        const uint32_t jl = j <= n - 2u ? j : n - 2u;
                        c_dce = edge[jl];
                        const bool ev = lane < nlong && j <= n - 2u && !(c_dce <= dkl[e]);  // (its edge towards c within its bound: b's reverse list has it)
                        c_c = perm[jl];
                        c_j = j;
    #pragma unroll
                        for (uint32_t q = 0; q < kDmLateRows; ++q) {
                            c_v[q] = ev && j >= rjmin[q];
                            c_gbe[q] = ld_f(dm, c_v[q] ? rb[q] * n + e : 0u);  // (lanes without a candidate read one shared word: no branch, and the
                                                                                        //  outstanding loads stay countable for s_waitcnt)
                        }
                    }
                    {   // A and B from the lists
                        uint32_t oth[kDmLateRows], cj[kDmLateRows];
                        float dce[kDmLateRows], gd[kDmLateRows], c_gac[kDmLateRows];
                        bool val[kDmLateRows];
    #pragma unroll
                        for (uint32_t q = 0; q < kDmLateRows; ++q) {
                            bool v = lu[q] != 0xFFFFu;
                            const uint32_t pu = pos[v ? lu[q] : 0u];
                            const uint32_t j = aside ? pu : pu - 1u;  // a's neighbour is c = p[j]; the others are e = p[j+1]
                            v = v && j >= rjmin[q] && j <= n - 2u;
                            const uint32_t jj = v ? j : n - 2u;
                            oth[q] = aside ? perm[jj + 1u] : perm[jj];
                            dce[q] = edge[jj];
                            cj[q] = j;
                            val[q] = v && (ld[q] < (aside ? rdab[q] : dce[q]));  // D[a][c] < D[a][b] resp. D[b][e] < D[c][e]
                        }
    #pragma unroll
                        for (uint32_t q = 0; q < kDmLateRows; ++q) {  // C: D[a][c] behind D[b][e] < D[c][e] (issued with the gathers of A and B)
                            c_v[q] = c_v[q] && c_gbe[q] < c_dce;
                            c_gac[q] = ld_f(dm, c_v[q] ? ra[q] * n + c_c : 0u);
                        }
#pragma unroll
                        for (uint32_t q = 0; q < kDmLateRows; ++q) gd[q] = ld_f(dm, val[q] ? (aside ? rb[q] : ra[q]) * n + oth[q] : 0u);
#pragma unroll
                        for (uint32_t q = 0; q < kDmLateRows; ++q) {
                            const float dac = aside ? ld[q] : gd[q], dbe = aside ? gd[q] : ld[q];
                            const bool imp = val[q] && (dac + dbe < rdab[q] + dce[q]);  // two_opt.rs:35-49
                            bj[q] = imp ? cj[q] : 0xFFFFFFFFu;
                            bdac[q] = dac;
                            bdbe[q] = dbe;
                        }
    #pragma unroll
                        for (uint32_t q = 0; q < kDmLateRows; ++q) {
                            const bool imp = c_v[q] && (c_gac[q] + c_gbe[q] < rdab[q] + c_dce);
                            if (imp && c_j < bj[q]) {
                                bj[q] = c_j;
                                bdac[q] = c_gac[q];
                                bdbe[q] = c_gbe[q];
                            }
                        }
                    }
                    TL_DSTAMP(0);
                    for (uint32_t base = 64u; base < nlong; base += 64u) {  // C: the long cities beyond the first 64
                        const uint32_t idx = base + lane;
                        const uint32_t e = longl[idx < nlong ? idx : 0u];
                        const uint32_t j = (uint32_t)pos[e] - 1u;
                        const uint32_t jl = j <= n - 2u ? j : n - 2u;
                        const float dce = edge[jl];
                        const bool ev = idx < nlong && j <= n - 2u && !(dce <= dkl[e]);
                        const uint32_t c = perm[jl];
                        float gbe[kDmLateRows], gac[kDmLateRows];
                        bool v2[kDmLateRows];
    #pragma unroll
                        for (uint32_t q = 0; q < kDmLateRows; ++q) {
                            v2[q] = ev && j >= rjmin[q];
                            gbe[q] = ld_f(dm, v2[q] ? rb[q] * n + e : 0u);
                        }
    #pragma unroll
                        for (uint32_t q = 0; q < kDmLateRows; ++q) {
                            v2[q] = v2[q] && gbe[q] < dce;
                            gac[q] = ld_f(dm, v2[q] ? ra[q] * n + c : 0u);
                        }
    #pragma unroll
                        for (uint32_t q = 0; q < kDmLateRows; ++q) {
                            const bool imp = v2[q] && (gac[q] + gbe[q] < rdab[q] + dce);
                            if (imp && j < bj[q]) {
                                bj[q] = j;
                                bdac[q] = gac[q];
                                bdbe[q] = gbe[q];
                            }
                        }
                    }
                    TL_DSTAMP(1);
    #ifdef TL_DM_PROFILE
                    qd[16 + 7] += 1;
    #endif
                    bool posted = false;
    #pragma unroll
                    for (uint32_t q = 0; q < kDmLateRows; ++q) {
                        if (posted || g + q >= iend) continue;
                        const uint32_t i = g + q;
                        const bool a_listed = rdab[q] <= dkl[ra[q]], b_listed = cntb[q] <= (uint32_t)kDmInv;
                        if (!(a_listed && b_listed)) {
                            // The matrix row of a (every city as c) and / or of b (as e), read by city (coalesced); the
                            // few cities that pass the one-sided test gather the other distance.  Not worth it once an earlier row has a hit.
                            const uint32_t kb2 = (uint32_t)__builtin_amdgcn_readfirstlane((int)keys[slot]);
                            if (kb2 != kNoKey && (kb2 >> 16) < i) {
                                posted = true;
                                continue;
                            }
                            const float *__restrict__ rowa = dm + (size_t)ra[q] * n;
                            const float *__restrict__ rowb = dm + (size_t)rb[q] * n;
                            // Row a's cities nearer than b are the same every time the scan meets this row (the matrix does not change) as long
                            // as D[a][b] has not grown: the first walk leaves them in the descent's cache (<= kDmXl of them; more: no entry),
                            // later ones read that one record.  A tour edge beyond the 16th distance tends to stay for many steps — the rows
                            // ahead of the scan are decided again by every step until it reaches them — and for good once 2-opt cannot mend it.
                            uint2 xm = make_uint2(0xFFFFFFFFu, 0u);
                            if (!a_listed) xm = xmeta[ra[q]];
                            if (!a_listed && xm.x <= kDmXl && rdab[q] <= __builtin_bit_cast(float, xm.y)) {
                                ++n_xl_rows;
                                const bool in = lane < xm.x;
                                const uint2 ent = xlist[(size_t)ra[q] * kDmXl + (in ? lane : 0u)];
                                const float dac = __builtin_bit_cast(float, ent.y);
                                const uint32_t j = pos[ent.x];
                                const bool f = in && j >= rjmin[q] && j <= n - 2u && dac < rdab[q];
                                const float dbe = rowb[f ? perm[j + 1u] : 0u];
                                if (f && dac + dbe < rdab[q] + edge[j] && j < bj[q]) {
                                    bj[q] = j;
                                    bdac[q] = dac;
                                    bdbe[q] = dbe;
                                }
                            } else if (!a_listed) {  // row a, every city as c: eight coalesced loads in flight, then the survivors' D[b][e] together
                                ++n_full_rows;
                                uint32_t xbase = 0u;
                                for (uint32_t u0 = 0; u0 < n; u0 += 512u) {
                                    float va[8], wa[8];
                                    uint32_t ja[8];
    #pragma unroll
                                    for (int t = 0; t < 8; ++t) {
                                        const uint32_t u = u0 + 64u * (uint32_t)t + lane;
                                        va[t] = rowa[u < n ? u : 0u];
                                    }
                                    uint32_t fa = 0u;
    #pragma unroll
                                    for (int t = 0; t < 8; ++t) {
                                        const uint32_t u = u0 + 64u * (uint32_t)t + lane;
                                        ja[t] = pos[u < n ? u : 0u];
                                        const bool near = u < n && u != ra[q] && va[t] < rdab[q];  // (what the cache keeps: at any position)
                                        const bool f = near && ja[t] >= rjmin[q] && ja[t] <= n - 2u;
                                        fa |= (f ? 1u : 0u) << t;
                                        wa[t] = rowb[f ? perm[ja[t] + 1u] : 0u];  // D[b][e]
                                        const uint64_t nm = __builtin_amdgcn_ballot_w64(near);
                                        const uint32_t at = xbase + __builtin_amdgcn_mbcnt_hi((uint32_t)(nm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)nm, 0u));
                                        if (near && at < kDmXl) xlist[(size_t)ra[q] * kDmXl + at] = make_uint2(u, __builtin_bit_cast(uint32_t, va[t]));
                                        xbase += (uint32_t)__builtin_popcountll(nm);
                                    }
    #pragma unroll
                                    for (int t = 0; t < 8; ++t) {
                                        if (((fa >> t) & 1u) && va[t] + wa[t] < rdab[q] + edge[ja[t]] && ja[t] < bj[q]) {
                                            bj[q] = ja[t];
                                            bdac[q] = va[t];
                                            bdbe[q] = wa[t];
                                        }
                                    }
                                }
                                if (lane == 0) xmeta[ra[q]] = make_uint2(xbase <= kDmXl ? xbase : 0xFFFFFFFFu, __builtin_bit_cast(uint32_t, rdab[q]));
                            }
                            if (!b_listed) {
                                ++n_full_rows;  // row b, every city as e (clustered instances: more than 48 cities hold b among their 16 nearest)
                                for (uint32_t u0 = 0; u0 < n; u0 += 256u) {
                                    float vb[4], wb[4];
                                    uint32_t jb2[4];
                                    bool fb[4];
    #pragma unroll
                                    for (int t = 0; t < 4; ++t) {
                                        const uint32_t u = u0 + 64u * (uint32_t)t + lane;
                                        vb[t] = rowb[u < n ? u : 0u];
                                    }
    #pragma unroll
                                    for (int t = 0; t < 4; ++t) {
                                        const uint32_t u = u0 + 64u * (uint32_t)t + lane;
                                        jb2[t] = (uint32_t)pos[u < n ? u : 0u] - 1u;
                                        fb[t] = u < n && jb2[t] >= rjmin[q] && jb2[t] <= n - 2u && vb[t] < edge[jb2[t] <= n - 2u ? jb2[t] : 0u];
                                        wb[t] = rowa[fb[t] ? perm[jb2[t]] : 0u];  // D[a][c]
                                    }
    #pragma unroll
                                    for (int t = 0; t < 4; ++t) {
                                        if (fb[t] && wb[t] + vb[t] < rdab[q] + edge[jb2[t]] && jb2[t] < bj[q]) {
                                            bj[q] = jb2[t];
                                            bdac[q] = wb[t];
                                            bdbe[q] = vb[t];
                                        }
                                    }
                                }
                            }
                        }
                        if (!__builtin_amdgcn_ballot_w64(bj[q] != 0xFFFFFFFFu)) continue;  // nothing in this row
                        const uint32_t col = ~rl_u(wave_max_key_lane63(~bj[q]), 63u);
                        const uint64_t wm = __builtin_amdgcn_ballot_w64(bj[q] == col);
                        if (lane == (uint32_t)(__builtin_ffsll((long long)wm) - 1)) {
                            lds_min_u32(&keys[slot], (i << 16) | col);
                            hkey[wave] = (i << 16) | col;
                            hl_j[wave * kDmChain] = col;
                            hl_a[wave * kDmChain] = bdac[q];
                            hl_b[wave * kDmChain] = bdbe[q];
                            hcnt[wave] = 1u;
                            hres[wave] = col + 1u;
                        }
                        posted = true;  // this wave's later rows are later rows
                    }
                    TL_DSTAMP(6);
                    if (posted) break;
                }
            }
            TL_DSTAMP(2);
            TL_SYNC();
            TL_DSTAMP(3);
            const uint32_t key = (uint32_t)__builtin_amdgcn_readfirstlane((int)keys[slot]);
            if (key == kNoKey) {
                since_rows += iend - i0;
                i0 = iend;
                j0 = i0 + 2u;
            } else {
                const uint32_t is = key >> 16, hi = key & 0xFFFFu, lo = is + 1u;
                gap_rows = (gap_rows + since_rows + (is - i0) + 1u) >> 1;  // running estimate of the rows between moves
                since_rows = 0;
                // the wave that posted the winning key (keys are distinct; a wave that posted none holds kNoKey)
                const uint32_t ww = (uint32_t)(__builtin_ffsll((long long)__builtin_amdgcn_ballot_w64(lane < NWv && hkey[lane < NWv ? lane : 0u] == key)) - 1);
                uint32_t ni0 = is, nj0 = hi + 1u;  // where the scan resumes (two_opt.rs:30-33: the same row, the next column)
                if (nj0 > n - 2u) {
                    ++ni0;
                    nj0 = ni0 + 2u;
                }
                const float nac = hl_a[ww * kDmChain], nbe = hl_b[ww * kDmChain];
                if (tid < 4) {
                    // The four cities whose tour edges change — a, the old b (now at hi), the old c (now at lo) and e, one lane each —
                    // enter the long list if a new edge exceeds their bound; the next step's row counter.
                    // (read here, before the swaps: this wave's own lane 0 swaps perm[lo] / perm[hi] and edge[lo] / edge[hi-1] below)
                    const uint32_t cp = tid == 0 ? is : tid == 1 ? lo : tid == 2 ? hi : hi + 1u;
                    const uint32_t city = perm[cp];
                    // a: (edge[is-1], new ac)   old b: (edge[lo], new be)   old c: (new ac, edge[hi-1])   e: (new be, edge[hi+1])
                    const uint32_t ep = tid == 0 ? is - 1u : tid == 1 ? lo : tid == 2 ? hi - 1u : hi + 1u;
                    const bool has = tid == 0 ? is >= 1u : tid == 3 ? hi + 2u < n : true;
                    const float e1 = (tid == 0 || tid == 2) ? nac : nbe, e2 = has ? edge[has ? ep : 0u] : e1;
                    const float dku = dkl[city];
                    if (!(e1 <= dku && e2 <= dku) && !lflag[city]) {
                        const uint32_t idx = atomicAdd(&lctl[0], 1u);
                        if (idx < kDmLongCap) longl[idx] = (uint16_t)city;
                        lflag[city] = 1;
                    }
                }
                const uint32_t half = (hi - lo + 1u) >> 1;  // swap_2opt(path, i+1, j), two_opt.rs:69-79
                for (uint32_t t = tid; t < half; t += kDmNT) {
                    const uint32_t u = perm[lo + t], v = perm[hi - t];
                    perm[lo + t] = v;
                    perm[hi - t] = u;
                    pos[v] = (uint16_t)(lo + t);
                    pos[u] = (uint16_t)(hi - t);
                }
                const uint32_t ehalf = (hi - lo) >> 1;
                for (uint32_t t = tid; t < ehalf; t += kDmNT) {
                    const float x = edge[lo + t], y = edge[hi - 1u - t];
                    edge[lo + t] = y;
                    edge[hi - 1u - t] = x;
                }
                if (tid == 0) {
                    edge[lo - 1u] = nac;
                    edge[hi] = nbe;
                }
                reversed += (uint64_t)(hi - is);
                if (mlog && tid == 0 && log_n < A.log_cap) mlog[log_n] = key;
                TL_SYNC();
                log_n += 1u;
                improved = true;
                moves += 1u;
                i0 = ni0;
                j0 = nj0;
                // more long cities than the list holds: the rest of the sweep in the other block shapes
                if ((uint32_t)__builtin_amdgcn_readfirstlane((int)lctl[0]) > kDmLongCap) late = false;
            }
            TL_DSTAMP(4);
#ifdef TL_DM_PROFILE
            qd[16 + 5] += 1;
#endif
            if (i0 >= nrows) sweep_end();
        }
        since_rows = 0;
    };
    while (!done) {
        if (LATE && late) {
            late_sweeps();
            continue;
        }
        while (!done && !(LATE && late)) {
            const uint32_t slot = step % 3u;
            if (tid == 0) keys[(step + 1u) % 3u] = kNoKey;
            ++step;
            // Two block shapes, chosen from the observed gap between moves (like the coordinate kernel):
            //  wide  — moves are rare: 16 rows per step, one per wave, lanes along j;
            //  dense — moves every few rows: one row per step, its columns dealt to the 16 waves.
            // Either way the lexicographically first improving (i, j) wins (ds_min_u32 on i << 16 | j) and the scan
            // resumes at (i, j+1) like the reference.
            constexpr bool late_step = false;
            (void)late_step;
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
                uint32_t ni0 = is, nj0 = resume;  // where the scan resumes (two_opt.rs:30-33: the same row, the next column)
                if (nj0 > n - 2u) {
                    ++ni0;
                    nj0 = ni0 + 2u;
                }
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
                i0 = ni0;
                j0 = nj0;
            }
            TL_DSTAMP(4);
#ifdef TL_DM_PROFILE
            qd[(late_step ? 16 : wide ? 8 : 0) + 5] += 1;
#endif
            if (i0 >= nrows) sweep_end();
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
        st[5] = n_late_steps;   // steps, sweeps begun on the lists, rows of those that walked their matrix rows (wave 0's)
        st[6] = n_late_sweeps;
        st[7] = n_full_rows;
        st[8] = n_xl_rows;      // ... and rows that read a's cached record instead
        st[15] = log_n;  // words offered to the move log (moves + sweep marks)
#ifdef TL_DM_PROFILE
        printf("dmprof late: lists A/B %lu long C %lu rows+post %lu rest %lu wait %lu apply %lu steps %lu grabs %lu matrix-row rows %u\n", qd[16], qd[17], qd[22], qd[18], qd[19], qd[20], qd[21], qd[23], n_full_rows);
        printf("dmprof dense: setup %lu first %lu chain %lu wait %lu boundary %lu steps %lu | wide: staging %lu scan %lu wait %lu boundary %lu steps %lu\n", qd[0], qd[1], qd[2], qd[3], qd[4], qd[5], qd[8], qd[10], qd[11], qd[12], qd[13]);
#endif
    }
}

size_t two_opt_ref_dm_lds_bytes(uint32_t n)
{
    return (size_t)((n + 1u + 3u) & ~3u) * 8 + 16 + (size_t)(kDmNT / 64) * (kDmChain * 12 + 12) + 16;
}
static size_t two_opt_ref_dm_stage_bytes(uint32_t n) { return (size_t)((n + 1u + 3u) & ~3u) * 4 * 2 * (kDmNT / 64); }
// ... the late sweeps' state beside it: every city's bound (4 B), position (2 B) and long flag (1 B), the long list
static size_t two_opt_ref_dm_late_bytes(uint32_t n) { return (size_t)((n + 1u + 3u) & ~3u) * 7 + (size_t)kDmLongCap * 2; }

hipError_t launch_dm_expand_full(const float *packed, uint32_t n, float *full, hipStream_t s)
{
    const uint32_t nt = (n + 63u) / 64u;
    hipLaunchKernelGGL(k_dm_expand_full, dim3(nt, nt), dim3(256), 0, s, packed, n, full);
    return hipGetLastError();
}

// workspace of the lists: id, d, dk, inv_id, inv_d, inv_cnt (each 256-byte aligned)
static size_t al256(size_t v) { return (v + 255u) & ~(size_t)255u; }
size_t dm_lists_ws_bytes(uint32_t n)
{
    return al256((size_t)n * kDmK * 2) + al256((size_t)n * kDmK * 4) + al256((size_t)n * 4) + al256((size_t)n * kDmInv * 2) +
           al256((size_t)n * kDmInv * 4) + al256((size_t)n * 4);
}
size_t two_opt_ref_dm_late_work_bytes(uint32_t n, uint32_t count) { return (size_t)count * n * (kDmXl + 1u) * 8u; }
bool two_opt_ref_dm_late_fits(uint32_t n, int lds_budget)
{
    // (n <= 16384: the late sweeps address the matrix and the lists with 32-bit byte offsets; the LDS budget is the tighter bound on gfx950)
    return n >= 8u && n <= 16384u && two_opt_ref_dm_lds_bytes(n) + two_opt_ref_dm_late_bytes(n) <= (size_t)lds_budget;
}
hipError_t launch_dm_lists_build(const float *full, uint32_t n, void *ws, DmLists *out, hipStream_t s)
{
    unsigned char *p = static_cast<unsigned char *>(ws);
    uint16_t *id = reinterpret_cast<uint16_t *>(p);
    p += al256((size_t)n * kDmK * 2);
    float *d = reinterpret_cast<float *>(p);
    p += al256((size_t)n * kDmK * 4);
    float *dk = reinterpret_cast<float *>(p);
    p += al256((size_t)n * 4);
    uint16_t *inv_id = reinterpret_cast<uint16_t *>(p);
    p += al256((size_t)n * kDmInv * 2);
    float *inv_d = reinterpret_cast<float *>(p);
    p += al256((size_t)n * kDmInv * 4);
    uint32_t *inv_cnt = reinterpret_cast<uint32_t *>(p);
    hipError_t e = hipMemsetAsync(inv_id, 0xFF, (size_t)n * kDmInv * 2, s);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(inv_cnt, 0, (size_t)n * 4, s);
    if (e != hipSuccess) return e;
    auto kern = n <= 1024u ? k_dm_lists<16> : n <= 2048u ? k_dm_lists<32> : k_dm_lists<0>;
    hipLaunchKernelGGL(kern, dim3((n + 3u) / 4u), dim3(256), 0, s, full, n, id, d, dk, inv_id, inv_d, inv_cnt);
    out->id = id;
    out->d = d;
    out->dk = dk;
    out->inv_id = inv_id;
    out->inv_d = inv_d;
    out->inv_cnt = inv_cnt;
    return hipGetLastError();
}

hipError_t launch_two_opt_ref_dm(const TwoOptBatchArgs &A, uint32_t count, int lds_budget, hipStream_t s)
{
    const bool late = A.dml.id != nullptr && two_opt_ref_dm_late_fits(A.n, lds_budget);
    const size_t base = two_opt_ref_dm_lds_bytes(A.n) + (late ? two_opt_ref_dm_late_bytes(A.n) : 0u), staged = base + two_opt_ref_dm_stage_bytes(A.n);
    const bool stage = staged <= (size_t)lds_budget;
    const size_t lds = stage ? staged : base;
    auto kern = late ? (stage ? k_two_opt_ref_dm<true, true> : k_two_opt_ref_dm<false, true>)
                     : (stage ? k_two_opt_ref_dm<true, false> : k_two_opt_ref_dm<false, false>);
    hipError_t e = allow_max_lds(reinterpret_cast<const void *>(kern));
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(count), dim3(kDmNT), lds, s, A);
    return hipGetLastError();
}

}  // namespace tl
