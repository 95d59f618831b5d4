// two_opt_ref.hip — REF_ORDER 2-opt: the reference's first-improvement descent
// (src/tsp/two_opt.rs:26-61) as ONE persistent workgroup per descent, whole tour resident in LDS.
//
// Reference loop, per sweep:  for i in 0..n-3 { for j in i+2..n-1 {
//     if d(p[i],p[j]) + d(p[i+1],p[j+1]) < d(p[i],p[i+1]) + d(p[j],p[j+1]) { reverse p[i+1..=j] } } }
// repeated until a sweep applies no move.  Move k+1 sees the path after move k, so the scan is
// sequential *between* moves and parallel only *within* the stretch of candidates up to the next
// improving one.  MI355X mapping:
//   - one descent = one workgroup = one CU; tour-ordered coordinates P[k] = xy[perm[k]] (float2) and
//     perm (u16) live in that CU's LDS for the whole descent (10 B/city: n <= ~15 K) — zero HBM
//     traffic inside the loop; 256 CUs = 256 concurrent restarts (north-star config 4);
//   - a step speculatively decides a block of R rows (i0..i0+R) x all j under "no move yet" and
//     reduces the lexicographically first improving (i,j) with one ds_min_u32 on a packed key; the
//     workgroup then applies that reversal cooperatively in LDS (dense rows: deferred, composed) and
//     resumes at (i, j+1), exactly where the reference's inner loop continues;
//   - wave 0 ("control") keeps the descent's accounting and does not scan; the other waves
//     ("workers") carry only a cursor and advance it themselves at a step's end from words all of
//     them read behind the same barrier (see "Role split" in the kernel body).
//
// Every candidate is decided exactly as the reference decides it, by a cascade of exact tests
// (DESIGN.md "Exact decision cascade"; f32 add, mul, sqrt are monotone, so each bound holds in
// the reference's own arithmetic, not just over the reals):
//   L0  tile bound   lanes = 64-position tiles.  With box_t the bounding box of P[64t..64t+64] and
//                    msq_t the largest squared tour-edge in the tile: if lb(a, box_t) >= sq_ab and
//                    lb(b, box_t) >= msq_t then no j in tile t can improve row (a,b).
//   L1  pair bound   lanes = j.  improving  =>  sq_ac < sq_ab  or  sq_be < sq_ce.
//   L2  approximate  v_sqrt_f32 (<= 1 ulp) decides unless |new - cur| is within ~3x the error bound.
//   L3  exact        four correctly rounded sqrt, two adds, strict compare (two_opt.rs:35-49).
// TL_FLAG_NO_PRUNE runs L3 on every candidate instead.
#include "tl_kernels.h"
#include "two_opt_common.h"

#include <type_traits>

#pragma clang fp contract(off)

namespace tl {

namespace {

constexpr uint32_t kNoKey = 0xFFFFFFFFu;
#ifndef TL_RMAX
#define TL_RMAX 63
#endif
#ifndef TL_DENSE_ROWS
#define TL_DENSE_ROWS 8.0f
#endif
constexpr int kRMax = TL_RMAX;        // rows per speculative block in pruned mode (<= 63: lane-resident row table)
constexpr uint32_t kQCap = 32;        // u32 words per hit list (4 used: count, resume column, the row's new b).  32 lists = 16 waves x
                                      // (step parity): a list is read after its step's barrier while its owner may already write
                                      // the next step's, so the parities alternate.  The 4 KB are also the cost-sum scratch.
constexpr uint32_t kPendMax = 64;     // deferred reversals of one row (ctl->pend): lane m of a flush reads hit m (= the flush's segment table)
constexpr int kFlushSlots = 15;       // elements per thread a flush can hold: 15 x 1024 covers every n that fits the LDS
constexpr int kFlushSlotsFx = 20;     // grid-coordinate form on 8 waves: 20 x 512 covers n = 10^4 (two tours per CU)
constexpr int kMaxGroups = 4;         // 64-tile groups: n_pad <= 4 * 64 * 64 = 16384


struct Ctl {                     // kCtlBytes of LDS
    uint2 kr[4];                // per key slot (step % 3): .x the step's first-hit key (ds_min_u32), .y the control wave's request for its
                                // descriptor at that step's end (block-shape change) — one 8-byte read gives a wave both
    uint32_t desc[4];           // the descriptor the control wave publishes before barrier B0 (sweep start / exit / block shape)
    unsigned long long cnt[6];  // cascade work of the descent (TileCounts summed over the waves; counting instantiation)
    unsigned long long clk0, rt0;  // s_memtime / s_memrealtime at the start of the descent (kept here, not in SGPRs)
    uint32_t bad_init;
    uint32_t nlong;             // late phase: entries of the long list (counted on beyond kNlLongCap: the list is then incomplete and unused)
    uint32_t late_cur;          // late phase: the next row of the step's scan nobody has taken yet
    uint32_t pad_[1];
    uint32_t pend[64];          // deferred hit columns of the current dense row (lane m of a flush reads hit m)
};
constexpr size_t kCtlBytes = 384;
static_assert(sizeof(Ctl) <= kCtlBytes, "Ctl block");

// Deferred reversals (dense mode).  The hits (i, g_0 < g_1 < ... < g_{k-1}) of ONE row all reverse a prefix that starts at
// lo = i+1 (two_opt.rs:50 swap_2opt(path, i+1, j)), and the rest of that row's scan reads only positions > g and
// b = p[i+1] = the old p[g_last] — so nothing inside [lo..g_last] is looked at before the row ends and the k reversals
// can be applied at once.  With S_0 = [lo..g_0], S_m = [g_{m-1}+1..g_m] their composition is
//     rev(S_{k-1}) rev(S_{k-3}) ... | ... S_{k-4} S_{k-2}
// (segments of the parity of k-1 reversed, in descending order, then the others as they are, ascending): every element
// moves once instead of once per hit.  Lane m of every wave holds segment m's table entry; a thread takes source
// positions lo + q*NT + tid, whose segment index is the number of hits below the position (ballot + popcount for the
// wave's 64-wide window, then a wave-uniform walk over the few segments the window overlaps), keeps value and target
// in registers across one barrier and stores.  Returns (per lane m < k) g_m - i, the reference's reversal length.
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_shr0(uint32_t v)
{
    // row_shr within rows of 16 lanes; lanes without a source read 0
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);
}
__device__ __forceinline__ uint32_t readlane_u(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); }

// `g`: lane m < k holds hit column g_m (the row's deferred-hit register of every wave), lanes >= k hold 0xFFFFFFFF.
template <int NT, int SLOTS, typename PT>
__device__ __forceinline__ uint32_t flush_deferred(const PT &P, uint16_t *perm, uint32_t g, uint32_t lo, uint32_t k, int lane, int wave)
{
    using Raw = decltype(pt_raw(P, 0u));
    const bool have = (uint32_t)lane < k;
    const uint32_t ghi = readlane_u(g, k - 1u);
    const uint32_t wfirst = lo + ((uint32_t)wave << 6);  // this wave's first window (wave-uniform)
    Raw val[SLOTS];
    uint32_t pk[SLOTS];  // perm id | target << 16
    if (wfirst <= ghi) {  // waves without a window go straight to the barriers
        // g of the lane below (row_shr stays inside its row of 16: lanes 16, 32, 48 are patched)
        uint32_t gprev = dpp_shr0<0x111>(g);
        gprev = lane == 16 ? readlane_u(g, 15) : gprev;
        gprev = lane == 32 ? readlane_u(g, 31) : gprev;
        gprev = lane == 48 ? readlane_u(g, 47) : gprev;
        gprev = lane == 0 ? lo - 1u : gprev;
        const uint32_t start = gprev + 1u;
        const uint32_t len = have ? g - gprev : 0u;
        // inclusive prefix sum over the lanes of the same parity: stride-2 scan inside each row of 16, then the rows below
        uint32_t pre = len;
        pre += dpp_shr0<0x112>(pre);
        pre += dpp_shr0<0x114>(pre);
        pre += dpp_shr0<0x118>(pre);
        if (k > 16u) {
            const uint32_t e0 = readlane_u(pre, 14), o0 = readlane_u(pre, 15);
            const uint32_t e1 = e0 + readlane_u(pre, 30), o1 = o0 + readlane_u(pre, 31);
            const uint32_t e2 = e1 + readlane_u(pre, 46), o2 = o1 + readlane_u(pre, 47);
            const bool odd = (lane & 1) != 0;
            uint32_t off = 0u;
            off = lane >= 16 ? (odd ? o0 : e0) : off;
            off = lane >= 32 ? (odd ? o1 : e1) : off;
            off = lane >= 48 ? (odd ? o2 : e2) : off;
            pre += off;
        }
        const uint32_t t_rev = readlane_u(pre, k - 1u);  // total length of the reversed group
        const bool isrev = ((k - 1u - (uint32_t)lane) & 1u) == 0u;
        const uint32_t base = lo + (isrev ? t_rev - pre : t_rev + pre - len);
        const uint32_t cst = isrev ? base + g : base - start;  // target = cst - p (reversed) or cst + p (kept)
#pragma unroll
        for (int q = 0; q < SLOTS; ++q) {
            const uint32_t w0 = wfirst + (uint32_t)(q * NT);
            if (w0 > ghi) break;
            const uint32_t p = w0 + (uint32_t)lane;
            const uint32_t w1 = (w0 + 63u < ghi) ? w0 + 63u : ghi;
            uint32_t m = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(g < w0));
            const uint32_t mhi = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(g < w1));
            const uint32_t c0 = readlane_u(cst, m);
            uint32_t dst = ((k - 1u - m) & 1u) == 0u ? c0 - p : c0 + p;  // the window's first segment (usually its only one)
            for (++m; m <= mhi; ++m) {
                const uint32_t sm = readlane_u(start, m), cm = readlane_u(cst, m);
                const bool rv = ((k - 1u - m) & 1u) == 0u;
                dst = p >= sm ? (rv ? cm - p : cm + p) : dst;
            }
            if (p <= ghi) {
                val[q] = pt_raw(P, p);
                pk[q] = (uint32_t)perm[p] | (dst << 16);
            }
        }
    }
    TL_SYNC();
    if (wfirst <= ghi) {
#pragma unroll
        for (int q = 0; q < SLOTS; ++q) {
            const uint32_t w0 = wfirst + (uint32_t)(q * NT);
            if (w0 > ghi) break;
            if (w0 + (uint32_t)lane <= ghi) {
                const uint32_t dst = pk[q] >> 16;
                pt_put(P, dst, val[q]);
                perm[dst] = (uint16_t)pk[q];
            }
        }
    }
    TL_SYNC();
    return have ? g - (lo - 1u) : 0u;
}

// two_opt.rs:50,69-79  swap_2opt(path, lo, hi): in-place reversal by the whole workgroup, two pairs per thread in flight.
// The caller puts a barrier behind it.
// `pos` (late phase): the city -> position table follows the tour where it is valid (nullptr: not kept).
template <int NT, typename PT>
__device__ __forceinline__ void reverse_segment(const PT &P, uint16_t *perm, uint32_t lo, uint32_t hi, int tid, uint16_t *pos = nullptr)
{
    const uint32_t half = (hi - lo + 1u) >> 1;
    for (uint32_t t = (uint32_t)tid; t < half; t += 2 * NT) {
        const uint32_t t2 = t + NT;
        const bool two = t2 < half;
        const auto x = pt_raw(P, lo + t), y = pt_raw(P, hi - t);
        const uint16_t u = perm[lo + t], v = perm[hi - t];
        uint16_t u2 = u, v2 = v;
        auto x2 = x, y2 = y;
        if (two) {
            x2 = pt_raw(P, lo + t2);
            y2 = pt_raw(P, hi - t2);
            u2 = perm[lo + t2];
            v2 = perm[hi - t2];
        }
        pt_put(P, lo + t, y);
        pt_put(P, hi - t, x);
        perm[lo + t] = v;
        perm[hi - t] = u;
        if (pos) {
            pos[v] = (uint16_t)(lo + t);
            pos[u] = (uint16_t)(hi - t);
        }
        if (two) {
            pt_put(P, lo + t2, y2);
            pt_put(P, hi - t2, x2);
            perm[lo + t2] = v2;
            perm[hi - t2] = u2;
            if (pos) {
                pos[v2] = (uint16_t)(lo + t2);
                pos[u2] = (uint16_t)(hi - t2);
            }
        }
    }
}

// ---- role-split kernel: the scan cursor every wave carries, the accounting only the control wave keeps, and the step boundary
struct Cursor {
    uint32_t i0, j0;              // where the reference's loop stands: row, resume column
    uint32_t np;                  // dense mode: hits of row i0 whose reversals are deferred (columns in ctl->pend)
    uint32_t slot, par;           // step % 3 (key slot), step & 1 (hit lists, request word)
    uint32_t dirty_lo, dirty_hi;  // tiles whose L0 metadata is stale
    bool pruned;                  // block shape
};
struct Acct {
    uint64_t moves, reversed, rev_lane;  // rev_lane: per-lane share of `reversed` from the flushes (summed over lanes at the end)
    float gap_est, since;                // candidates between moves (block-shape heuristic only)
    bool improved;
    uint32_t *log;                       // optional move log of this descent (row << 16 | column per applied move, in order; 0xFFFFFFFF
    uint32_t log_cap, log_n;             // where a new sweep begins), its capacity and the number of words so far (counted on beyond it)
};

// The deferred reversals of row c.i0, composed (flush_deferred), by every wave.  `sync_first`: the hits were filed in this very
// boundary (by their owner, behind B2), so a barrier comes before they are read.
template <bool CONTROL, int NT, int SLOTS, typename PT>
__device__ __forceinline__ void flush_pending(Cursor &c, Acct &a, const PT &P, uint16_t *perm, Ctl *ctl, int lane, int wave, bool sync_first)
{
    if (sync_first) TL_SYNC();
    uint32_t g = ctl->pend[lane];
    g = (uint32_t)lane < c.np ? g : 0xFFFFFFFFu;
    const uint32_t ghi = readlane_u(g, c.np - 1u);
    const uint32_t r = flush_deferred<NT, SLOTS>(P, perm, g, c.i0 + 1u, c.np, lane, wave);
    if (CONTROL) {
        a.rev_lane += r;
        if (a.log) {  // the row's hits, in column order = the order the reference applies them in
            const uint32_t at = a.log_n + (uint32_t)lane;
            if ((uint32_t)lane < c.np && at < a.log_cap) a.log[at] = (c.i0 << 16) | g;
            a.log_n += c.np;
        }
    }
    const uint32_t t0 = c.i0 >> 6, t1 = ghi >> 6;  // L0 metadata of tiles with a changed position or tour-edge
    c.dirty_lo = t0 < c.dirty_lo ? t0 : c.dirty_lo;
    c.dirty_hi = t1 > c.dirty_hi ? t1 : c.dirty_hi;
    c.np = 0;
}

// End of a step, behind barrier B2: every wave reads the same words and advances its cursor the same way (the control wave adds
// its accounting).  R: rows of the block just scanned.  my_hits: the hit columns this wave chained in the step (lane h = h-th
// hit), filed in ctl->pend if they are the step's.  Returns true where the control wave's descriptor comes next (sweep ended, or
// it asked for another block shape).
template <bool CONTROL, int NT, int SLOTS, bool FOLD, typename PT>
__device__ __forceinline__ bool step_boundary(Cursor &c, Acct &a, const PT &P, uint16_t *perm, Ctl *ctl, const uint32_t *queues, uint32_t n, uint32_t nrows,
                                              uint32_t R, int lane, int wave, int tid, uint32_t my_hits, float &bx, float &by, bool &reload, float *tmsq)
{
    const uint2 kr = ctl->kr[c.slot];
    const uint32_t key = (uint32_t)__builtin_amdgcn_readfirstlane((int)kr.x);
    const uint32_t req = (uint32_t)__builtin_amdgcn_readfirstlane((int)kr.y);
    const float rowlen = (float)(n - 2u - c.i0);
    reload = false;
    if (!c.pruned) {
        // what a dense step's end asks for: flush the row's deferred reversals (behind a barrier if hits were filed just now),
        // go to the next row
        bool flush = false, sync_first = false, next_row = false;
        if (key == kNoKey) {  // row i0 is finished: its reversals are due, composed
            if (CONTROL) a.since += rowlen;
            flush = c.np != 0u;
            next_row = true;
        } else {
            // the wave owning the first hit chained every improving move inside its tile (dense_tile).  Nothing of the row's
            // remaining scan reads [i+1..hit], so the reversals wait in ctl->pend until the row ends (flush_deferred); the scan
            // goes on at `resume` with b = the old p[hit].
            const uint4 hv = *reinterpret_cast<const uint4 *>(queues + ((key & 0xFFFFu) * 2u + c.par) * kQCap);
            const uint32_t nh = (uint32_t)__builtin_amdgcn_readfirstlane((int)hv.x);
            const uint32_t resume = (uint32_t)__builtin_amdgcn_readfirstlane((int)hv.y);
            if (!CONTROL && (key & 0xFFFFu) == (uint32_t)wave && (uint32_t)lane < nh) ctl->pend[c.np + (uint32_t)lane] = my_hits;
            if (CONTROL) {
                a.moves += nh;
                // gap estimate: the first hit exactly, chained hits as evenly spaced
                a.since += (float)((key >> 16) - c.j0);
                a.gap_est = 0.5f * (a.gap_est + a.since);
                a.since = 0.0f;
                if (nh > 1u) a.gap_est = fminf(a.gap_est, 0.5f * a.gap_est + 16.0f);  // chained hits are < 64 columns apart
                a.improved = true;
            }
            c.np += nh;
            c.j0 = resume;
            next_row = c.j0 > n - 2u;                                    // the row's last column is behind us
            flush = next_row || c.np + kMaxChainHits > kPendMax;          // ... or the hit register is nearly full
            sync_first = true;
            if (!flush) {
                bx = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane((int)hv.z));
                by = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane((int)hv.w));
            }
        }
        if (flush) flush_pending<CONTROL, NT, SLOTS>(c, a, P, perm, ctl, lane, wave, sync_first);
        if (next_row) {
            c.i0 += 1u;
            c.j0 = c.i0 + 2u;
        }
        reload = flush || next_row;
    } else if (key == kNoKey) {
        if (CONTROL) a.since += (float)R * rowlen;
        c.i0 += R;
        c.j0 = c.i0 + 2u;
    } else {
        // pruned mode: one hit, applied at once by every wave (moves are rare here)
        const uint32_t is = key >> 16, js = key & 0xFFFFu;
        const uint32_t t0 = is >> 6, t1 = js >> 6;
        // A reversal inside ONE tile (most moves of the sweeps this shape runs in are a dozen positions long) leaves the tile's point
        // set, hence its box, as it is, and of its tour edges only (i, i+1) and (j, j+1) change: the larger of the two new squares
        // is folded into the tile's msq (an upper bound is all L0 needs) and no tile is marked stale — no rebuild, no barrier in
        // front of the next scan.  (Thread 0 alone writes positions i+1 and j, later in its own program order.)
        // (FOLD: only where the late phase takes the hit-free sweeps — a folded msq is never lowered again, and in a loop that also runs
        //  those sweeps the weaker bound costs more than the rebuilds: two descents per CU 174 -> 194 ms)
        const bool fold = FOLD && t0 == t1;
        if (fold && tid == 0) {
            const float m = fmaxf(sqdist(pt_get(P, is), pt_get(P, js)), sqdist(pt_get(P, is + 1u), pt_get(P, js + 1u)));  // (fmaxf drops a NaN, as build_tile_meta does)
            if (m > tmsq[t0]) tmsq[t0] = m;
        }
        reverse_segment<NT>(P, perm, is + 1u, js, tid);  // two_opt.rs:50,69-79  swap_2opt(path, i+1, j)
        TL_SYNC();
        if (CONTROL) {
            if (a.log) {
                if (lane == 0 && a.log_n < a.log_cap) a.log[a.log_n] = key;  // row << 16 | column
                a.log_n += 1u;
            }
            a.since += (float)(is - c.i0) * rowlen;
            a.moves += 1u;
            a.reversed += (uint64_t)(js - is);
            // For a hit in a later row of the block js - j0 wraps to a huge value: the estimate then keeps the pruned block
            // shape for a dozen moves, which measured faster than the "correct" gap (126 vs 137 ms per 256 restarts) — except
            // in the one step a requested change to the dense shape lags behind: there the true gap is taken, or that step's hit
            // would undo the request.
            a.since += (req != 0u && is > c.i0) ? (float)(js - (is + 2u)) : (float)(js - c.j0);
            a.gap_est = 0.5f * (a.gap_est + a.since);
            a.since = 0.0f;
            a.improved = true;
        }
        if (!fold) {  // L0 metadata of every tile that saw a changed position or tour-edge
            c.dirty_lo = t0 < c.dirty_lo ? t0 : c.dirty_lo;
            c.dirty_hi = t1 > c.dirty_hi ? t1 : c.dirty_hi;
        }
        c.i0 = is;
        c.j0 = js + 1u;
        if (c.j0 > n - 2u) {
            c.i0 += 1u;
            c.j0 = c.i0 + 2u;
        }
    }
    c.slot = c.slot == 2u ? 0u : c.slot + 1u;
    c.par ^= 1u;
    return c.i0 >= nrows || req != 0u;
}


// ------------------------------------------------------------------------------------------------
// Late phase: the sweeps behind the first one (from TwoOptNl::sweep_min - 1 on) that applied fewer than TwoOptNl::moves_max moves
// (two_opt_nl.hip has the argument).  By then a sweep applies a few hundred moves at most and nearly every row is decided "no move"; what such a row costs in the
// main loop is L0 over every tile plus L1 over its ~3 live ones (~200 instructions).  Here a row (a, b) reads ONE 128-byte
// record pair — the KA nearest cities of a (as c), the cities that have b among their KB nearest (as e) — adds the few "long"
// cities whose disc reaches the block, and runs the same exact cascade once over those <= 64 candidates; rows the lists cannot
// cover (an (a, b) longer than a's KA-th distance, a b with too long a reverse list) and sweeps with too many long cities walk
// their tiles as before.  The loop is the main loop's pruned shape only (any block shape gives the reference's moves), without a
// descriptor and without a control wave: every wave carries the whole cursor — row, column, key slot, dirty tiles, `improved` —,
// scans, and meets the same barriers; a step scans the REST of the sweep (rows taken a few at a time from an LDS counter), so a sweep
// costs one barrier per move; wave 0 keeps the accounting beside its scan, resets the next key slot and enters each move's new
// edges in the long list.
// State of a descent beside the tour: pos (city -> tour position), the long list (cities with a tour edge beyond their KB-th
// squared distance: x, y, the larger of their two tour edges squared, city) and one scratch row per wave.
struct NlLds {
    uint16_t *pos;
    uint4 *longe;
    uint16_t *surv;
};

// the larger of the two tour edges at position k, squared, as bits (squares are >= +0: bits order like the values, NaNs last);
// the closing edge p[n-1] -> p[0] is never a candidate's (c, e) and does not count
template <typename PT>
__device__ __forceinline__ uint32_t nl_r2_bits(const PT &P, uint32_t k, uint32_t n)
{
    const float2 pk = pt_get(P, k);
    uint32_t r2 = 0u;
    if (k >= 1u) r2 = __builtin_bit_cast(uint32_t, sqdist(pt_get(P, k - 1u), pk));
    if (k + 1u < n) {
        const uint32_t s = __builtin_bit_cast(uint32_t, sqdist(pk, pt_get(P, k + 1u)));
        r2 = s > r2 ? s : r2;
    }
    return r2;
}

// After the hit (fi, fj) — reversal of [fi+1 .. fj] — the tour edges at positions fi, fi+1, fj, fj+1 are new: their four cities
// get their entry's edge length refreshed, or an entry if they have become long.  One wave.
template <typename PT>
__device__ __forceinline__ void nl_fix(uint32_t fi, uint32_t fj, const PT &P, const uint16_t *perm, uint32_t n, Ctl *ctl, uint4 *longe, const uint32_t *__restrict__ dkb2,
                                       int lane)
{
    uint32_t cnt = (uint32_t)__builtin_amdgcn_readfirstlane((int)ctl->nlong);
    if (cnt > (uint32_t)kNlLongCap) return;  // an incomplete list stays unused until the next rebuild
    uint32_t p = lane == 0 ? fi : lane == 1 ? fi + 1u : lane == 2 ? fj : fj + 1u;
    p = p < n ? p : n - 1u;
    const uint32_t u = perm[p];
    const uint32_t r2 = nl_r2_bits(P, p, n);
    const float2 pk = pt_get(P, p);
    const uint32_t lng = r2 > dkb2[u] ? 1u : 0u;
    for (int q = 0; q < 4; ++q) {
        const uint32_t uq = readlane_u(u, (uint32_t)q), r2q = readlane_u(r2, (uint32_t)q);
        bool found = false;
        for (uint32_t base = 0; base < cnt && base < (uint32_t)kNlLongCap; base += 64u) {
            const uint32_t idx = base + (uint32_t)lane;
            const uint32_t city = idx < cnt ? longe[idx].w : 0xFFFFFFFFu;
            const uint64_t m = __builtin_amdgcn_ballot_w64(city == uq);
            if (m) {
                if (lane == 0) longe[base + (uint32_t)(__builtin_ffsll((long long)m) - 1)].z = r2q;
                found = true;
            }
        }
        if (!found && readlane_u(lng, (uint32_t)q)) {
            if (cnt < (uint32_t)kNlLongCap && lane == 0)
                longe[cnt] = make_uint4(readlane_u(__builtin_bit_cast(uint32_t, pk.x), (uint32_t)q), readlane_u(__builtin_bit_cast(uint32_t, pk.y), (uint32_t)q), r2q, uq);
            cnt += 1u;
        }
    }
    if (lane == 0) ctl->nlong = cnt;
}

// one row of the tile path: L0 (lanes = tiles) over the groups from the row's first column on, L1.. on the live tiles; the row's
// first improving column or 0xFFFFFFFF  (the main loop has the same code inline)
template <bool PRUNE, typename TC, typename PT>
__device__ __forceinline__ uint32_t row_tiles(const PT &P, uint32_t n, int G, const float4 *box, const float *msq, uint32_t jmin, float rax, float ray, float rbx,
                                              float rby, float rsqab, int lane, TC &tc)
{
    const uint32_t tmin = jmin >> 6;
#pragma unroll
    for (int gI = 0; gI < kMaxGroups; ++gI) {
        if (gI < G && (((uint32_t)gI + 1u) << 6) > tmin) {
            const uint32_t tl = ((uint32_t)gI << 6) + (uint32_t)lane;
            uint64_t m = __builtin_amdgcn_ballot_w64(tl >= tmin) &
                         (__builtin_amdgcn_ballot_w64(box_lb(rax, ray, box[gI]) < rsqab) | __builtin_amdgcn_ballot_w64(box_lb(rbx, rby, box[gI]) < msq[gI]));
            tc.l0 += 64u;
            while (m != 0) {
                const uint32_t t = ((uint32_t)gI << 6) + (uint32_t)(__builtin_ffsll((long long)m) - 1);
                m &= m - 1;
                tc.ptile += 1u;
                const uint64_t hm = tile_improving_mask<PRUNE>(P, n, t << 6, jmin, rax, ray, rbx, rby, rsqab, lane, tc);
                if (hm) return (t << 6) + (uint32_t)(__builtin_ffsll((long long)hm) - 1);
            }
        }
    }
    return 0xFFFFFFFFu;
}

// the first improving column among per-lane candidates (c = p[j], e = p[j+1], invalid lanes: j = 0xFFFFFFFF), or 0xFFFFFFFF
template <typename TC, typename PT>
__device__ __forceinline__ uint32_t nl_pass(const PT &P, uint32_t n, uint32_t j, uint32_t jmin, float rax, float ray, float rbx, float rby, float rsqab, int lane, TC &tc)
{
    const uint32_t jl = j < n - 2u ? j : n - 2u;  // (out-of-range columns are masked by the cascade's range test)
    const uint64_t m = tile_mask_core<true>(pt_get(P, jl), pt_get(P, jl + 1u), j, n, jmin, rax, ray, rbx, rby, rsqab, tc);
    if (!m) return 0xFFFFFFFFu;
    const uint32_t key = ((m >> lane) & 1ull) ? ~j : 0u;
    return ~readlane_u(wave_max_key_lane63(key), 63u);
}

template <int NT, typename TC, typename PT>
__device__ __forceinline__ void late_phase(const TwoOptBatchArgs &A, const PT &P, uint16_t *perm, float4 *tbox, float *tmsq, Ctl *ctl, const NlLds &L, uint32_t d,
                                           uint32_t n, uint32_t nrows, uint32_t ntile, int G, uint32_t dirty_lo, uint32_t dirty_hi, Acct &acct, uint32_t &sweeps,
                                           uint32_t &step, uint32_t &status, uint32_t &n_late_steps, TC &tc, int lane, int wave, int tid)
{
    constexpr int NW = NT / 64;
#ifndef TL_LATE_ROWS
#define TL_LATE_ROWS 8
#endif
    constexpr int kLateRowsPerWave = TL_LATE_ROWS;  // rows a wave takes at a time (a divisor of the 64-row chunk)
    static_assert(64 % kLateRowsPerWave == 0, "a take lies inside one chunk");
    const uint16_t *__restrict__ rec = A.nl.rec;
    const uint32_t *__restrict__ dkb2 = A.nl.dkb2;
    const bool control = wave == 0;  // wave 0 scans like the others and keeps the accounting beside it
    if (tid < 3) ctl->kr[tid] = make_uint2(kNoKey, 0u);
    TL_SYNC();
    uint32_t i0 = nrows, j0 = 2u, slot = 0u;  // handed over at the end of a sweep that moved something
    uint32_t fi = 0u, fj = 0u;
    bool improved = true, nl_ok = false, fix = false;
#ifdef TL_PROFILE_LATE
    uint64_t lq[6] = {0, 0, 0, 0, 0, 0};  // wave 1 of descent 0: cycles of chunk set-up, long-list prefilter, rows; rows; wait at B2; rest of the scan loop
#endif
    for (;;) {
        if (i0 >= nrows) {  // sweep finished (two_opt.rs:26-28)
            if (!improved) break;
            if (sweeps >= A.max_sweeps) {
                status = 1;
                break;
            }
            improved = false;
            ++sweeps;
            i0 = 0u;
            j0 = 2u;
            nl_ok = false;  // the long list is rebuilt once per sweep (entries that are no longer long go)
            if (control && acct.log) {
                if (lane == 0 && acct.log_n < acct.log_cap) acct.log[acct.log_n] = 0xFFFFFFFFu;
                acct.log_n += 1u;
            }
        }
        // A step scans the REST of the sweep under "no move yet": rows are dealt in chunks of 64 (the lane-resident row table), wave w
        // takes rows 4w .. 4w+3 of every chunk and goes from chunk to chunk without a barrier — the tour does not change while a
        // step scans — until it has posted a hit or finds one posted in an earlier row.  So a sweep costs one barrier per move.
        const uint32_t R = nrows - i0;
        uint32_t *keyslot = &ctl->kr[slot].x;
        const uint32_t slot_next = slot == 2u ? 0u : slot + 1u;
        if (control) {
            if (lane == 0) ctl->kr[slot_next].x = kNoKey;  // (its last readers are two barriers behind)
            ++step;
            ++n_late_steps;
        }
        // ---- in front of the scan, by every wave alike: stale tile boxes (each wave its share), the city -> position table and the
        // long list (rebuilt at a sweep's start, brought up to date after a move), one barrier (a rebuild: two)
        if (!nl_ok) {  // once per sweep every tile's bound afresh: a move inside one tile only ever RAISES that tile's msq (fold), so without
            dirty_lo = 0u;  // this the L0 bound of the rows that still walk tiles would loosen for the rest of the descent
            dirty_hi = ntile - 1u;
        }
        const bool dirty = dirty_lo <= dirty_hi;
        if (tid == 0) ctl->late_cur = i0;
        {
            if (dirty) {
                for (uint32_t t = dirty_lo + (uint32_t)wave; t <= dirty_hi; t += (uint32_t)NW) build_tile_meta(P, n, t, lane, tbox, tmsq);
                dirty_lo = 0xFFFFFFFFu;
                dirty_hi = 0;
            }
            if (!nl_ok) {
                for (uint32_t k = (uint32_t)tid; k < n; k += (uint32_t)NT) L.pos[perm[k]] = (uint16_t)k;
                if (tid == 0) ctl->nlong = 0u;
                TL_SYNC();
                for (uint32_t k = (uint32_t)tid; k < n; k += (uint32_t)NT) {
                    const uint32_t u = perm[k];
                    const uint32_t r2 = nl_r2_bits(P, k, n);
                    if (r2 > dkb2[u]) {
                        const uint32_t idx = atomicAdd(&ctl->nlong, 1u);
                        if (idx < (uint32_t)kNlLongCap) {
                            const float2 pk = pt_get(P, k);
                            L.longe[idx] = make_uint4(__builtin_bit_cast(uint32_t, pk.x), __builtin_bit_cast(uint32_t, pk.y), r2, u);
                        }
                    }
                }
                nl_ok = true;
            } else if (fix) {
                if (control) nl_fix(fi, fj, P, perm, n, ctl, L.longe, dkb2, lane);
            }
            fix = false;
            TL_SYNC();
        }
        {
            const uint32_t nlong = (uint32_t)__builtin_amdgcn_readfirstlane((int)ctl->nlong);
            const bool use_nl = nlong <= (uint32_t)kNlLongCap;
            const uint32_t iend = i0 + R;
            uint16_t *sv = L.surv + ((uint32_t)wave << 6);
            // the cities at the chunk's positions (lane l: position c0 + l, and c0 + l + 1 for the rows' b), one chunk ahead of the scan:
            // the first record of the next chunk is requested while this chunk's last row is decided
            auto load_ids = [&](uint32_t c0, uint32_t &ida, uint32_t &idb) {
                const uint32_t pa = c0 + (uint32_t)lane, pb = pa + 1u;
                ida = perm[pa < n ? pa : n - 1u];
                idb = perm[pb < n ? pb : n - 1u];
            };
            // a row's record pair: word l of a's record (the KA-th distance, a's nearest), of b's for the count and the reverse list
            auto load_rec = [&](uint32_t ida, uint32_t idb, uint32_t r) -> uint32_t {
                const uint32_t a_id = readlane_u(ida, r), b_id = readlane_u(idb, r);
                const uint32_t src = (lane < kNlRecB0 && lane != 2) ? a_id : b_id;
                return rec[(size_t)src * 64u + (uint32_t)lane];
            };
            // rows are taken kLateRowsPerWave at a time from a counter, in order: a wave that meets an expensive row (tiles) simply takes
            // fewer; the next take is made, and its first record requested, while the current rows are decided
            auto grab = [&]() -> uint32_t {
                uint32_t v = 0u;
                if (lane == 0) v = atomicAdd(&ctl->late_cur, (uint32_t)kLateRowsPerWave);
                return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
            };
            uint32_t g = grab();
            uint32_t ida = 0u, idb = 0u, recw = 0u;
            if (g < iend) {
                load_ids(i0 + ((g - i0) & ~63u), ida, idb);
                if (use_nl) recw = load_rec(ida, idb, (g - i0) & 63u);
            }
            bool stop = false;
#ifdef TL_PROFILE_LATE
            uint64_t tq = __builtin_amdgcn_s_memtime();
#define TL_LSTAMP(k) do { const uint64_t t_ = __builtin_amdgcn_s_memtime(); lq[k] += t_ - tq; tq = t_; } while (0)
#else
#define TL_LSTAMP(k) do { } while (0)
#endif
            while (g < iend && !stop) {
                const uint32_t c0 = i0 + ((g - i0) & ~63u), r0 = (g - i0) & 63u;  // the chunk of 64 rows (the lane-resident row table) these rows lie in
                const uint32_t gn = grab();
                const uint32_t r0n = (gn - i0) & 63u;
                TL_LSTAMP(5);
                // lane-resident row table: lane l holds P[c0+l], P[c0+l+1] and their squared distance
                const float2 rp = pt_get(P, c0 + (uint32_t)lane);
                const float2 rq = pt_get(P, c0 + (uint32_t)lane + 1u);
                const float rowsq = sqdist(rp, rq);
                const uint32_t cida = ida, cidb = idb;
                if (gn < iend) load_ids(i0 + ((gn - i0) & ~63u), ida, idb);
                // the long cities that can matter for this chunk: sq(b, e) >= lb(e, box of the chunk's b's) >= r2(e) >= sq(c, e) rules e
                // out for every row of it.  Every b of the chunk lies in the boxes of the (at most two) tiles its positions fall in.
                uint32_t surv8 = 0u, ns = 0u;
                TL_LSTAMP(0);
                if (use_nl && nlong) {
                    const uint32_t tA = c0 >> 6, tB1 = (c0 + 64u) >> 6, tB = tB1 < ntile ? tB1 : ntile - 1u;
                    const float4 ba = tbox[tA], bb = tbox[tB];
                    const float4 bu = make_float4(fminf(ba.x, bb.x), fminf(ba.y, bb.y), fmaxf(ba.z, bb.z), fmaxf(ba.w, bb.w));
                    for (uint32_t base = 0; base < nlong; base += 64u) {
                        const uint32_t idx = base + (uint32_t)lane;
                        const uint4 le = L.longe[idx < nlong ? idx : 0u];
                        const float lb = box_lb(__builtin_bit_cast(float, le.x), __builtin_bit_cast(float, le.y), bu);
                        const bool live = idx < nlong && !(lb >= __builtin_bit_cast(float, le.z));
                        const uint64_t m = __builtin_amdgcn_ballot_w64(live);
                        if (m) {
                            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                            if (live && ns + rank < 64u) sv[ns + rank] = (uint16_t)le.w;
                            ns += (uint32_t)__builtin_popcountll(m);
                        }
                    }
                    if (ns) surv8 = sv[lane >= kNlSurv0 ? lane - kNlSurv0 : 0];
                }
                const bool nl_rows = use_nl && ns <= 64u;  // (more survivors than a pass holds: this wave walks tiles, and decides the same)
                if (ns < (uint32_t)kNlSurvSlots) surv8 = (uint32_t)lane >= (uint32_t)kNlSurv0 + ns ? 0xFFFFu : surv8;  // (0xFFFF: no city)
                TL_LSTAMP(1);
                {   // a hit posted in a row before this wave's rows of the chunk makes them, and every later one, moot
                    const uint32_t kb = (uint32_t)__builtin_amdgcn_readfirstlane((int)*keyslot);
                    if (kb != kNoKey && (kb >> 16) < g) break;
                }
#pragma unroll
                for (uint32_t q = 0; q < (uint32_t)kLateRowsPerWave; ++q) {
                    const uint32_t r = r0 + q, i = c0 + r;
                    if (i >= iend) break;
                    const uint32_t cur = recw;
                    if (use_nl) {  // the next row's record is on its way while this one is decided
                        if (q + 1u < (uint32_t)kLateRowsPerWave) recw = load_rec(cida, cidb, r + 1u);  // (past the sweep's end: some valid city's, unused)
                        else recw = load_rec(ida, idb, r0n);                                           // (no next take: this chunk's cities again, unused)
                    }
                    const float rax = readlane_f(rp.x, r), ray = readlane_f(rp.y, r);
                    const float rbx = readlane_f(rq.x, r), rby = readlane_f(rq.y, r);
                    const float rsqab = readlane_f(rowsq, r);
                    const uint32_t jmin = (i == i0) ? j0 : (i + 2u);
                    tc.prow += 1u;
                    uint32_t col = 0xFFFFFFFFu;
                    bool listed = false;
                    if (nl_rows) {
                        // word 0 (a's): the high half of a's KA-th squared distance — an (a, b) that is not below it could have a closer
                        // c outside a's list; word 2 (b's): set when b's reverse list is incomplete.  Either way: tiles.
                        const uint32_t w = cur & 0xFFFFu;
                        const bool bad = lane == 0 ? !((__builtin_bit_cast(uint32_t, rsqab) >> 16) < w) : (lane == 2 && w != 0u);
                        if (!__builtin_amdgcn_ballot_w64(bad)) {
                            listed = true;
                            uint32_t u = lane >= kNlSurv0 ? surv8 : w;
                            const bool valid = lane >= kNlRecA0 && u != 0xFFFFu;  // (empty slots hold 0xFFFF)
                            u = valid ? u : 0u;
                            const uint32_t pu = L.pos[u];
                            uint32_t j = lane < kNlRecB0 ? pu : pu - 1u;  // a's neighbour is c = p[j]; the others are e = p[j+1]
                            j = valid ? j : 0xFFFFFFFFu;
                            col = nl_pass(P, n, j, jmin, rax, ray, rbx, rby, rsqab, lane, tc);
                            if (ns > (uint32_t)kNlSurvSlots) {  // more long cities than the pass has spare lanes: one more pass
                                const uint32_t q2 = (uint32_t)kNlSurvSlots + (uint32_t)lane;
                                const bool v2 = q2 < ns;
                                const uint32_t u2 = v2 ? sv[q2 < 64u ? q2 : 0u] : 0u;
                                const uint32_t p2 = L.pos[u2];
                                const uint32_t c2 = nl_pass(P, n, v2 ? p2 - 1u : 0xFFFFFFFFu, jmin, rax, ray, rbx, rby, rsqab, lane, tc);
                                col = c2 < col ? c2 : col;
                            }
                        }
                    }
                    if (!listed) {  // the tile path: L0 (lanes = tiles) against the boxes, L1.. on the live tiles
                        float4 box[kMaxGroups];
                        float msq[kMaxGroups];
#pragma unroll
                        for (int gI = 0; gI < kMaxGroups; ++gI) {
                            if (gI < G) {
                                box[gI] = tbox[((uint32_t)gI << 6) + (uint32_t)lane];
                                msq[gI] = tmsq[((uint32_t)gI << 6) + (uint32_t)lane];
                            }
                        }
                        col = row_tiles<true>(P, n, G, box, msq, jmin, rax, ray, rbx, rby, rsqab, lane, tc);
                    }
                    TL_LSTAMP(2);
#ifdef TL_PROFILE_LATE
                    lq[3] += 1;
#endif
                    if (col == 0xFFFFFFFFu) continue;  // nothing in this row
                    // (fenced: the way from here to the barrier holds no tracked LDS operation of this wave, tl_device.h)
                    if (lane == 0) lds_min_u32_fenced(keyslot, (i << 16) | col);
                    stop = true;  // this wave's later rows are later rows
                    break;
                }
                g = gn;
            }
            TL_LSTAMP(5);
        }
#ifdef TL_PROFILE_LATE
        const uint64_t tb2 = __builtin_amdgcn_s_memtime();
#endif
        TL_SYNC();  // B2
#ifdef TL_PROFILE_LATE
        lq[4] += __builtin_amdgcn_s_memtime() - tb2;
#endif
        const uint32_t key = (uint32_t)__builtin_amdgcn_readfirstlane((int)ctl->kr[slot].x);
        if (key == kNoKey) {
            i0 += R;
            j0 = i0 + 2u;
        } else {
            const uint32_t is = key >> 16, js = key & 0xFFFFu;
            const uint32_t t0 = is >> 6, t1 = js >> 6;
            if (t0 == t1 && tid == 0) {  // a reversal inside one tile: its box stands, its msq takes the new edges (step_boundary)
                const float m = fmaxf(sqdist(pt_get(P, is), pt_get(P, js)), sqdist(pt_get(P, is + 1u), pt_get(P, js + 1u)));
                if (m > tmsq[t0]) tmsq[t0] = m;
            }
            reverse_segment<NT>(P, perm, is + 1u, js, tid, L.pos);  // two_opt.rs:50,69-79  swap_2opt(path, i+1, j)
            TL_SYNC();
            if (control) {
                if (acct.log) {
                    if (lane == 0 && acct.log_n < acct.log_cap) acct.log[acct.log_n] = key;  // row << 16 | column
                    acct.log_n += 1u;
                }
                acct.moves += 1u;
                acct.reversed += (uint64_t)(js - is);
            }
            improved = true;
            if (t0 != t1) {  // L0 metadata of every tile that saw a changed position or tour-edge
                dirty_lo = t0 < dirty_lo ? t0 : dirty_lo;
                dirty_hi = t1 > dirty_hi ? t1 : dirty_hi;
            }
            fix = true;  // the move's two new edges are entered in the long list in front of the next scan
            fi = is;
            fj = js;
            i0 = is;
            j0 = js + 1u;
            if (j0 > n - 2u) {
                i0 += 1u;
                j0 = i0 + 2u;
            }
        }
        slot = slot_next;
    }
#ifdef TL_PROFILE_LATE
    if (d == 0 && lane == 0 && (wave == 1 || wave == 9))
        printf("late phase, wave %d: chunk set-up %lu, prefilter %lu, rows %lu cycles in %lu rows (%lu per row), B2 wait %lu, other %lu\n", wave, lq[0], lq[1], lq[2], lq[3],
               lq[2] / (lq[3] ? lq[3] : 1), lq[4], lq[5]);
#endif
    (void)d;
}

}  // namespace

// NT = 1024 (16 waves) is the form of a descent that has a CU to itself.  Where the LDS holds two or four tours (n <= ~7100 /
// ~3000) and the batch has more descents than the chip has CUs, the 8- or 4-wave forms run 2 or 4 descents per CU: a descent
// leaves its SIMDs idle ~80 % of the time and half the waves cost it only 7 %, so a neighbour's descent
// fills the issue slots — 1.57 x the restarts per second at n = 7000.  At most 128 VGPRs (4 waves per SIMD) in every form.
template <int NT, bool PRUNE, bool COUNT, bool FX, bool NL = false>
__global__ __launch_bounds__(NT, 4) void k_two_opt_ref_lds(TwoOptBatchArgs A)
{
    static_assert(!NL || (PRUNE && !FX), "late phase: the float2 forms with pruning");
    constexpr int kSlots = FX ? kFlushSlotsFx : kFlushSlots;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NW = NT / 64;
    static_assert(NW == 16 || NW == 8 || NW == 4, "16 waves; 8 or 4 for two or four descents per CU");
    const uint32_t n = A.n, npad = A.n_pad;
    const uint32_t ntile = npad >> 6;                  // tiles incl. the pad tile
    const int G = (int)((ntile + 63u) >> 6);           // 64-tile groups (<= kMaxGroups)
    // tour-ordered points: float2 (8 B per city), or grid coordinates (4 + 1 B, PtsFx) — then two tours of n = 10^4 fit one CU
    using PT = typename std::conditional<FX, PtsFx, float2 *>::type;
    PT P;
    uint16_t *perm;
    unsigned char *tailp;
    if constexpr (FX) {
        P.lo = reinterpret_cast<uint32_t *>(smem);
        P.hi = reinterpret_cast<uint8_t *>(smem + (size_t)npad * 4);
        P.inv = A.fx_inv;
        perm = reinterpret_cast<uint16_t *>(smem + (size_t)npad * 5);
        tailp = smem + (size_t)npad * 7;
    } else {
        P = reinterpret_cast<float2 *>(smem);
        perm = reinterpret_cast<uint16_t *>(smem + (size_t)npad * 8);
        tailp = smem + (size_t)npad * 10;
    }
    float4 *tbox = reinterpret_cast<float4 *>(tailp);                        // kMaxGroups*64 entries
    float *tmsq = reinterpret_cast<float *>(tailp + kMaxGroups * 64 * 16);     // kMaxGroups*64
    Ctl *ctl = reinterpret_cast<Ctl *>(tailp + kMaxGroups * 64 * 20);
    // hit lists (kQCap words per (wave, step parity)) during the descent; reused as NT floats for the cost sum
    uint32_t *queues = reinterpret_cast<uint32_t *>(reinterpret_cast<unsigned char *>(ctl) + kCtlBytes);
    float *scratch = reinterpret_cast<float *>(queues);
    NlLds L{nullptr, nullptr, nullptr};  // late phase: city -> position, long list, per-wave scratch (behind the hit lists)
    if constexpr (NL) {
        unsigned char *nlp = reinterpret_cast<unsigned char *>(queues) + 32 * kQCap * 4;
        L.pos = reinterpret_cast<uint16_t *>(nlp);
        L.longe = reinterpret_cast<uint4 *>(nlp + (size_t)npad * 2);
        L.surv = reinterpret_cast<uint16_t *>(nlp + (size_t)npad * 2 + (size_t)kNlLongCap * 16);
    }

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform -> SGPR loop control
    const uint32_t d = blockIdx.x;
    const float2 *__restrict__ xy = A.xy;

    // ---------------------------------------------------------------- initial tour
    if (A.init_mode == TL_INIT_ARRAY) {
        // a device-resident initial tour has not been through the host's permutation check (tl_two_opt_batch_dev): an
        // entry >= n would index xy out of bounds, so the descent is refused (status 2, cost NaN, tour untouched)
        const uint32_t *__restrict__ src = A.init + (size_t)d * n;
        if (tid == 0) ctl->bad_init = 0u;
        TL_SYNC();
        bool bad = false;
        for (uint32_t k = tid; k < n; k += NT) {
            const uint32_t v = src[k];
            bad |= v >= n;
            perm[k] = (uint16_t)v;
        }
        if (bad) ctl->bad_init = 1u;
        TL_SYNC();
        if (ctl->bad_init) {
            if (tid == 0) {
                A.out_cost[d] = __builtin_nanf("");
                uint64_t *st = A.out_stats + (size_t)d * TL_STATS_STRIDE;
                for (int q = 0; q < TL_STATS_STRIDE; ++q) st[q] = 0;
                st[3] = 2;
            }
            return;
        }
    } else {
        for (uint32_t k = tid; k < n; k += NT) perm[k] = (uint16_t)k;  // two_opt.rs:18-20
    }
    if (A.init_mode == TL_INIT_SEEDED) {
        // Fisher-Yates `for i in (1..n).rev(): j = rng % (i+1); swap` from splitmix64(seed + r):
        // the draws are counter-based, so compute them in parallel, then one lane applies the swaps.
        uint16_t *draws = reinterpret_cast<uint16_t *>(smem);  // the points are not live yet
        const uint64_t s = A.seed + (uint64_t)(A.first + d);
        for (uint32_t i = 1 + tid; i < n; i += NT) {
            const uint64_t kth = (uint64_t)(n - 1 - i);  // i = n-1 is draw 0
            draws[i] = (uint16_t)(splitmix64_at(s, kth) % ((uint64_t)i + 1));
        }
        TL_SYNC();
        if (tid == 0) {
            for (uint32_t i = n - 1; i >= 1; --i) {
                const uint32_t j = draws[i];
                const uint16_t t = perm[i];
                perm[i] = perm[j];
                perm[j] = t;
            }
        }
    }
    if (tid < 4) ctl->kr[tid] = make_uint2(kNoKey, 0u);
    if (tid < 6) ctl->cnt[tid] = 0ull;
    if (tid == 0) {
        ctl->nlong = 0u;
        ctl->clk0 = __builtin_amdgcn_s_memtime();
        ctl->rt0 = __builtin_amdgcn_s_memrealtime();
    }
    for (uint32_t k = tid; k < (uint32_t)kMaxGroups * 64u; k += NT) {
        const float inf = __builtin_inff();
        tbox[k] = make_float4(inf, inf, -inf, -inf);  // empty box: never live
        tmsq[k] = -1.0f;
    }
    TL_SYNC();
    if constexpr (FX) {
        for (uint32_t k = tid; k < npad; k += NT) {
            const uint2 g = (k < n) ? A.fx_xy[perm[k]] : make_uint2(0u, 0u);
            pt_put(P, k, PtsFx::Raw{g.x, g.y});
        }
    } else {
        for (uint32_t k = tid; k < npad; k += NT) P[k] = (k < n) ? xy[perm[k]] : make_float2(0.f, 0.f);
    }
    TL_SYNC();
    for (uint32_t t = (uint32_t)wave; t < ntile; t += NW) build_tile_meta(P, n, t, lane, tbox, tmsq);
    TL_SYNC();

    // ---------------------------------------------------------------- descent
    const uint32_t nrows = n - 3;  // rows i in [0, n-3); j in [i+2, n-2]
    const uint32_t last_tile = n >= 2 ? (n - 2u) >> 6 : 0u;
    uint32_t sweeps = 1, step = 0, status = 0;
    uint64_t moves = 0, reversed = 0;
    uint64_t rev_lane = 0;        // per-lane share of `reversed` from the flushes (summed over lanes at the end)
    typename std::conditional<COUNT, TileCounts, NoCounts>::type tc;  // work really done by this wave's cascade (SALU counters)

    // Role split.  The reference's loop is a chain of ~35 k steps per descent (n = 10^4, random start), and what a step costs is not
    // only its scan but the instructions EVERY wave executes around it: four waves share a SIMD's issue port, so an instruction
    // all 16 waves run costs 16 cycles per step and a dependent instruction of a lone wave 8 (tests/probes/step_sync_probe.hip;
    // an s_barrier of 16 waves: 21).  So the descent's ACCOUNTING — move / reversal / sweep counters, the gap estimate behind the
    // block shape, the sweep-end and exit decisions — lives in ONE wave (wave 0, "control"), which does not scan, and the other
    // waves ("workers") carry only the cursor a scan needs: (i0, j0, pending hits, block shape, key slot).
    //  * Every wave advances that cursor by itself at a step's end (step_boundary): it is a function of words all waves read
    //    behind the same barrier — the step's key, the owner's hit list (count, resume column, the row's new b), the control
    //    wave's request word — so the workers go from step to step without waiting for anybody; the control wave runs the
    //    same function (it must meet every barrier) and does its accounting beside the workers' next scan.
    //  * The control wave steps in — an 8-byte descriptor (op | pruned << 2, i0 | j0 << 16) before barrier B0 — only where a
    //    decision is its own: a sweep has ended (next sweep or exit) or the block shape should change (requested through
    //    the request word beside the next step's key slot, i.e. one boundary ahead: either shape gives the same results, so
    //    the request may lag).
    //  * Barriers per step, met by every wave:  (B0 | flush: 2) | dense: B1 (lead round / rest), B2 | pruned: (tile
    //    metadata: 1), B2 | boundary: (flush behind a hit step: 1 + 2; flush: 2; pruned hit: 1).
    // Dense keys are (column << 16) | posting wave (the row is implied; the tag finds the owner's hit list), pruned keys
    // (row << 16) | column.  Key slots rotate with step % 3 — the control wave resets the next step's slot before B2; the slot's
    // second word is the request, so a wave reads both with one 8-byte LDS read — and
    // every wave has two hit lists, used by the step's parity (a list is read after B2 while its owner may already write the
    // next step's).  The owner of a step's hits files them in ctl->pend itself (it still holds them in a register).
    enum : uint32_t { OP_GO = 0u, OP_LATE = 1u, OP_EXIT = 2u };  // OP_LATE: the main loop ends here, the late phase takes the next sweep
    bool go_late = false;
    uint32_t n_late_steps = 0;
    uint64_t late_clk = 0;  // shader clocks of the late phase (stats word 13, high part)
    constexpr int NWK = NW - 1;                                  // workers
    constexpr uint32_t kLead = kDenseLead < (uint32_t)NWK ? kDenseLead : (uint32_t)NWK;
    static_assert(4u <= kQCap, "a hit list fits its slot");
    if (n >= 4 && wave == 0) {
        // ------------------------------------------------------------ control wave
        Cursor c{0u, 2u, 0u, 0u, 0u, 0xFFFFFFFFu, 0u, false};
        Acct acct{0ull, 0ull, 0ull, 0.0f, 0.0f, false, A.move_log ? A.move_log + (size_t)d * A.log_cap : nullptr, A.log_cap, 0u};
        bool need_desc = true;
        uint32_t n_desc = 0, n_pruned_steps = 0;  // diagnostics (stats words 13, 14)
        uint64_t sweep_moves0 = 0;                // acct.moves when the current sweep began
#ifdef TL_PROFILE2
        uint64_t q2[13] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        uint64_t t_it = __builtin_amdgcn_s_memtime();
#endif
#ifdef TL_PROFILE4
        // per sweep (printed for descent 0 of the launch): steps and cycles by block shape and outcome; for dense steps without a
        // hit (the step that ends a row: the rest of the row + the row's flush) also the tiles they scanned and how many of those
        // lay outside the stale-box range (where an L0 bound would have been valid)
        uint64_t q4[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        uint64_t q4c[3] = {0, 0, 0}, t4b = __builtin_amdgcn_s_memtime();  // control wave: cycles from B2 to its arrival at B1, its wait at B1, dense steps
        uint64_t q4m[3] = {0, 0, 0};  // tile-box rebuilds in front of pruned steps: cycles (to the barrier behind them), count, tiles
        uint64_t q4l[4] = {0, 0, 0, 0};  // dense steps with a hit, split: first hit inside the lead tiles (steps, cycles) / behind them
        uint64_t t4 = __builtin_amdgcn_s_memtime();
#endif
        for (;;) {
            if (need_desc) {
                ++n_desc;
                bool done = false, new_sweep = false;
                if (c.i0 >= nrows) {  // sweep finished (two_opt.rs:26-28)
                    if (!acct.improved) {
                        done = true;
                    } else if (sweeps >= A.max_sweeps) {
                        status = 1;
                        done = true;
                    } else if (NL && sweeps + 1u >= A.nl.sweep_min && acct.moves - sweep_moves0 < (uint64_t)A.nl.moves_max) {
                        go_late = true;  // (few moves per sweep from here on: the late phase's cheaper rows pay for its dearer moves)
                    } else {
#ifdef TL_PROFILE4
                        if (d == 0 && lane == 0)
                            printf("sweep %u: dense hit in lead tiles %lu steps %lu cyc, behind %lu steps %lu cyc\n", sweeps, q4l[0], q4l[1], q4l[2], q4l[3]);
                        for (int q = 0; q < 4; ++q) q4l[q] = 0;
                        if (d == 0 && lane == 0)
                            printf("sweep %u: dense hit %lu steps %lu cyc | dense none %lu steps %lu cyc, tiles %lu of which with valid boxes %lu | pruned hit %lu steps %lu cyc | pruned none %lu steps %lu cyc\n",
                                   sweeps, q4[0], q4[1], q4[2], q4[3], q4[8], q4[9], q4[4], q4[5], q4[6], q4[7]);
                        for (int q = 0; q < 10; ++q) q4[q] = 0;
#endif
                        acct.improved = false;
                        ++sweeps;
                        sweep_moves0 = acct.moves;
                        c.i0 = 0;
                        c.j0 = 2;
                        if (NL) {  // folded tile bounds (step_boundary) are rebuilt with the next stale ones; the workers do the same (bit 3)
                            c.dirty_lo = 0u;
                            c.dirty_hi = ntile - 1u;
                            new_sweep = true;
                        }
                        if (acct.log) {  // a new sweep begins here (the same row can hold moves of two consecutive sweeps back to back)
                            if (lane == 0 && acct.log_n < acct.log_cap) acct.log[acct.log_n] = 0xFFFFFFFFu;
                            acct.log_n += 1u;
                        }
                    }
                }
                // block shape: dense (moves every few rows: one row, every tile) or pruned (up to kRMax rows, L0)
                c.pruned = PRUNE && !done && fmaxf(acct.gap_est, acct.since) > TL_DENSE_ROWS * (float)(n - 2u - c.i0);
                if (lane == 0) {
                    *reinterpret_cast<uint2 *>(ctl->desc) = make_uint2((done ? OP_EXIT : go_late ? OP_LATE : OP_GO) | (c.pruned ? 4u : 0u) | (new_sweep ? 8u : 0u), c.i0 | (c.j0 << 16));
                    if (NL) ctl->desc[2] = sweeps;
                }
                TL_SYNC();  // B0
                if (lane == 0) {  // behind B0: a slow worker may have been reading its request word until it got here
                    ctl->kr[0].y = 0u;
                    ctl->kr[1].y = 0u;
                    ctl->kr[2].y = 0u;
                }
                if (c.np) flush_pending<true, NT, kSlots>(c, acct, P, perm, ctl, lane, wave, false);
                if (done || go_late) break;
            }
            const uint32_t slot_next = c.slot == 2u ? 0u : c.slot + 1u;
            if (lane == 0) ctl->kr[slot_next].x = kNoKey;  // slot of the next step (its last readers are two barriers behind)
            ++step;
            n_pruned_steps += c.pruned ? 1u : 0u;
            const uint32_t R = c.pruned ? ((uint32_t)kRMax < nrows - c.i0 ? (uint32_t)kRMax : nrows - c.i0) : 1u;
            if (c.pruned) {
                if (c.dirty_lo <= c.dirty_hi) {  // stale tile boxes: every wave takes its share (this one has nothing else to do here)
#ifdef TL_PROFILE4
                    const uint64_t tm0 = __builtin_amdgcn_s_memtime();
                    q4m[2] += c.dirty_hi - c.dirty_lo + 1u;
#endif
                    for (uint32_t t = c.dirty_lo; t <= c.dirty_hi; t += (uint32_t)NW) build_tile_meta(P, n, t, lane, tbox, tmsq);
                    c.dirty_lo = 0xFFFFFFFFu;
                    c.dirty_hi = 0;
                    TL_SYNC();
#ifdef TL_PROFILE4
                    q4m[0] += __builtin_amdgcn_s_memtime() - tm0;
                    q4m[1] += 1;
#endif
                }
            } else {
#ifdef TL_PROFILE4
                const uint64_t tb0 = __builtin_amdgcn_s_memtime();
#endif
                TL_SYNC();  // B1: lead round
#ifdef TL_PROFILE4
                {   // the control wave's own way to B1 (from the previous step's B2) and its wait there
                    const uint64_t tb1 = __builtin_amdgcn_s_memtime();
                    q4c[0] += tb0 - t4b;
                    q4c[1] += tb1 - tb0;
                    q4c[2] += 1;
                }
#endif
            }
            TL_SYNC();  // B2
#ifdef TL_PROFILE4
            t4b = __builtin_amdgcn_s_memtime();
#endif
#ifdef TL_PROFILE2
            const bool was_pruned = c.pruned;
            const uint64_t moves_before = acct.moves;
#endif
#ifdef TL_PROFILE4
            const bool was_pruned4 = c.pruned;
            const uint64_t moves_before4 = acct.moves;
            const uint32_t t0_4 = c.j0 >> 6, dlo4 = c.dirty_lo, dhi4 = c.dirty_hi;
            const uint32_t key4 = (uint32_t)__builtin_amdgcn_readfirstlane((int)ctl->kr[c.slot].x);
#endif
            float bx_ = 0.f, by_ = 0.f;
            bool reload_ = false;
            need_desc = step_boundary<true, NT, kSlots, NL>(c, acct, P, perm, ctl, queues, n, nrows, R, lane, wave, tid, 0u, bx_, by_, reload_, tmsq);
            // the block shape the gap estimate asks for; a change is requested for the next boundary
            if (PRUNE && c.i0 < nrows) {
                const bool want = fmaxf(acct.gap_est, acct.since) > TL_DENSE_ROWS * (float)(n - 2u - c.i0);
                if (lane == 0) ctl->kr[c.slot].y = want != c.pruned ? 1u : 0u;  // (c.slot is the next step's slot by now)
            }
#ifdef TL_PROFILE4
            {
                const uint64_t t2 = __builtin_amdgcn_s_memtime();
                const bool none = acct.moves == moves_before4;
                const int b = (was_pruned4 ? 4 : 0) + (none ? 2 : 0);
                q4[b] += 1;
                q4[b + 1] += t2 - t4;
                if (!was_pruned4 && !none) {
                    const int l = ((key4 >> 16) >> 6) < t0_4 + kLead ? 0 : 2;
                    q4l[l] += 1;
                    q4l[l + 1] += t2 - t4;
                }
                t4 = t2;
                if (!was_pruned4 && none) {
                    const uint32_t tiles = last_tile - t0_4 + 1u;
                    const uint32_t lo = dlo4 > t0_4 ? dlo4 : t0_4, hi = dhi4 < last_tile ? dhi4 : last_tile;
                    const uint32_t stale = (dlo4 <= dhi4 && lo <= hi) ? hi - lo + 1u : 0u;
                    q4[8] += tiles;
                    q4[9] += tiles - stale;
                }
            }
#endif
#ifdef TL_PROFILE2
            {
                const uint64_t t2 = __builtin_amdgcn_s_memtime();
                const int b = (was_pruned ? 0 : 6) + (acct.moves == moves_before ? 3 : 0);
                q2[b] += 1;
                q2[b + 1] += t2 - t_it;
                t_it = t2;
            }
#endif
        }
        if constexpr (NL) {
            if (go_late) {
                late_clk = __builtin_amdgcn_s_memtime();
                late_phase<NT>(A, P, perm, tbox, tmsq, ctl, L, d, n, nrows, ntile, G, c.dirty_lo, c.dirty_hi, acct, sweeps, step, status, n_late_steps, tc, lane, wave, tid);
                late_clk = __builtin_amdgcn_s_memtime() - late_clk;
            }
        }
        moves = acct.moves;
        reversed = acct.reversed;
        rev_lane = acct.rev_lane;
#ifdef TL_PROFILE4
        if (d == 0 && lane == 0)
            printf("sweep %u: dense hit in lead tiles %lu steps %lu cyc, behind %lu steps %lu cyc\n", sweeps, q4l[0], q4l[1], q4l[2], q4l[3]);
        if (d == 0 && lane == 0)
            printf("tile-box rebuilds: %lu, %lu cycles and %lu tiles apiece\n", q4m[1], q4m[0] / (q4m[1] ? q4m[1] : 1), q4m[2] / (q4m[1] ? q4m[1] : 1));
        if (d == 0 && lane == 0)
            printf("control wave, dense steps: %lu; per step from B2 to its arrival at B1 %lu cycles, its wait at B1 %lu\n", q4c[2], q4c[0] / (q4c[2] ? q4c[2] : 1), q4c[1] / (q4c[2] ? q4c[2] : 1));
        if (d == 0 && lane == 0)
            printf("sweep %u: dense hit %lu steps %lu cyc | dense none %lu steps %lu cyc, tiles %lu of which with valid boxes %lu | pruned hit %lu steps %lu cyc | pruned none %lu steps %lu cyc\n",
                   sweeps, q4[0], q4[1], q4[2], q4[3], q4[8], q4[9], q4[4], q4[5], q4[6], q4[7]);
#endif
#if !defined(TL_PROFILE2) && !defined(TL_PROFILE3)
        if (tid == 0) {
            uint64_t *st = A.out_stats + (size_t)d * TL_STATS_STRIDE;
            st[13] = (uint64_t)n_desc | (late_clk << 24);  // descriptors published (sweep ends + block-shape changes) | shader clocks of the late phase << 24
            st[14] = (uint64_t)(n_pruned_steps + n_late_steps) | ((uint64_t)n_late_steps << 32);  // steps in pruned shape | of those, late-phase steps << 32
            st[15] = acct.log_n;      // words offered to the move log: moves + sweep marks (more than its capacity: the log is a prefix)
        }
#endif
#ifdef TL_PROFILE2
        if (tid == 0) {
            uint64_t *st = A.out_stats + (size_t)d * TL_STATS_STRIDE;
            st[5] = q2[0]; st[6] = q2[1]; st[7] = q2[2]; st[8] = q2[3]; st[9] = q2[4]; st[10] = q2[5];
            st[11] = q2[6]; st[12] = q2[7]; st[13] = q2[9]; st[14] = q2[10]; st[15] = q2[12];
        }
#endif
    } else if (n >= 4) {
        // ------------------------------------------------------------ worker waves
        const int wk = wave - 1;
#ifdef TL_PROFILE3
        // dense steps of this wave by outcome (0: first hit inside the lead tiles, 1: behind them, 2: none): cycles from the previous
        // step's B2 to the lead round (boundary, flushes, row set-up), lead tile, B1 wait, round 2, B2 wait; [5] steps.  Printed by
        // wave 1 (a lead wave) and wave 6 of descent 0.
        uint64_t q3[3][6] = {{0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0}};
        uint64_t d3[5] = {0, 0, 0, 0, 0};
        uint64_t t3 = __builtin_amdgcn_s_memtime();
#define TL_STAMP3(k) do { const uint64_t t_ = __builtin_amdgcn_s_memtime(); d3[k] = t_ - t3; t3 = t_; } while (0)
#else
#define TL_STAMP3(k) do { } while (0)
#endif
        Cursor c{0u, 2u, 0u, 0u, 0u, 0xFFFFFFFFu, 0u, false};
        Acct acct{0ull, 0ull, 0ull, 0.0f, 0.0f, false, nullptr, 0u, 0u};  // unused by a worker
#ifdef TL_PROFILE4
        uint64_t qp[5] = {0, 0, 0, 0, 0};  // pruned steps of this wave: cycles in the scan, rows, tile passes, steps, wait at B2
#endif
        bool need_desc = true, reload = true;
        float ax = 0.f, ay = 0.f, bx = 0.f, by = 0.f;
        for (;;) {
            if (need_desc) {
                TL_SYNC();  // B0
                const uint2 dv = *reinterpret_cast<const uint2 *>(ctl->desc);
                const uint32_t w0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)dv.x), w1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)dv.y);
                if (c.np) flush_pending<false, NT, kSlots>(c, acct, P, perm, ctl, lane, wave, false);
                if ((w0 & 3u) != OP_GO) {
                    go_late = NL && (w0 & 3u) == OP_LATE;
                    if (NL) sweeps = (uint32_t)__builtin_amdgcn_readfirstlane((int)ctl->desc[2]);
                    break;
                }
                c.pruned = (w0 & 4u) != 0u;
                c.i0 = w1 & 0xFFFFu;
                c.j0 = w1 >> 16;
                reload = true;
                if (NL && (w0 & 8u)) {  // a sweep begins: folded tile bounds (step_boundary) are rebuilt with the next stale ones
                    c.dirty_lo = 0u;
                    c.dirty_hi = ntile - 1u;
                }
            }
            uint32_t *keyslot = &ctl->kr[c.slot].x;
            uint32_t my_hits = 0;
            const uint32_t R = c.pruned ? ((uint32_t)kRMax < nrows - c.i0 ? (uint32_t)kRMax : nrows - c.i0) : 1u;
            if (!c.pruned) {
                // ---- dense step: one row, every tile from the resume column on.  Lead round: kLead waves (one per SIMD) look at
                // the first tiles, everybody else parks at B1 — moves come every few candidates here, and a wave chaining hits
                // runs ~3x faster when it does not share its SIMD's issue slots with busy neighbours.  Then (no hit yet) all
                // workers take the rest of the row.  a = p[i]; b = p[i+1], or after a step with hits the old p[g_last]
                // (two_opt.rs:50), which the owner of the hits left in its list.
                if (reload) {
                    const float2 ab = pt_get(P, c.i0 + (lane == 0 ? 0u : 1u));
                    ax = readlane_f(ab.x, 0);
                    ay = readlane_f(ab.y, 0);
                    bx = readlane_f(ab.x, 1);
                    by = readlane_f(ab.y, 1);
                }
                const float sqab = sqdist(make_float2(ax, ay), make_float2(bx, by));
                uint32_t *hl_mine = queues + ((uint32_t)wave * 2u + c.par) * kQCap;
                const uint32_t t0 = c.j0 >> 6;
                TL_STAMP3(0);
                if ((uint32_t)wk < kLead && t0 + (uint32_t)wk <= last_tile)
                    dense_tile<PRUNE, false>(P, n, (uint32_t)wave, (t0 + (uint32_t)wk) << 6, c.j0, ax, ay, bx, by, sqab, hl_mine, keyslot, lane, tc, &my_hits);
                TL_STAMP3(1);
                TL_SYNC();  // B1
                TL_STAMP3(2);
                // A wave stops only at a hit in a column BEFORE its tile (a lead-round hit stops everybody at once).  A hit that
                // another wave has just posted further right must not stop it: this tile may hold an earlier one.
                // (dense_tile reads the key beside the tile's points and looks at it after the tile's first mask: one LDS round trip
                //  per tile instead of two)
                for (uint32_t t = t0 + kLead + (uint32_t)wk; t <= last_tile; t += (uint32_t)NWK)
                    if (dense_tile<PRUNE, true>(P, n, (uint32_t)wave, t << 6, c.j0, ax, ay, bx, by, sqab, hl_mine, keyslot, lane, tc, &my_hits)) break;
                TL_STAMP3(3);
            } else {
                // ---- pruned step: worker w owns rows w, w+NWK, ... of the block.  L0 (lanes = tiles) yields the live-tile
                // mask of the row in SGPRs, L1 (lanes = j) runs on those tiles right away — no exchange between waves,
                // and every row has its diagonal tile live, so the rows are naturally balanced.
                if (c.dirty_lo <= c.dirty_hi) {  // L0 metadata is rebuilt lazily: only tiles touched by reversals since the last pruned step
                    for (uint32_t t = c.dirty_lo + (uint32_t)wave; t <= c.dirty_hi; t += (uint32_t)NW) build_tile_meta(P, n, t, lane, tbox, tmsq);
                    c.dirty_lo = 0xFFFFFFFFu;
                    c.dirty_hi = 0;
                    TL_SYNC();
                }
                // lane-resident row table: lane l holds P[i0+l] and sq(P[i0+l], P[i0+l+1])
#ifdef TL_PROFILE4
                const uint64_t tp0 = __builtin_amdgcn_s_memtime();
#endif
                const uint32_t i0 = c.i0, j0 = c.j0;
                const float2 rp = pt_get(P, i0 + (uint32_t)lane);
                const float2 rq = pt_get(P, i0 + (uint32_t)lane + 1u);
                const float rowsq = sqdist(rp, rq);
                float4 box[kMaxGroups];
                float msq[kMaxGroups];
#pragma unroll
                for (int gI = 0; gI < kMaxGroups; ++gI) {
                    if (gI < G) {
                        box[gI] = tbox[((uint32_t)gI << 6) + (uint32_t)lane];
                        msq[gI] = tmsq[((uint32_t)gI << 6) + (uint32_t)lane];
                    }
                }
                for (int r = wk; r < (int)R; r += NWK) {
                    const uint32_t i = i0 + (uint32_t)r;
#ifdef TL_PROFILE4
                    qp[1] += 1;
#endif
                    if (r >= NWK) {  // a hit in an earlier row makes this one moot
                        const uint32_t kb = (uint32_t)__builtin_amdgcn_readfirstlane((int)*keyslot);
                        if (kb != kNoKey && (kb >> 16) < i) break;
                    }
                    const float rax = readlane_f(rp.x, r), ray = readlane_f(rp.y, r);
                    const float rbx = readlane_f(rp.x, r + 1), rby = readlane_f(rp.y, r + 1);
                    const float rsqab = readlane_f(rowsq, r);
                    const uint32_t jmin = (r == 0) ? j0 : (i + 2u);
                    const uint32_t tmin = jmin >> 6;
                    uint64_t hm = 0;
                    uint32_t t = 0;
                    tc.prow += 1u;
#pragma unroll
                    for (int gI = 0; gI < kMaxGroups; ++gI) {
                        if (gI < G && (((uint32_t)gI + 1u) << 6) > tmin) {  // groups wholly before the row's first column: nothing to test
                            const uint32_t tl = ((uint32_t)gI << 6) + (uint32_t)lane;
                            // live tiles of the group as a lane mask: every ballot is a comparison's own result, the rest is scalar
                            uint64_t m = __builtin_amdgcn_ballot_w64(tl >= tmin) &
                                         (__builtin_amdgcn_ballot_w64(box_lb(rax, ray, box[gI]) < rsqab) | __builtin_amdgcn_ballot_w64(box_lb(rbx, rby, box[gI]) < msq[gI]));
                            tc.l0 += 64u;
                            while (m != 0) {  // later tiles of this row are later columns
                                t = ((uint32_t)gI << 6) + (uint32_t)(__builtin_ffsll((long long)m) - 1);
                                m &= m - 1;
                                tc.ptile += 1u;
#ifdef TL_PROFILE4
                                qp[2] += 1;
#endif
                                if constexpr (NL) {
                                    // a hit posted in an earlier row makes the rest of this row moot: the step ends when its slowest wave
                                    // arrives, and a row of the early sweeps has up to a dozen live tiles (round 4: 101.5 -> 96.5 ms).
                                    // Only where the late phase takes the hit-free sweeps: in a loop that also runs those, the extra LDS
                                    // read per tile costs more than it saves (two descents per CU: 174 -> 201 ms; round 2 measured the same).
                                    const uint32_t kb = (uint32_t)__builtin_amdgcn_readfirstlane((int)*keyslot);
                                    if (kb != kNoKey && (kb >> 16) < i) goto rows_done;
                                }
                                hm = tile_improving_mask<PRUNE>(P, n, t << 6, jmin, rax, ray, rbx, rby, rsqab, lane, tc);
                                if (hm) goto row_hit;
                            }
                        }
                    }
                    continue;  // nothing in this row: rows w+NWK, ... next
                row_hit:
                    // (fenced: the break below runs straight into B2 with no tracked LDS operation of this wave in between, tl_device.h)
                    if (lane == 0) lds_min_u32_fenced(keyslot, (i << 16) | ((t << 6) + (uint32_t)(__builtin_ffsll((long long)hm) - 1)));
                    break;  // rows w+NWK, ... are later rows
                }
            rows_done:;
#ifdef TL_PROFILE4
                qp[0] += __builtin_amdgcn_s_memtime() - tp0;
                qp[3] += 1;
#endif
            }
#ifdef TL_PROFILE4
            const uint64_t tpb = __builtin_amdgcn_s_memtime();
#endif
            TL_SYNC();  // B2
#ifdef TL_PROFILE4
            if (c.pruned) qp[4] += __builtin_amdgcn_s_memtime() - tpb;
#endif
            TL_STAMP3(4);
#ifdef TL_PROFILE3
            if (!c.pruned) {
                const uint32_t key3 = (uint32_t)__builtin_amdgcn_readfirstlane((int)ctl->kr[c.slot].x);
                const int ty = key3 == kNoKey ? 2 : (((key3 >> 16) >> 6) < (c.j0 >> 6) + kLead ? 0 : 1);
                for (int q = 0; q < 5; ++q) q3[ty][q] += d3[q];
                q3[ty][5] += 1;
            } else {
                t3 = __builtin_amdgcn_s_memtime();  // a pruned step's cycles belong to no dense step
            }
#endif
            need_desc = step_boundary<false, NT, kSlots, NL>(c, acct, P, perm, ctl, queues, n, nrows, R, lane, wave, tid, my_hits, bx, by, reload, tmsq);
        }
#ifdef TL_PROFILE4
        if (d == 0 && lane == 0 && (wave == 1 || wave == 8))
            printf("wave %d, pruned steps of the main loop: %lu; scan %lu cycles per step, %lu rows and %lu tile passes per step, B2 wait %lu\n", wave, qp[3], qp[0] / (qp[3] ? qp[3] : 1),
                   qp[1] * 100 / (qp[3] ? qp[3] : 1), qp[2] * 100 / (qp[3] ? qp[3] : 1), qp[4] / (qp[3] ? qp[3] : 1));
#endif
        if constexpr (NL) {
            if (go_late) late_phase<NT>(A, P, perm, tbox, tmsq, ctl, L, d, n, nrows, ntile, G, c.dirty_lo, c.dirty_hi, acct, sweeps, step, status, n_late_steps, tc, lane, wave, tid);
        }
#ifdef TL_PROFILE3
        if (d == 0 && lane == 0 && (wave == 1 || wave == 6))  // a lead wave and a non-lead one
            for (int ty = 0; ty < 3; ++ty)
                printf("wave %d, dense steps %s: %lu steps; per step: to the lead round %lu, lead tile %lu, B1 wait %lu, round 2 %lu, B2 wait %lu\n", wave,
                       ty == 0 ? "with the first hit in the lead tiles" : ty == 1 ? "with the first hit behind them" : "without a hit", q3[ty][5],
                       q3[ty][0] / (q3[ty][5] ? q3[ty][5] : 1), q3[ty][1] / (q3[ty][5] ? q3[ty][5] : 1), q3[ty][2] / (q3[ty][5] ? q3[ty][5] : 1),
                       q3[ty][3] / (q3[ty][5] ? q3[ty][5] : 1), q3[ty][4] / (q3[ty][5] ? q3[ty][5] : 1));
#endif
    }

    // ---------------------------------------------------------------- results
    uint32_t *__restrict__ out = A.out_pos + (size_t)d * n;
    for (uint32_t k = tid; k < n; k += NT) out[k] = perm[k];

    // Solution::from_parts -> tour_length_by_pos (distance_matrix.rs:235-245): sequential f32 sum,
    // closing edge first.  Edge lengths in parallel, the sum by one lane in tour order.
    if constexpr (COUNT) {
        if (lane == 0) {
            atomicAdd(&ctl->cnt[0], (unsigned long long)tc.l0);
            atomicAdd(&ctl->cnt[1], (unsigned long long)tc.l1);
            atomicAdd(&ctl->cnt[2], (unsigned long long)tc.l2);
            atomicAdd(&ctl->cnt[3], (unsigned long long)tc.l3);
            atomicAdd(&ctl->cnt[4], (unsigned long long)tc.prow);
            atomicAdd(&ctl->cnt[5], (unsigned long long)tc.ptile);
        }
    }
    float total = 0.0f;
    if (n >= 2) total = dist(pt_get(P, n - 1), pt_get(P, 0));
    TL_SYNC();
    for (uint32_t base = 0; base + 1 < n; base += NT) {
        const uint32_t k = base + tid;
        scratch[tid] = (k + 1 < n) ? dist(pt_get(P, k), pt_get(P, k + 1)) : 0.0f;
        TL_SYNC();
        if (tid == 0) {
            const uint32_t cnt = (n - 1 - base) < (uint32_t)NT ? (n - 1 - base) : (uint32_t)NT;
            const float4 *s4 = reinterpret_cast<const float4 *>(scratch);
            uint32_t qd = 0;
            for (; qd + 4 <= cnt; qd += 4) {
                const float4 v = s4[qd >> 2];
                total += v.x;
                total += v.y;
                total += v.z;
                total += v.w;
            }
            for (; qd < cnt; ++qd) total += scratch[qd];
        }
        TL_SYNC();
    }
    // `reversed` of the deferred (dense-mode) reversals: per-lane shares of wave 0, summed
    if (wave == 0) {
#pragma unroll
        for (int sft = 32; sft >= 1; sft >>= 1) {
            const uint32_t lo32 = (uint32_t)__shfl_xor((int)(uint32_t)rev_lane, sft), hi32 = (uint32_t)__shfl_xor((int)(uint32_t)(rev_lane >> 32), sft);
            rev_lane += ((uint64_t)hi32 << 32) | lo32;
        }
        reversed += rev_lane;
    }
    if (tid == 0) {
        A.out_cost[d] = total;
        uint64_t *st = A.out_stats + (size_t)d * TL_STATS_STRIDE;
        st[0] = sweeps;
        st[1] = moves;
        st[2] = reversed;
        st[3] = status;
        st[4] = step;
#if !defined(TL_PROFILE2) && !defined(TL_PROFILE3)
        // [5..8] cascade work (counting instantiation only, else 0): L0 tile bounds, candidates into L1, into L2, into L3;
        // [9] shader clocks of the descent
        // (s_memtime), [10] the same interval in constant 100 MHz ticks (s_memrealtime) -> the clock the CU really held
        st[5] = ctl->cnt[0];
        st[6] = ctl->cnt[1];
        st[7] = ctl->cnt[2];
        st[8] = ctl->cnt[3];
        st[11] = ctl->cnt[4];  // pruned mode: rows bounded by L0 and the tile passes they ran
        st[12] = ctl->cnt[5];
        st[9] = __builtin_amdgcn_s_memtime() - ctl->clk0;
        st[10] = __builtin_amdgcn_s_memrealtime() - ctl->rt0;
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
    if (n_pad > (uint32_t)kMaxGroups * 64u * 64u) return ~(size_t)0;  // beyond the tile-table capacity
    const size_t meta = (size_t)kMaxGroups * 64 * 20 + kCtlBytes;
    const size_t lists = (size_t)32 * kQCap * 4;  // chained-hit lists, also the nt floats of cost-sum scratch
    static_assert(32 * kQCap * 4 >= TL_TWO_OPT_NT * 4, "the hit lists double as the cost-sum scratch");
    (void)nt;
    return (size_t)n_pad * 10 + meta + lists;
}

size_t two_opt_ref_fx_lds_bytes(uint32_t n)
{
    const uint32_t n_pad = ((n + 64u + 63u) / 64u) * 64u;
    if (n_pad > (uint32_t)kMaxGroups * 64u * 64u) return ~(size_t)0;
    return (size_t)n_pad * 7 + (size_t)kMaxGroups * 64 * 20 + kCtlBytes + (size_t)32 * kQCap * 4;
}

// The grid form costs a decode per point read, so it is used only where it buys a second descent per CU: a batch with more
// descents than CUs whose tours fit the LDS twice at 7 B per city but not at 10 (7 100 < n <= 10 240 on MI355X).
// ... or a third and fourth: more than two descents per CU where four tours fit at 7 B per city but not at 10 (3 050 < n <= 4 300).
bool two_opt_ref_fx_pays(uint32_t n, uint32_t count, int cus, int lds_budget)
{
    const size_t plain = two_opt_ref_lds_bytes(n, nullptr, TL_TWO_OPT_NT), fx = two_opt_ref_fx_lds_bytes(n);
    if (cus <= 0 || count <= (uint32_t)cus) return false;
    const bool two = 2 * plain > (size_t)lds_budget && 2 * fx <= (size_t)lds_budget && (size_t)n <= (size_t)kFlushSlotsFx * 512;
    const bool four = count > 2u * (uint32_t)cus && 4 * plain > (size_t)lds_budget && 4 * fx <= (size_t)lds_budget &&
                      (size_t)n <= (size_t)kFlushSlotsFx * 256;
    return two || four;
}

__global__ __launch_bounds__(256) void k_fx_encode(const float2 *__restrict__ xy, uint32_t n, double scale, double inv, uint2 *__restrict__ out,
                                                   uint32_t *__restrict__ bad)
{
    const uint32_t c = blockIdx.x * 256u + threadIdx.x;
    if (c >= n) return;
    const float2 p = xy[c];
    const double gx = rint((double)p.x * scale), gy = rint((double)p.y * scale);
    bool ok = gx >= 0.0 && gy >= 0.0 && gx < 1048576.0 && gy < 1048576.0;
    uint2 g = make_uint2(0u, 0u);
    if (ok) {
        const uint32_t kx = (uint32_t)gx, ky = (uint32_t)gy;
        g = make_uint2(kx | ((ky & 0xFFFu) << 20), ky >> 12);
        const float2 q = fx_decode(g.x, g.y, inv);  // the kernel's own decode: bit for bit, or the form is not used
        ok = __float_as_uint(q.x) == __float_as_uint(p.x) && __float_as_uint(q.y) == __float_as_uint(p.y);
    }
    out[c] = g;
    if (!ok) atomicAdd(bad, 1u);
}

hipError_t launch_fx_encode(const float2 *xy, uint32_t n, double scale, uint2 *out, uint32_t *bad, hipStream_t s)
{
    hipLaunchKernelGGL(k_fx_encode, dim3((n + 255u) / 256u), dim3(256), 0, s, xy, n, scale, 1.0 / scale, out, bad);
    return hipGetLastError();
}

template <int NT, bool PRUNE, bool COUNT, bool FX, bool NL = false>
static hipError_t launch_one(const TwoOptBatchArgs &A, uint32_t count, size_t lds, hipStream_t s)
{
    auto kern = k_two_opt_ref_lds<NT, PRUNE, COUNT, FX, NL>;
    hipError_t e = allow_max_lds(reinterpret_cast<const void *>(kern));
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(count), dim3(NT), lds, s, A);
    return hipGetLastError();
}

template <int NT, bool FX>
static hipError_t launch_nt(const TwoOptBatchArgs &B, uint32_t count, size_t lds, bool prune, bool count_work, hipStream_t s)
{
    // the counting instantiation (TL_FLAG_COUNT_WORK) exists for the form bench.py counts — one descent per CU on 16 waves, plain
    // points; the narrower / grid-coordinate forms run uncounted (stats words 5..8 stay 0): 14 instantiations to compile, not 24
    if constexpr (NT == 1024 && !FX) {
        if (prune && B.nl.rec)  // neighbour lists for the late sweeps (the caller has sized `lds` for their state)
            return count_work ? launch_one<NT, true, true, FX, true>(B, count, lds, s) : launch_one<NT, true, false, FX, true>(B, count, lds, s);
    }
    if constexpr ((NT == 512 || NT == 256) && !FX) {  // two / four descents per CU, each with its late phase (uncounted, like every narrow form)
        if (prune && B.nl.rec) return launch_one<NT, true, false, FX, true>(B, count, lds, s);
    }
    if constexpr (NT == TL_TWO_OPT_NT && !FX) {
        if (count_work) return prune ? launch_one<NT, true, true, FX>(B, count, lds, s) : launch_one<NT, false, true, FX>(B, count, lds, s);
    }
    return prune ? launch_one<NT, true, false, FX>(B, count, lds, s) : launch_one<NT, false, false, FX>(B, count, lds, s);
}

// threads per descent: 0 = by the batch (below), else 1024 / 512 / 256 as forced by a tl_create flag
hipError_t launch_two_opt_ref_lds(const TwoOptBatchArgs &A, uint32_t count, bool prune, hipStream_t s, bool count_work, int cus, int lds_budget,
                                  int force_nt)
{
    uint32_t n_pad = 0;
    const size_t lds = two_opt_ref_lds_bytes(A.n, &n_pad, TL_TWO_OPT_NT);
    TwoOptBatchArgs B = A;
    B.n_pad = n_pad;
    if (A.fx_xy && A.fx_inv != 0.0) {  // grid-coordinate form: two descents per CU on 8 waves, or four on 4 (the caller has checked that it fits)
        const size_t fl = two_opt_ref_fx_lds_bytes(A.n);
        if (cus > 0 && count > 2u * (uint32_t)cus && 4 * fl <= (size_t)lds_budget && (size_t)A.n <= (size_t)kFlushSlotsFx * 256)
            return launch_nt<256, true>(B, count, fl, prune, count_work, s);
        if ((size_t)A.n > (size_t)kFlushSlotsFx * 512) return hipErrorInvalidValue;
        return launch_nt<512, true>(B, count, fl, prune, count_work, s);
    }
    const int nt = two_opt_ref_pick_nt(A.n, count, cus, lds_budget, force_nt);
    const size_t nl_lds = two_opt_ref_nl_lds_bytes(A.n);
    // the lists are read by the 16-wave form and — where two tours WITH their late-phase state share a CU — by the 8-wave form
    const bool nl = prune && B.nl.rec && two_opt_ref_nl_form(A.n, count, cus, lds_budget, force_nt);
    if (!nl) B.nl.rec = nullptr;
    if (nt == 256) return launch_nt<256, false>(B, count, nl ? nl_lds : lds, prune, count_work, s);
    if (nt == 512) return launch_nt<512, false>(B, count, nl ? nl_lds : lds, prune, count_work, s);
    return launch_nt<TL_TWO_OPT_NT, false>(B, count, nl ? nl_lds : lds, prune, count_work, s);
}

// threads per descent of a batch: 1024 where a descent has a CU to itself; where the batch exceeds the CUs and the LDS holds two or
// four tours, 512 / 256 (2 / 4 descents per CU); a flush holds kFlushSlots elements per thread, so a narrow form also needs n <= 15 NT
int two_opt_ref_pick_nt(uint32_t n, uint32_t count, int cus, int lds_budget, int force_nt)
{
    const size_t lds = two_opt_ref_lds_bytes(n, nullptr, TL_TWO_OPT_NT);
    const size_t fit = lds ? (size_t)lds_budget / lds : 1;
    int nt = TL_TWO_OPT_NT;
    if (force_nt) {
        nt = force_nt;
    } else if (cus > 0 && count > (uint32_t)cus) {
        if (fit >= 4 && count > 2u * (uint32_t)cus) nt = 256;  // (up to two per CU the 8-wave form is the faster one: scripts/per_cu_threshold.py)
        else if (fit >= 2) nt = 512;
    }
    while (nt < TL_TWO_OPT_NT && (size_t)n > (size_t)kFlushSlots * (size_t)nt) nt *= 2;
    return nt;
}

// does that form read neighbour lists?  16 waves: wherever the late-phase state fits beside the tour; 8 / 4 waves: where two / four
// such descents still share a CU (a forced narrow form: wherever one fits)
bool two_opt_ref_nl_form(uint32_t n, uint32_t count, int cus, int lds_budget, int force_nt)
{
    if (TL_TWO_OPT_NT != 1024 || n <= (uint32_t)kNlKB + 1u || n > 65535u) return false;
    const size_t nl_lds = two_opt_ref_nl_lds_bytes(n);
    if (nl_lds > (size_t)lds_budget) return false;
    const int nt = two_opt_ref_pick_nt(n, count, cus, lds_budget, force_nt);
    if (nt == 1024) return true;
    if (nt == 512) return force_nt == 512 || 2 * nl_lds <= (size_t)lds_budget;
    if (nt == 256) return force_nt == 256 || 4 * nl_lds <= (size_t)lds_budget;
    return false;
}

size_t two_opt_ref_nl_lds_bytes(uint32_t n)
{
    uint32_t n_pad = 0;
    const size_t base = two_opt_ref_lds_bytes(n, &n_pad, TL_TWO_OPT_NT);
    if (base == ~(size_t)0) return base;
    return base + (size_t)n_pad * 2 + (size_t)kNlLongCap * 16 + (size_t)16 * 64 * 2;
}

bool two_opt_ref_nl_applies(uint32_t n, uint32_t count, int cus, int lds_budget, int force_nt)
{
    return two_opt_ref_nl_form(n, count, cus, lds_budget, force_nt);
}

}  // namespace tl
