// lk.hip — Lin–Kernighan (reference: src/tsp/lin_kernighan.rs).  (Candidate lists: kdtree.hip; NN seed and brute-force lists:
// nn_knn.hip.)
//
// Compiled twice: as it stands (namespace tl, chains of up to TL_LK_MAX_DEPTH = 6 exchanges held in registers — LKOptions::default()
// is 5, the CLI's --max-depth default too) and through lk_deep.hip with TL_LK_MAX_DEPTH 16 in namespace tl_lk_deep for
// max_depth 7..16 (the reference's max_depth is an unbounded usize, mod.rs:1252; recursion lin_kernighan.rs:265-340).
//
//   k_lk_solve   lin_kernighan::solve (lin_kernighan.rs:35-100) as ONE persistent workgroup: lk_pass
//                (:454-481) scans the (t1, orientation) pairs in the reference's order 1024 at a time — every
//                lane runs the depth-limited sequential chain search find_lk_chain (:265-340) for its pair,
//                checks the closed chain for a single Hamiltonian cycle, and the lowest pair index with a
//                valid chain wins (ds_min) — then the group applies the chain (apply_lk_chain, :397-450) by
//                copying tour arcs, and rescans from the start exactly like find_lk_move does.  The ILS loop
//                (double_bridge kicks :485-499, accept-if-shorter, plateau stop :75-97) runs on-device too.
//
// The reference checks chain validity and applies chains by rebuilding adjacency lists and tracing n cities
// (:181-250, :397-450).  Here both are O(k): the k removed edges cut the tour into k arcs; walking arcs through
// the k added edges visits all k arcs iff the result is one cycle (the new graph is 2-regular), and the new flat
// tour is those arcs copied in walk order starting at tour[0] — forward if the edge (tour[0],tour[1]) survives,
// else backward, which is what the reference's trace from `adj[start][0]` does.
#include "tl_kernels.h"
#include <type_traits>

#pragma clang fp contract(off)

#ifndef TL_LK_MAX_DEPTH
#define TL_LK_MAX_DEPTH 6
#endif
#ifndef TL_LK_NS
#define TL_LK_NS tl
#endif

namespace TL_LK_NS {
using namespace tl;  // (a no-op for the default namespace)

namespace {

constexpr float kLkEps = 1e-6f;  // lin_kernighan.rs:252
// moves after which an lk_pass is taken to be cycling (the reference's loop does not end on such an input: lin_kernighan.rs:468-478
// goes on while find_lk_move returns a chain, and a chain's "gain" is a sum of rounded f32 terms).  A pass from a random tour takes
// about n moves.
__host__ __device__ __forceinline__ uint64_t lk_pass_cap(uint32_t n) { return 64ull * n + 4096ull; }
constexpr int kLkNT = 1024;
#ifndef TL_LK_WINDOW_MARGIN
#define TL_LK_WINDOW_MARGIN 2048
#endif
constexpr uint32_t kLkWindowMargin = TL_LK_WINDOW_MARGIN;  // pairs scanned beyond the previous hit before the whole pass is looked at
constexpr uint32_t kLkRebuildSplitN = 1500;                // from this size on the post-move rebuild is its own chip-wide kernel
constexpr uint32_t kLkWindowFirst = 4096;                  // prefix of a fresh pass

// ---------------------------------------------------------------------------------------------- LK
template <typename NextT>
struct LkViewT {
    const float2 *xy;      // global, or an LDS copy (k_lk_scan)
    const uint32_t *cand;
    const NextT *next;     // successor in the current tour; u16 LDS copy in k_lk_scan
    uint32_t k;
    uint32_t max_depth;
};
using LkView = LkViewT<uint32_t>;

// The packed view (round 5): what a branch of the chain search needs of a candidate and of a tour successor sits
// NEXT TO the index it follows from — candd[t][q] = (candidate id, d(t, candidate)), nx[c] = (next[c], x, y of next[c], d(c, next[c])) —
// so that a branch costs two dependent look-ups (candd -> nx) instead of four (cand -> xy -> next -> xy) and one square root (the
// closing edge) instead of three.  The two distances are the same f32 values the classic view computes (same operands, same order).
struct LkViewPk {
    const uint2 *candd;   // [n][k]: candidate id, bits of d(t, candidate)
    const float4 *nx;     // [n]: bits of next[c], next[c]'s x, y, d(c, next[c]) of the CURRENT tour (kept by the step kernel)
    uint32_t k;
    uint32_t max_depth;
};
// candidate q of t_open: its id, d(t_open, id), and next[t_open] (for is_tour_edge)
template <typename NextT>
__device__ __forceinline__ void lk_probe(const LkViewT<NextT> &V, uint32_t t_open, float2 p_open, uint32_t q, uint32_t &t_next, float2 &p_next, float &d1)
{
    t_next = V.cand[(size_t)t_open * V.k + q];
    p_next = V.xy[t_next];
    d1 = dist(p_open, p_next);
}
__device__ __forceinline__ void lk_probe(const LkViewPk &V, uint32_t t_open, float2, uint32_t q, uint32_t &t_next, float2 &p_next, float &d1)
{
    const uint2 c = V.candd[(size_t)t_open * V.k + q];
    t_next = c.x;
    p_next = make_float2(0.0f, 0.0f);  // (not needed: both of its distances come ready)
    d1 = __uint_as_float(c.y);
}
template <typename NextT>
__device__ __forceinline__ uint32_t lk_next_of(const LkViewT<NextT> &V, uint32_t c) { return (uint32_t)V.next[c]; }
__device__ __forceinline__ uint32_t lk_next_of(const LkViewPk &V, uint32_t c) { return __float_as_uint(V.nx[c].x); }
// regime A: t_break = next[t_next], its point, d(t_next, t_break)
template <typename NextT>
__device__ __forceinline__ uint32_t lk_break_id(const LkViewT<NextT> &V, uint32_t t_next, float4 &rec)
{
    (void)rec;
    return (uint32_t)V.next[t_next];
}
__device__ __forceinline__ uint32_t lk_break_id(const LkViewPk &V, uint32_t t_next, float4 &rec)
{
    rec = V.nx[t_next];
    return __float_as_uint(rec.x);
}
template <typename NextT>
__device__ __forceinline__ void lk_break_pt(const LkViewT<NextT> &V, uint32_t t_break, float2 p_next, const float4 &, float2 &p_break, float &d2)
{
    p_break = V.xy[t_break];
    d2 = dist(p_next, p_break);
}
__device__ __forceinline__ void lk_break_pt(const LkViewPk &, uint32_t, float2, const float4 &rec, float2 &p_break, float &d2)
{
    p_break = make_float2(rec.y, rec.z);
    d2 = rec.w;
}

__host__ __device__ __forceinline__ uint32_t lk_subs(const LkArgs &G)  // sub-searches per (t1, orientation) pair
{
    return G.k * (G.k + 1u) * (G.split_levels == 3u ? G.k + 1u : 1u);
}

constexpr int kLkMaxDepth = TL_LK_MAX_DEPTH;         // compile-time recursion bound (6: default max_depth = 5; 16: the deep build, lk_deep.hip)
constexpr int kLkMaxChain = 2 * kLkMaxDepth + 2;
// u32 words per pair in LkArgs::chains: length, the chain's cities, (chip-wide step) their tour positions, then the segment count and
// the move's segment table (src, len, dst, dir per segment: the scan's validating lane builds it, round 4) — 96 at depth 6
constexpr uint32_t kLkMaxSeg = 2u * ((uint32_t)kLkMaxDepth + 2u);
constexpr uint32_t kLkSlotSeg = 1u + 2u * (uint32_t)kLkMaxChain;  // word of the segment count; the table follows
constexpr uint32_t kLkSlot = ((kLkSlotSeg + 1u + 4u * kLkMaxSeg + 31u) / 32u) * 32u;
constexpr uint32_t kLkTailCap = 48, kLkTailWords = 10;  // parked depth-3 walks of one scan workgroup (k_lk_scan_sub)
// u32 words per kept sub-search chain (len + kLkMaxChain cities, 64-byte slots): 16 at depth 6
constexpr uint32_t kLkSubSlot = ((1u + (uint32_t)kLkMaxChain + 15u) / 16u) * 16u;
static_assert(kLkMaxChain + 1 <= (int)kLkSubSlot && 1 + 2 * kLkMaxChain <= (int)kLkSlot, "chain slots");
static_assert(TL_LK_MAX_DEPTH != 6 || (kLkSlot == 96u && kLkSubSlot == 16u), "the default build's layout");

template <int LEN>
__device__ __forceinline__ bool in_chain(const uint32_t (&chain)[kLkMaxChain], uint32_t x)
{
    bool f = false;
#pragma unroll
    for (int t = 0; t < LEN; ++t) f |= (chain[t] == x);
    return f;
}

// lin_kernighan.rs:265-340 find_lk_chain; DEPTH is the reference's `depth`, chain holds 2*DEPTH+2 cities on entry.
template <int DEPTH, typename VT>
__device__ bool lk_chain(const VT &V, uint32_t (&chain)[kLkMaxChain], uint32_t &clen, const float2 p1,
                         const uint32_t t_open, const float2 p_open, const float gain)
{
    if (DEPTH >= 1) {
        const float close_gain = gain - dist(p_open, p1);  // :280-283
        if (close_gain > kLkEps) {
            clen = 2 * DEPTH + 2;
            return true;
        }
    }
    if ((uint32_t)DEPTH >= V.max_depth) return false;  // :286-288
    if constexpr (DEPTH < kLkMaxDepth) {
        for (uint32_t q = 0; q < V.k; ++q) {
            uint32_t t_next;
            float2 p_next, p_break;
            float d1, d2;
            float4 rec;
            lk_probe(V, t_open, p_open, q, t_next, p_next, d1);
            const float g1 = gain - d1;
            if (g1 <= kLkEps) break;                                   // :292-295 candidates are sorted
            if (in_chain<2 * DEPTH + 2>(chain, t_next)) continue;      // used[t_next]
            const uint32_t t_break = lk_break_id(V, t_next, rec);      // :305 regime A
            if (lk_next_of(V, t_open) == t_next || t_break == t_open) continue;  // is_tour_edge: prev[a]==b <=> next[b]==a
            if (in_chain<2 * DEPTH + 2>(chain, t_break)) continue;     // used[t_break]
            lk_break_pt(V, t_break, p_next, rec, p_break, d2);
            const float g2 = g1 + d2;
            chain[2 * DEPTH + 2] = t_next;
            chain[2 * DEPTH + 3] = t_break;
            if (lk_chain<DEPTH + 1, VT>(V, chain, clen, p1, t_break, p_break, g2)) return true;
        }
    }
    return false;
}

// Arc model of a closed chain c[0..2k): removed edges (c[2i],c[2i+1]), added edges (c[2i+1],c[2i+2]) and (c[2k-1],c[0]).
struct Arcs {
    uint32_t lo[kLkMaxDepth + 1];  // sorted: removed edge m joins positions lo[m] and lo[m]+1 (mod n)
    uint32_t k;
};

// Tour positions of the chain's cities, all loads independent (one L2 latency).  Every arc boundary of the arc model is the
// position of a chain city, so the validity walk and the application need no further pos[] / tour[] look-ups: the city
// at a boundary position is found among these (index_at), its added-edge partner by index (partner_index).
template <typename PosT>
__device__ __forceinline__ void chain_positions(const uint32_t (&chain)[kLkMaxChain], uint32_t clen, const PosT *pos,
                                                uint32_t (&cpos)[kLkMaxChain])
{
#pragma unroll
    for (int t = 0; t < kLkMaxChain; ++t) cpos[t] = (uint32_t)t < clen ? pos[chain[t]] : 0xFFFFFFFFu;
}
__device__ __forceinline__ uint32_t sel_at(const uint32_t (&a)[kLkMaxChain], uint32_t idx)
{
    uint32_t r = 0xFFFFFFFFu;
#pragma unroll
    for (int t = 0; t < kLkMaxChain; ++t) r = ((uint32_t)t == idx) ? a[t] : r;
    return r;
}
__device__ __forceinline__ uint32_t index_at(const uint32_t (&cpos)[kLkMaxChain], uint32_t clen, uint32_t p)
{
    uint32_t idx = 0xFFFFFFFFu;
#pragma unroll
    for (int t = kLkMaxChain - 1; t >= 0; --t) idx = ((uint32_t)t < clen && cpos[t] == p) ? (uint32_t)t : idx;
    return idx;
}
__device__ __forceinline__ uint32_t partner_index(uint32_t t, uint32_t clen)  // the city joined to chain[t] by an added edge
{
    if (t == 0u) return clen - 1u;
    if (t == clen - 1u) return 0u;
    return (t & 1u) ? t + 1u : t - 1u;
}

__device__ __forceinline__ void arcs_build(const uint32_t (&cpos)[kLkMaxChain], uint32_t clen, uint32_t n, Arcs &A)
{
    A.k = clen / 2;
#pragma unroll
    for (int m = 0; m < kLkMaxDepth + 1; ++m) {
        if ((uint32_t)m >= A.k) break;
        const uint32_t pu = cpos[2 * m], pv = cpos[2 * m + 1];
        // adjacent positions: the edge sits after the one whose successor is the other
        A.lo[m] = ((pu + 1u == pv) || (pu == n - 1u && pv == 0u)) ? pu : pv;
    }
    for (uint32_t a = 1; a < A.k; ++a) {  // insertion sort, k <= 7
        const uint32_t v = A.lo[a];
        uint32_t b = a;
        while (b > 0 && A.lo[b - 1] > v) { A.lo[b] = A.lo[b - 1]; --b; }
        A.lo[b] = v;
    }
}

// endpoint city at position p: is its removed edge behind it (p is an arc START) and where is the arc's other end
__device__ __forceinline__ bool arc_is_start(const Arcs &A, uint32_t p, uint32_t n)
{
    const uint32_t before = p == 0u ? n - 1u : p - 1u;
    for (uint32_t m = 0; m < A.k; ++m)
        if (A.lo[m] == before) return true;
    return false;
}
__device__ __forceinline__ uint32_t arc_end_from_start(const Arcs &A, uint32_t p)
{
    // smallest lo >= p, else (wrap) the smallest lo
    for (uint32_t m = 0; m < A.k; ++m)
        if (A.lo[m] >= p) return A.lo[m];
    return A.lo[0];
}
__device__ __forceinline__ uint32_t arc_start_from_end(const Arcs &A, uint32_t p, uint32_t n)
{
    // largest lo < p, +1; else (wrap) largest lo + 1
    uint32_t best = 0xFFFFFFFFu;
    for (uint32_t m = 0; m < A.k; ++m)
        if (A.lo[m] < p) best = A.lo[m];
    if (best == 0xFFFFFFFFu) best = A.lo[A.k - 1];
    return best + 1u == n ? 0u : best + 1u;
}
// chain_is_valid_tour (lin_kernighan.rs:181-250): one Hamiltonian cycle <=> the arc walk sees all k arcs
// (cpos: the chain's tour positions — chain_positions — which the caller may want to keep for lk_build_segments)
__device__ bool chain_valid_at(const uint32_t (&cpos)[kLkMaxChain], uint32_t clen, uint32_t n)
{
    if (clen < 4) return false;
    Arcs A;
    arcs_build(cpos, clen, n, A);
    uint32_t ti = 0, arcs = 0;  // chain cities are distinct: back at index 0 <=> back at chain[0]
    do {
        const uint32_t p = sel_at(cpos, ti);
        const uint32_t other = arc_is_start(A, p, n) ? arc_end_from_start(A, p) : arc_start_from_end(A, p, n);
        ++arcs;
        if (arcs > A.k) return false;
        const uint32_t tj = index_at(cpos, clen, other);  // the city at an arc boundary is a chain city
        if (tj == 0xFFFFFFFFu) return false;
        ti = partner_index(tj, clen);
    } while (ti != 0u);
    return arcs == A.k;
}
__device__ bool chain_valid(const uint32_t (&chain)[kLkMaxChain], uint32_t clen, const uint32_t *tour, const uint32_t *pos, uint32_t n)
{
    (void)tour;
    if (clen < 4) return false;
    uint32_t cpos[kLkMaxChain];
    chain_positions(chain, clen, pos, cpos);
    return chain_valid_at(cpos, clen, n);
}

struct LkSeg {
    uint32_t src, len, dst;
    int dir;
};

// apply_lk_chain (lin_kernighan.rs:397-450) as a table: the new flat tour is the old tour's arcs copied in walk order starting at
// tour[0] — forward if the edge (tour[0], tour[1]) survives, else backward.  cpos: tour positions of the chain's cities.
__device__ __forceinline__ uint32_t lk_build_segments(const uint32_t (&cpos)[kLkMaxChain], uint32_t clen, uint32_t n, LkSeg *seg)
{
    Arcs A;
    arcs_build(cpos, clen, n, A);
    bool first_removed = false;
    for (uint32_t m = 0; m < A.k; ++m) first_removed |= (A.lo[m] == 0u);
    uint32_t nseg = 0, emitted = 0, p = 0;
    int dir = first_removed ? -1 : +1;
    while (emitted < n && nseg < kLkMaxSeg) {
        const uint32_t endp = dir > 0 ? arc_end_from_start(A, p) : arc_start_from_end(A, p, n);
        uint32_t len = dir > 0 ? (endp >= p ? endp - p + 1u : endp + n - p + 1u) : (p >= endp ? p - endp + 1u : p + n - endp + 1u);
        if (len > n - emitted) len = n - emitted;
        seg[nseg].src = p;
        seg[nseg].len = len;
        seg[nseg].dst = emitted;
        seg[nseg].dir = dir;
        ++nseg;
        emitted += len;
        if (emitted >= n) break;
        p = sel_at(cpos, partner_index(index_at(cpos, clen, endp), clen));  // position of the boundary city's added-edge partner
        dir = arc_is_start(A, p, n) ? +1 : -1;
    }
    return nseg;
}

// What the chip-wide step needs of a pair's valid chain, filed in the pair's slot by the scan lane that validated it (a call, not
// inlined: the scan kernel lives on 64 VGPRs, and this runs for the few pairs of a scan that hold a valid chain): the chain's tour
// positions (the step kernel rebuilds pos[] while its other workgroups still read it) and the move's segment table — which used
// to be ONE thread's serial prologue of every step workgroup (k_lk_control<true>: 8.5 us per round, of which ~3 were this).
__device__ __attribute__((noinline)) void lk_file_move(uint32_t *slot, uint32_t clen, const uint32_t *pos, uint32_t n)
{
    uint32_t cpos[kLkMaxChain];
#pragma unroll
    for (int t = 0; t < kLkMaxChain; ++t) cpos[t] = (uint32_t)t < clen ? pos[slot[1 + t]] : 0xFFFFFFFFu;
#pragma unroll
    for (int t = 0; t < kLkMaxChain; ++t) slot[1 + kLkMaxChain + t] = cpos[t];
    LkSeg seg[kLkMaxSeg];
    const uint32_t nseg = lk_build_segments(cpos, clen, n, seg);
    slot[kLkSlotSeg] = nseg;
    for (uint32_t q = 0; q < nseg; ++q) {
        slot[kLkSlotSeg + 1u + 4u * q] = seg[q].src;
        slot[kLkSlotSeg + 2u + 4u * q] = seg[q].len;
        slot[kLkSlotSeg + 3u + 4u * q] = seg[q].dst;
        slot[kLkSlotSeg + 4u + 4u * q] = (uint32_t)seg[q].dir;
    }
}

}  // namespace

// The whole ILS in ONE persistent workgroup of NT threads.  SMALL: every array the search touches — coordinates, candidate
// lists, tour, scratch tour, rank, successor, predecessor, scan order, best tour — lives in this CU's LDS (n (36 + 4k)
// bytes), so a node of the chain search costs LDS latencies and a move costs no kernel boundary.  Built as the candidate
// for small instances and MEASURED SLOWER than the chip-wide scans at every size: berlin52 with the
// CLI's options (3 327 scans) 129 ms against 80 ms, a280 517 against 349 ms, n = 1 500 933 against 58 ms
// (scripts/timing_lk.py, 64 / 256 / 1024 threads) — what a scan costs here is the single-lane stretches (segment table of
// apply_lk_chain, the ordered cost sum), not memory latency.  Kept as TL_FLAG_LK_SMALL, a cross-check of the state machine.
// !SMALL: the same code on the global arrays (TL_FLAG_LK_ONE_WORKGROUP).
template <int NT, bool SMALL>
__global__ __launch_bounds__(NT) void k_lk_solve(LkArgs G)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lk_small_smem[];
    __shared__ uint32_t s_key, s_clen, s_nseg, s_flag;
    __shared__ uint32_t s_chain[kLkMaxChain];
    __shared__ LkSeg s_seg[2 * (kLkMaxDepth + 2)];
    __shared__ float s_part[NT];
    constexpr uint32_t kLkNT = NT;  // shadows the file-level constant inside this kernel
    const uint32_t tid = threadIdx.x, n = G.n;
    uint32_t *tour = G.tour, *alt = G.alt, *pos = G.pos, *next = G.next, *prev = G.prev;
    uint32_t *city_ids = G.city_ids, *best = G.best;
    const float2 *xy = G.xy;
    const uint32_t *cand = G.cand;
    if (SMALL) {
        float2 *lxy = reinterpret_cast<float2 *>(lk_small_smem);
        uint32_t *w = reinterpret_cast<uint32_t *>(lk_small_smem + (size_t)n * 8);
        uint32_t *lcand = w;
        w += (size_t)n * G.k;
        tour = w; alt = w + n; pos = w + 2 * (size_t)n; next = w + 3 * (size_t)n; prev = w + 4 * (size_t)n;
        city_ids = w + 5 * (size_t)n; best = w + 6 * (size_t)n;
        for (uint32_t r = tid; r < n; r += kLkNT) {
            lxy[r] = G.xy[r];
            tour[r] = G.tour[r];
        }
        for (size_t e = tid; e < (size_t)n * G.k; e += kLkNT) lcand[e] = G.cand[e];
        xy = lxy;
        cand = lcand;
        TL_SYNC();
    }
    uint64_t scans = 0, searches = 0, moves = 0, exchanged = 0;

    auto rebuild = [&](const uint32_t *t) {  // make_pos + flat_to_next_prev (:110-145)
        for (uint32_t r = tid; r < n; r += kLkNT) {
            const uint32_t c = t[r];
            pos[c] = r;
            next[c] = t[r + 1u == n ? 0u : r + 1u];
            prev[c] = t[r == 0u ? n - 1u : r - 1u];
        }
        TL_SYNC();
    };

    // tour_distance (:118-122): (0..n).map(d(t[i], t[(i+1)%n])).sum() — sequential f32 from 0, closing edge last
    auto tour_distance = [&](const uint32_t *t) -> float {
        float total = 0.0f;
        for (uint32_t base = 0; base < n; base += kLkNT) {
            const uint32_t r = base + tid;
            if (r < n) {
                const uint32_t a = t[r], b = t[r + 1u == n ? 0u : r + 1u];
                s_part[tid] = a == b ? 0.0f : dist(xy[a], xy[b]);
            }
            TL_SYNC();
            if (tid == 0) {
                const uint32_t cnt = (n - base) < (uint32_t)kLkNT ? (n - base) : (uint32_t)kLkNT;
                for (uint32_t q = 0; q < cnt; ++q) total += s_part[q];
            }
            TL_SYNC();
        }
        if (tid == 0) s_part[0] = total;
        TL_SYNC();
        const float r = s_part[0];
        TL_SYNC();
        return r;
    };

    // lk_pass (:454-481) on `tour` (which must be G.tour); returns nothing, leaves the improved tour in G.tour
    auto lk_pass = [&]() {
        for (uint32_t r = tid; r < n; r += kLkNT) city_ids[r] = tour[r];  // :466 fixed for the whole pass
        TL_SYNC();
        while (true) {
            rebuild(tour);  // :470
            LkView V{xy, cand, next, G.k, G.max_depth};
            bool found_any = false;
            for (uint32_t base = 0; base < 2u * n; base += kLkNT) {  // find_lk_move (:345-389) in its own order
                if (tid == 0) s_key = 0xFFFFFFFFu;
                TL_SYNC();
                const uint32_t idx = base + tid;
                uint32_t chain[kLkMaxChain];
                uint32_t clen = 0;
                bool ok = false;
                if (idx < 2u * n) {
                    const uint32_t t1 = city_ids[idx >> 1];
                    const uint32_t t2 = (idx & 1u) ? prev[t1] : next[t1];  // :359 [next[t1], prev[t1]]
                    const float2 p1 = xy[t1], p2 = xy[t2];
                    chain[0] = t1;
                    chain[1] = t2;
                    const float g0 = t1 == t2 ? 0.0f : dist(p1, p2);
                    if (lk_chain<0, LkView>(V, chain, clen, p1, t2, p2, g0)) ok = chain_valid(chain, clen, tour, pos, n);
                }
                if (ok) atomicMin(&s_key, idx);
                TL_SYNC();
                const uint32_t key = s_key;
                if (key != 0xFFFFFFFFu) {
                    if (idx == key) {
                        s_clen = clen;
                        for (uint32_t t = 0; t < clen; ++t) s_chain[t] = chain[t];
                    }
                    searches += (uint64_t)key + 1u;
                    found_any = true;
                    TL_SYNC();
                    break;
                }
                TL_SYNC();
            }
            ++scans;
            if (!found_any) {
                searches += 2ull * n;
                break;
            }
            // ---- apply_lk_chain (:397-450): arcs copied in walk order from tour[0]
            if (tid == 0) {
                uint32_t chain[kLkMaxChain];
                const uint32_t clen = s_clen;
                for (uint32_t t = 0; t < clen; ++t) chain[t] = s_chain[t];
                uint32_t cpos[kLkMaxChain];
                chain_positions(chain, clen, pos, cpos);
                Arcs A;
                arcs_build(cpos, clen, n, A);
                bool first_removed = false;  // is the edge (tour[0], tour[1]) one of the removed ones?
                for (uint32_t m = 0; m < A.k; ++m) first_removed |= (A.lo[m] == 0u);
                uint32_t nseg = 0, emitted = 0;
                uint32_t p = 0;
                int dir = first_removed ? -1 : +1;
                while (emitted < n && nseg < 2 * (kLkMaxDepth + 2)) {
                    // run from p in direction dir to the end of the arc
                    uint32_t endp;
                    if (dir > 0) endp = arc_end_from_start(A, p);   // first removed edge at or after p
                    else endp = arc_start_from_end(A, p, n);        // first position after the previous removed edge
                    // length of the run p -> endp in direction dir, cyclic
                    uint32_t len = dir > 0 ? (endp >= p ? endp - p + 1u : endp + n - p + 1u)
                                           : (p >= endp ? p - endp + 1u : p + n - endp + 1u);
                    if (len > n - emitted) len = n - emitted;  // the arc we started inside is finished at the very end
                    s_seg[nseg].src = p;
                    s_seg[nseg].len = len;
                    s_seg[nseg].dst = emitted;
                    s_seg[nseg].dir = dir;
                    ++nseg;
                    emitted += len;
                    if (emitted >= n) break;
                    p = sel_at(cpos, partner_index(index_at(cpos, clen, endp), clen));  // position of the boundary city's added-edge partner
                    dir = arc_is_start(A, p, n) ? +1 : -1;
                }
                s_nseg = nseg;
            }
            TL_SYNC();
            const uint32_t nseg = s_nseg;
            for (uint32_t sidx = 0; sidx < nseg; ++sidx) {
                const LkSeg sg = s_seg[sidx];
                for (uint32_t t = tid; t < sg.len; t += kLkNT) {
                    uint32_t sp = sg.dir > 0 ? sg.src + t : sg.src + n - t;
                    if (sp >= n) sp -= n;
                    alt[sg.dst + t] = tour[sp];
                }
            }
            TL_SYNC();
            for (uint32_t r = tid; r < n; r += kLkNT) tour[r] = alt[r];
            ++moves;
            exchanged += (uint64_t)(s_clen / 2u);
            TL_SYNC();
        }
    };

    // ---- lin_kernighan::solve (:35-100)
    if (n >= 4) {
        lk_pass();                                                   // :61-68
        for (uint32_t r = tid; r < n; r += kLkNT) best[r] = tour[r];
        TL_SYNC();
        float best_dist = tour_distance(best);                       // :70
        uint32_t platoo = 0;
        uint64_t draws = 0;
        for (uint32_t e = 0; e < G.epochs; ++e) {                    // :75
            // double_bridge (:485-499) with seeded draws r = splitmix64 % (n/4)
            if (n < 8) {
                for (uint32_t r = tid; r < n; r += kLkNT) tour[r] = best[r];
            } else {
                const uint32_t qn = n / 4u;
                const uint32_t r1 = (uint32_t)(splitmix64_at(G.seed, draws) % qn), r2 = (uint32_t)(splitmix64_at(G.seed, draws + 1) % qn),
                               r3 = (uint32_t)(splitmix64_at(G.seed, draws + 2) % qn);
                draws += 3;
                const uint32_t p1 = 1u + r1, p2 = p1 + 1u + r2, p3 = p2 + 1u + r3;
                for (uint32_t w = tid; w < n; w += kLkNT) {
                    uint32_t src;
                    if (w < p1) src = w;                                  // tour[0..p1]
                    else if (w < p1 + (p3 - p2)) src = p2 + (w - p1);     // tour[p2..p3]
                    else if (w < p3) src = p1 + (w - p1 - (p3 - p2));     // tour[p1..p2]
                    else src = w;                                         // tour[p3..n]
                    tour[w] = best[src];
                }
            }
            TL_SYNC();
            lk_pass();
            const float dcur = tour_distance(tour);
            if (dcur < best_dist) {                                  // :86
                for (uint32_t r = tid; r < n; r += kLkNT) best[r] = tour[r];
                TL_SYNC();
                best_dist = dcur;
                platoo = 0;
            } else {
                if (++platoo >= G.platoo_epochs) break;              // :92-95
            }
        }
    } else {
        for (uint32_t r = tid; r < n; r += kLkNT) best[r] = tour[r];
    }
    TL_SYNC();
    if (SMALL)
        for (uint32_t r = tid; r < n; r += kLkNT) G.best[r] = best[r];
    if (tid == 0) {
        G.counters[0] = scans;
        G.counters[1] = searches;
        G.counters[2] = moves;
        G.counters[3] = exchanged;
        (void)s_flag;
    }
}

// ------------------------------------------------------------------------------------------------
// Multi-CU variant for large n: the same algorithm as k_lk_solve, cut at its only grid-wide seam.
//   k_lk_scan     every (t1, orientation) pair of find_lk_move at once, one lane each, over all CUs; lanes with a
//                 valid chain store it in their slot and atomicMin the pair index — the reference's "first in
//                 order" is the minimum.
//   k_lk_control  one workgroup, a device-side state machine: apply the winning chain and rescan, or — when the scan
//                 found nothing, i.e. the lk_pass is over — evaluate the tour, accept/reject, kick (double_bridge)
//                 or finish.  The host only enqueues (scan, control) pairs in batches and polls `finished`.
// ------------------------------------------------------------------------------------------------
template <bool LDS>
__global__ __launch_bounds__(256) void k_lk_scan(LkArgs G)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lk_smem[];
    LkState *S = G.state;
    if (S->finished) return;
    const uint32_t n = G.n, idx = blockIdx.x * 256u + threadIdx.x;
    // The chain search is a divergent, latency-bound walk over xy[] and next[]: stage both in this CU's LDS
    // (8 B + 2 B per city) so a node costs one L2 access (the candidate list) instead of four.
    float2 *xyL = reinterpret_cast<float2 *>(lk_smem);
    uint16_t *nextL = reinterpret_cast<uint16_t *>(lk_smem + (size_t)n * 8);
    if (LDS) {
        for (uint32_t c = threadIdx.x; c < n; c += 256u) {
            xyL[c] = G.xy[c];
            nextL[c] = (uint16_t)G.next[c];
        }
        TL_SYNC();
    }
    if (idx >= 2u * n || idx >= S->window) return;
    uint32_t chain[kLkMaxChain];
    uint32_t clen = 0;
    const uint32_t t1 = G.city_ids[idx >> 1];
    const uint32_t t2 = (idx & 1u) ? G.prev[t1] : G.next[t1];
    const float2 p1 = G.xy[t1], p2 = G.xy[t2];
    chain[0] = t1;
    chain[1] = t2;
    const float g0 = t1 == t2 ? 0.0f : dist(p1, p2);
    bool found;
    if (LDS) {
        LkViewT<uint16_t> V{xyL, G.cand, nextL, G.k, G.max_depth};
        found = lk_chain<0, LkViewT<uint16_t>>(V, chain, clen, p1, t2, p2, g0);
    } else {
        LkView V{G.xy, G.cand, G.next, G.k, G.max_depth};
        found = lk_chain<0, LkView>(V, chain, clen, p1, t2, p2, g0);
    }
    if (found && chain_valid(chain, clen, G.tour, G.pos, n)) {
        uint32_t *slot = G.chains + (size_t)idx * kLkSlot;
        slot[0] = clen;
        for (uint32_t t = 0; t < clen; ++t) slot[1 + t] = chain[t];
        atomicMin(&S->key, idx);
    }
}

// ---- split scan: the chain search of one (t1, orientation) pair is itself cut into k*(k+1) independent
// sub-searches — branch q1 at depth 0 and, at depth 1, either the closing test (s = 0) or branch q2 = s-1 — so a
// find_lk_move becomes 2n*k*(k+1) short walks (810 K lanes at n = 13 509, k = 5) instead of 2n long divergent ones.
// Exactness: candidates are sorted by the same f32 distances, so g1 is non-increasing along a list and the
// reference's `break` at g1 <= EPS (:292-295) is implied by each sub-search's own test; `used[]` depends only on the
// chain.  The reference keeps the FIRST chain found in DFS order and drops the pair if that one is invalid (:373-381),
// hence two steps: k_lk_scan_sub records the minimum (q1, s) with a chain per pair, k_lk_scan_pick re-derives that
// chain, checks it, and posts the pair.
template <typename VT>
__device__ bool lk_subsearch(const VT &V, uint32_t (&chain)[kLkMaxChain], uint32_t &clen, const float2 p1,
                             const uint32_t t1, const uint32_t t2, const float2 p2, const float g0, const uint32_t q1, const uint32_t s)
{
    // depth 0 (find_lk_chain :290-337 with depth = 0)
    const uint32_t t3 = V.cand[(size_t)t2 * V.k + q1];
    const float2 p3 = V.xy[t3];
    const float g1 = g0 - dist(p2, p3);
    if (g1 <= kLkEps) return false;
    if (t3 == t1 || t3 == t2) return false;
    const uint32_t t4 = V.next[t3];
    if ((uint32_t)V.next[t2] == t3 || t4 == t2) return false;
    if (t4 == t1 || t4 == t2) return false;
    const float2 p4 = V.xy[t4];
    const float g2 = g1 + dist(p3, p4);
    chain[2] = t3;
    chain[3] = t4;
    // depth 1
    const float close_gain = g2 - dist(p4, p1);
    if (s == 0u) {
        if (close_gain > kLkEps) {
            clen = 4;
            return true;
        }
        return false;
    }
    if (close_gain > kLkEps) return false;  // the sequential search returned at s = 0
    if (1u >= V.max_depth) return false;
    const uint32_t t5 = V.cand[(size_t)t4 * V.k + (s - 1u)];
    const float2 p5 = V.xy[t5];
    const float g3 = g2 - dist(p4, p5);
    if (g3 <= kLkEps) return false;
    if (in_chain<4>(chain, t5)) return false;
    const uint32_t t6 = V.next[t5];
    if ((uint32_t)V.next[t4] == t5 || t6 == t4) return false;
    if (in_chain<4>(chain, t6)) return false;
    const float2 p6 = V.xy[t6];
    const float g4 = g3 + dist(p5, p6);
    chain[4] = t5;
    chain[5] = t6;
    return lk_chain<2, VT>(V, chain, clen, p1, t6, p6, g4);
}

// One explicit branch of find_lk_chain's loop (:290-337) at depth D: candidate q of t_open.  On success the chain grows by
// (t_next, t_break) and (t_open, p_open, gain) move on; a rejected branch returns false (for the `break` at g1 <= EPS see
// the exactness note above: it is implied by the later branches' own tests).
template <int D, typename VT>
__device__ __forceinline__ bool lk_branch(const VT &V, uint32_t (&chain)[kLkMaxChain], uint32_t &t_open, float2 &p_open,
                                          float &gain, const uint32_t q)
{
    uint32_t t_next;
    float2 p_next, p_break;
    float d1, d2;
    float4 rec;
    lk_probe(V, t_open, p_open, q, t_next, p_next, d1);
    const float g1 = gain - d1;
    if (g1 <= kLkEps) return false;
    if (in_chain<2 * D + 2>(chain, t_next)) return false;
    const uint32_t t_break = lk_break_id(V, t_next, rec);
    if (lk_next_of(V, t_open) == t_next || t_break == t_open) return false;
    if (in_chain<2 * D + 2>(chain, t_break)) return false;
    lk_break_pt(V, t_break, p_next, rec, p_break, d2);
    chain[2 * D + 2] = t_next;
    chain[2 * D + 3] = t_break;
    gain = g1 + d2;
    t_open = t_break;
    p_open = p_break;
    return true;
}

// Three split levels: sub = (q1 (k+1) + s1)(k+1) + s2 — branch q1 at depth 0; at depth 1 the closing test (s1 = 0) or
// branch s1-1; at depth 2 the closing test (s2 = 0) or branch s2-1, below which the walk is sequential (lk_chain<3>: at most
// k^2 nodes instead of k^3).  Lexicographic order of (q1, s1, s2) is the reference's DFS order, so the minimum index with a
// chain is the chain the sequential search returns; (q1, 0, s2 > 0) does not exist.
template <typename VT>
__device__ bool lk_subsearch3(const VT &V, uint32_t (&chain)[kLkMaxChain], uint32_t &clen, const float2 p1,
                              const uint32_t t2, const float2 p2, const float g0, const uint32_t q1, const uint32_t s1, const uint32_t s2)
{
    uint32_t t_open = t2;
    float2 p_open = p2;
    float gain = g0;
    if (0u >= V.max_depth) return false;
    if (!lk_branch<0, VT>(V, chain, t_open, p_open, gain, q1)) return false;
    {   // depth 1 (:280-288)
        const float close_gain = gain - dist(p_open, p1);
        if (s1 == 0u) {
            if (s2 == 0u && close_gain > kLkEps) {
                clen = 4;
                return true;
            }
            return false;
        }
        if (close_gain > kLkEps) return false;  // the sequential search returned at s1 = 0
        if (1u >= V.max_depth) return false;
    }
    if (!lk_branch<1, VT>(V, chain, t_open, p_open, gain, s1 - 1u)) return false;
    {   // depth 2
        const float close_gain = gain - dist(p_open, p1);
        if (s2 == 0u) {
            if (close_gain > kLkEps) {
                clen = 6;
                return true;
            }
            return false;
        }
        if (close_gain > kLkEps) return false;
        if (2u >= V.max_depth) return false;
    }
    if (!lk_branch<2, VT>(V, chain, t_open, p_open, gain, s2 - 1u)) return false;
    return lk_chain<3, VT>(V, chain, clen, p1, t_open, p_open, gain);
}

// lk_subsearch3 cut in two for the fused scan (k_lk_scan_sub): the FRONT walks the three split levels and stops where the
// sequential walk lk_chain<3> would begin — 0: no chain, 1: chain (clen set), 2: a walk is pending at depth 3 with
// (chain[0..8), t_open = chain[7], p_open, gain), its depth-3 closing test already failed.
template <typename VT>
__device__ int lk_subsearch3_front(const VT &V, uint32_t (&chain)[kLkMaxChain], uint32_t &clen, const float2 p1, const uint32_t t2,
                                   const float2 p2, const float g0, const uint32_t q1, const uint32_t s1, const uint32_t s2, float2 &p_open_out,
                                   float &gain_out)
{
    uint32_t t_open = t2;
    float2 p_open = p2;
    float gain = g0;
    if (0u >= V.max_depth) return 0;
    if (!lk_branch<0, VT>(V, chain, t_open, p_open, gain, q1)) return 0;
    {
        const float close_gain = gain - dist(p_open, p1);
        if (s1 == 0u) {
            if (s2 == 0u && close_gain > kLkEps) {
                clen = 4;
                return 1;
            }
            return 0;
        }
        if (close_gain > kLkEps) return 0;
        if (1u >= V.max_depth) return 0;
    }
    if (!lk_branch<1, VT>(V, chain, t_open, p_open, gain, s1 - 1u)) return 0;
    {
        const float close_gain = gain - dist(p_open, p1);
        if (s2 == 0u) {
            if (close_gain > kLkEps) {
                clen = 6;
                return 1;
            }
            return 0;
        }
        if (close_gain > kLkEps) return 0;
        if (2u >= V.max_depth) return 0;
    }
    if (!lk_branch<2, VT>(V, chain, t_open, p_open, gain, s2 - 1u)) return 0;
    // lk_chain<3>'s head (:280-288)
    if (gain - dist(p_open, p1) > kLkEps) {
        clen = 8;
        return 1;
    }
    if (3u >= V.max_depth) return 0;
    p_open_out = p_open;
    gain_out = gain;
    return 2;
}

// One (q3, s4) branch of a pending depth-3 walk: candidate q3 of t_open at depth 3; at depth 4 the closing test (s4 = 0) or
// candidate s4 - 1, below which lk_chain<5> is a closing test (and, only for max_depth = 6, one more candidate loop).
// Lexicographic order of (q3, s4) is the DFS order of lk_chain<3>, for the reasons given at lk_subsearch.
template <typename VT>
__device__ bool lk_tail_branch(const VT &V, uint32_t (&chain)[kLkMaxChain], uint32_t &clen, const float2 p1, float2 p_open, float gain,
                               const uint32_t q3, const uint32_t s4)
{
    uint32_t t_open = chain[7];
    if (!lk_branch<3, VT>(V, chain, t_open, p_open, gain, q3)) return false;
    const float close_gain = gain - dist(p_open, p1);
    if (s4 == 0u) {
        if (close_gain > kLkEps) {
            clen = 10;
            return true;
        }
        return false;
    }
    if (close_gain > kLkEps) return false;  // the sequential walk returned at s4 = 0
    if (4u >= V.max_depth) return false;
    if (!lk_branch<4, VT>(V, chain, t_open, p_open, gain, s4 - 1u)) return false;
    return lk_chain<5, VT>(V, chain, clen, p1, t_open, p_open, gain);
}

// the rest of lk_chain<3> for a walk whose head (closing test, depth limit) the front has already passed: the head is
// repeated (it fails again) — only used when the workgroup's queue of parked walks is full
template <typename VT>
__device__ bool lk_chain_from3(const VT &V, uint32_t (&chain)[kLkMaxChain], uint32_t &clen, const float2 p1, const float2 p_open, const float gain)
{
    return lk_chain<3, VT>(V, chain, clen, p1, chain[7], p_open, gain);
}

// the sub-search `sub` of pair (t1, t2) under the launch's split depth
__device__ __forceinline__ bool lk_run_sub(const LkArgs &G, uint32_t (&chain)[kLkMaxChain], uint32_t &clen, const float2 p1, const uint32_t t1,
                                           const uint32_t t2, const float2 p2, const float g0, const uint32_t sub)
{
    LkView V{G.xy, G.cand, G.next, G.k, G.max_depth};
    const uint32_t k1 = G.k + 1u;
    if (G.split_levels == 3u) return lk_subsearch3<LkView>(V, chain, clen, p1, t2, p2, g0, sub / (k1 * k1), (sub / k1) % k1, sub % k1);
    return lk_subsearch<LkView>(V, chain, clen, p1, t1, t2, p2, g0, sub / k1, sub % k1);
}

// BLOCK3D: one workgroup = one pair, threadIdx = (s2, s1, q1) — the sub-search digits come for free; the flat form (for
// k(k+1)^2 > 1024) recovers them with a 64-bit and three 32-bit divisions per lane, which is most of what a lane that fails
// its first test executes.
template <bool BLOCK3D, bool PK = false>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_lk_scan_sub(LkArgs G)
{
    if (G.state->finished) return;
    const uint32_t n = G.n, subs = lk_subs(G);
    uint32_t idx, sub, q1, s1, s2;
    uint64_t g;
    if (BLOCK3D) {
        // workgroups are dispatched in index order: the pairs at the END of the window — around the previous hit, where the next
        // chain most likely is and the long walks are — go first, so their walks overlap with the bulk instead of trailing it
        const uint32_t win = G.state->window;
        if (blockIdx.x >= win) return;
        idx = win - 1u - blockIdx.x;
#ifdef TL_LK_EXPERIMENT_EXIT
        if (G.city_ids[idx >> 1] != 0xFFFFFFFEu) return;  // one load, then exit: what does dispatching the window cost?
#endif
        const uint32_t k1 = G.k + 1u;
        if (G.split_levels == 3u) {
            q1 = threadIdx.z; s1 = threadIdx.y; s2 = threadIdx.x;
            sub = (q1 * k1 + s1) * k1 + s2;
        } else {
            q1 = threadIdx.y; s1 = threadIdx.x; s2 = 0u;
            sub = q1 * k1 + s1;
        }
        g = (uint64_t)idx * subs + sub;
    } else {
        g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
        if (g >= (uint64_t)2u * n * subs) return;
        idx = (uint32_t)(g / subs);
        sub = (uint32_t)(g % subs);
        const uint32_t k1 = G.k + 1u;
        if (G.split_levels == 3u) { q1 = sub / (k1 * k1); s1 = (sub / k1) % k1; s2 = sub % k1; }
        else { q1 = sub / k1; s1 = sub % k1; s2 = 0u; }
    }
    if (idx >= G.state->window) return;  // prefix window (k_lk_control)
    // fused pick (one workgroup = one pair): the pair's first chain in DFS order = the lowest sub-search index that found one;
    // its lane still holds the chain, validates it and posts the pair (what k_lk_scan_pick does from pairmin / subchains)
    __shared__ uint32_t s_minkey, s_qn;
    __shared__ uint32_t s_q[kLkTailCap][kLkTailWords];
    const bool fused = BLOCK3D && G.fused_pick != 0u;
    if (fused) {
        if (sub == 0u) {
            s_minkey = 0xFFFFFFFFu;
            s_qn = 0u;
            // chip-wide step: which of the two tour buffers is current changes hands here — no kernel reads it during a scan
            if (G.chip_step && idx == 0u) G.state->flip = G.state->flip_next;
        }
        TL_SYNC();
    }
    const uint32_t t1 = G.city_ids[idx >> 1];
    const uint32_t t2 = (idx & 1u) ? G.prev[t1] : G.next[t1];
    const float2 p1 = G.xy[t1], p2 = G.xy[t2];
    uint32_t chain[kLkMaxChain];
    uint32_t clen = 0;
    chain[0] = t1;
    chain[1] = t2;
    const float g0 = t1 == t2 ? 0.0f : dist(p1, p2);
    if (fused && G.split_levels == 3u) {
        // (PK: the packed view, where the step kernel keeps LkArgs::nx — an instantiation of its own, the classic one's code is untouched)
        using VT = typename std::conditional<PK, LkViewPk, LkView>::type;
        VT V;
        if constexpr (PK) V = LkViewPk{G.candd, G.nx, G.k, G.max_depth};
        else V = LkView{G.xy, G.cand, G.next, G.k, G.max_depth};
        // Deferred tails.  What a scan waits for is the few lanes whose walk survives the three split levels and goes on
        // alone through up to k + k^2 candidates, three dependent look-ups each (measured: 58 us per scan, 23 us with
        // those walks cut off).  So a surviving lane parks its state in LDS and the workgroup — most of whose lanes failed
        // their first test long ago — deals the next two levels out again: lane = (parked walk, candidate q3 at depth 3,
        // closing test or candidate at depth 4).  The order key (sub, q3, s4) is the walk's DFS order, so the smallest
        // key with a chain is the reference's first chain.
        const uint32_t nthr = blockDim.x * blockDim.y * blockDim.z, k1 = G.k + 1u, per = G.k * k1;
        uint32_t mykey = 0xFFFFFFFFu;
        float2 p_open;
        float gain;
        const int st = lk_subsearch3_front(V, chain, clen, p1, t2, p2, g0, q1, s1, s2, p_open, gain);
        if (st == 1) mykey = sub * (per + 1u);
        if (st == 2) {
            const uint32_t slot = atomicAdd(&s_qn, 1u);
            if (slot < kLkTailCap) {
#pragma unroll
                for (int t = 0; t < 6; ++t) s_q[slot][t] = chain[2 + t];
                s_q[slot][6] = __float_as_uint(gain);
                s_q[slot][7] = sub;
                s_q[slot][8] = __float_as_uint(p_open.x);
                s_q[slot][9] = __float_as_uint(p_open.y);
            } else if (lk_chain_from3(V, chain, clen, p1, p_open, gain)) {  // queue full: walk on alone
                mykey = sub * (per + 1u);
            }
        }
        TL_SYNC();
#ifdef TL_LK_EXPERIMENT_NOPHASE2  // timing experiments only (wrong results)
        const uint32_t nq = 0;
#else
        const uint32_t nq = s_qn < kLkTailCap ? s_qn : kLkTailCap;
#endif
        for (uint32_t w = sub; w < nq * per; w += nthr) {
            const uint32_t e = w / per, r = w - e * per, q3 = r / k1, s4 = r - q3 * k1;
            const uint32_t key = s_q[e][7] * (per + 1u) + 1u + r;
            if (key >= mykey) continue;  // this lane already holds an earlier chain
            uint32_t c2[kLkMaxChain];
            uint32_t l2 = 0;
            c2[0] = t1;
            c2[1] = t2;
#pragma unroll
            for (int t = 0; t < 6; ++t) c2[2 + t] = s_q[e][t];
            const float2 po = make_float2(__uint_as_float(s_q[e][8]), __uint_as_float(s_q[e][9]));
            if (lk_tail_branch(V, c2, l2, p1, po, __uint_as_float(s_q[e][6]), q3, s4)) {
                mykey = key;
                clen = l2;
#pragma unroll
                for (int t = 0; t < kLkMaxChain; ++t) chain[t] = c2[t];
            }
        }
        if (mykey != 0xFFFFFFFFu) atomicMin(&s_minkey, mykey);
        TL_SYNC();
        if (mykey != 0xFFFFFFFFu && s_minkey == mykey && chain_valid(chain, clen, G.tour, G.pos, n)) {
            uint32_t *slot = G.chains + (size_t)idx * kLkSlot;
            slot[0] = clen;
            for (uint32_t t = 0; t < clen; ++t) slot[1 + t] = chain[t];
            if (G.chip_step) {  // the chain's tour positions and the move's segment table, for the step kernel (a call: see lk_file_move)
                lk_file_move(slot, clen, G.pos, n);
                atomicMin(&G.state->key2[G.parity], idx);
            } else {
                atomicMin(&G.state->key, idx);
            }
        }
        return;
    
        return;
    }
    LkView V{G.xy, G.cand, G.next, G.k, G.max_depth};
    const bool got = G.split_levels == 3u ? lk_subsearch3<LkView>(V, chain, clen, p1, t2, p2, g0, q1, s1, s2)
                                          : lk_subsearch<LkView>(V, chain, clen, p1, t1, t2, p2, g0, q1, s1);
    if (fused) {
        if (got) atomicMin(&s_minkey, sub);
        TL_SYNC();
        if (got && s_minkey == sub && chain_valid(chain, clen, G.tour, G.pos, n)) {
            uint32_t *slot = G.chains + (size_t)idx * kLkSlot;
            slot[0] = clen;
            for (uint32_t t = 0; t < clen; ++t) slot[1 + t] = chain[t];
            atomicMin(&G.state->key, idx);
        }
        return;
    }
    if (got) {
        if (G.subchains) {  // keep the chain: the pick step reads the winner's instead of walking it again
            uint32_t *slot = G.subchains + g * kLkSubSlot;
            slot[0] = clen;
#pragma unroll
            for (int t = 0; t < kLkMaxChain; ++t) slot[1 + t] = chain[t];
        }
        atomicMin(&G.pairmin[idx], sub);
    }
}

#ifdef TL_TUNE
// MEASURED AND REJECTED (tuning build only): the fused three-level scan as a PERSISTENT grid (round 4): a fixed number of workgroups — what the chip
// holds at 8 waves per SIMD — stride over the window's pairs instead of one workgroup per pair of the whole tour, of which all
// beyond the window exited after one load (dispatching those 2n workgroups was 8-9 us of a 34.5 us scan at n = 13 509).  Same
// per-pair code as k_lk_scan_sub<true>'s fused branch; the per-pair LDS words are double-buffered by the iteration's parity (a
// workgroup's barriers keep its waves within one iteration of each other), and a pair whose index lies above an already posted
// one does nothing (it cannot be the lowest valid pair: lin_kernighan.rs:345-389 takes the first in t1 order).
#ifndef TL_LK_PERSIST_ASCENDING
#define TL_LK_PERSIST_ASCENDING 0
#endif
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_lk_scan_persist(LkArgs G)
{
    if (G.state->finished) return;
    const uint32_t n = G.n;
    const uint32_t win = G.state->window;
    __shared__ uint32_t s_minkey[2], s_qn[2];
    __shared__ uint32_t s_q[2][kLkTailCap][kLkTailWords];
    const uint32_t k1 = G.k + 1u, per = G.k * k1;
    const uint32_t q1 = threadIdx.z, s1 = threadIdx.y, s2 = threadIdx.x;
    const uint32_t sub = (q1 * k1 + s1) * k1 + s2;
    const uint32_t nthr = blockDim.x * blockDim.y * blockDim.z;
    // chip-wide step: which of the two tour buffers is current changes hands here — no kernel reads it during a scan
    if (G.chip_step && blockIdx.x == 0u && sub == 0u) G.state->flip = G.state->flip_next;
    LkView V{G.xy, G.cand, G.next, G.k, G.max_depth};
    uint32_t par = 0u;
    for (uint32_t blk = blockIdx.x; blk < win; blk += gridDim.x, par ^= 1u) {
        const uint32_t idx = TL_LK_PERSIST_ASCENDING ? blk : win - 1u - blk;
        if (sub == 0u) {
            s_minkey[par] = 0xFFFFFFFFu;
            s_qn[par] = 0u;
        }
        // (An early-out for pairs above an already posted one was measured and dropped: an agent-scope read of the key word by
        //  every workgroup — one address for the whole chip — cost more than the skipped walks saved: +10 us per round with one lane
        //  reading in front of the barrier, +27 us with every lane reading beside its own loads.)
        const bool dead = false;
        const uint32_t t1 = G.city_ids[idx >> 1];
        const uint32_t t2 = (idx & 1u) ? G.prev[t1] : G.next[t1];
        const float2 p1 = G.xy[t1], p2 = G.xy[t2];
        TL_SYNC();
        uint32_t chain[kLkMaxChain];
        uint32_t clen = 0;
        chain[0] = t1;
        chain[1] = t2;
        const float g0 = t1 == t2 ? 0.0f : dist(p1, p2);
        uint32_t mykey = 0xFFFFFFFFu;
        float2 p_open;
        float gain;
        const int st = dead ? 0 : lk_subsearch3_front<LkView>(V, chain, clen, p1, t2, p2, g0, q1, s1, s2, p_open, gain);
        if (st == 1) mykey = sub * (per + 1u);
        if (st == 2) {
            const uint32_t slot = atomicAdd(&s_qn[par], 1u);
            if (slot < kLkTailCap) {
#pragma unroll
                for (int t = 0; t < 6; ++t) s_q[par][slot][t] = chain[2 + t];
                s_q[par][slot][6] = __float_as_uint(gain);
                s_q[par][slot][7] = sub;
                s_q[par][slot][8] = __float_as_uint(p_open.x);
                s_q[par][slot][9] = __float_as_uint(p_open.y);
            } else if (lk_chain_from3<LkView>(V, chain, clen, p1, p_open, gain)) {  // queue full: walk on alone
                mykey = sub * (per + 1u);
            }
        }
        TL_SYNC();
        const uint32_t nq = s_qn[par] < kLkTailCap ? s_qn[par] : kLkTailCap;
        for (uint32_t w = sub; w < nq * per; w += nthr) {
            const uint32_t e = w / per, r = w - e * per, q3 = r / k1, s4 = r - q3 * k1;
            const uint32_t key = s_q[par][e][7] * (per + 1u) + 1u + r;
            if (key >= mykey) continue;  // this lane already holds an earlier chain
            uint32_t c2[kLkMaxChain];
            uint32_t l2 = 0;
            c2[0] = t1;
            c2[1] = t2;
#pragma unroll
            for (int t = 0; t < 6; ++t) c2[2 + t] = s_q[par][e][t];
            const float2 po = make_float2(__uint_as_float(s_q[par][e][8]), __uint_as_float(s_q[par][e][9]));
            if (lk_tail_branch<LkView>(V, c2, l2, p1, po, __uint_as_float(s_q[par][e][6]), q3, s4)) {
                mykey = key;
                clen = l2;
#pragma unroll
                for (int t = 0; t < kLkMaxChain; ++t) chain[t] = c2[t];
            }
        }
        if (mykey != 0xFFFFFFFFu) atomicMin(&s_minkey[par], mykey);
        TL_SYNC();
        if (mykey != 0xFFFFFFFFu && s_minkey[par] == mykey && chain_valid(chain, clen, G.tour, G.pos, n)) {
            uint32_t *slot = G.chains + (size_t)idx * kLkSlot;
            slot[0] = clen;
            for (uint32_t t = 0; t < clen; ++t) slot[1 + t] = chain[t];
            if (G.chip_step) {
                lk_file_move(slot, clen, G.pos, n);
                atomicMin(&G.state->key2[G.parity], idx);
            } else {
                atomicMin(&G.state->key, idx);
            }
        }
    }
}
#endif  // TL_TUNE

__global__ __launch_bounds__(256) void k_lk_scan_pick(LkArgs G)
{
    LkState *S = G.state;
    if (S->finished) return;
    const uint32_t n = G.n, idx = blockIdx.x * 256u + threadIdx.x;
    if (idx >= 2u * n || idx >= S->window) return;
    const uint32_t sub = G.pairmin[idx];
    if (sub == 0xFFFFFFFFu) return;
    G.pairmin[idx] = 0xFFFFFFFFu;  // ready for the next scan
    const uint32_t t1 = G.city_ids[idx >> 1];
    const uint32_t t2 = (idx & 1u) ? G.prev[t1] : G.next[t1];
    const float2 p1 = G.xy[t1], p2 = G.xy[t2];
    uint32_t chain[kLkMaxChain];
    uint32_t clen = 0;
    bool have;
    if (G.subchains) {
        const uint32_t *src = G.subchains + ((uint64_t)idx * lk_subs(G) + sub) * kLkSubSlot;
        clen = src[0];
#pragma unroll
        for (int t = 0; t < kLkMaxChain; ++t) chain[t] = src[1 + t];
        have = true;
    } else {
        chain[0] = t1;
        chain[1] = t2;
        const float g0 = t1 == t2 ? 0.0f : dist(p1, p2);
        have = lk_run_sub(G, chain, clen, p1, t1, t2, p2, g0, sub);
    }
    if (have && chain_valid(chain, clen, G.tour, G.pos, n)) {
        uint32_t *slot = G.chains + (size_t)idx * kLkSlot;
        slot[0] = clen;
        for (uint32_t t = 0; t < clen; ++t) slot[1 + t] = chain[t];
        atomicMin(&S->key, idx);
    }
}

// The state machine behind a scan.  CHIP (n >= kLkRebuildSplitN with the fused scan): one launch of ceil(n / 1024) workgroups
// does what k_lk_control + k_lk_rebuild do in two — every workgroup builds the move's segment table for itself (from the chain
// and the tour positions the scan left in the pair's slot) and writes its slice of the new tour into the OTHER tour buffer
// together with rank / successor / predecessor; workgroup 0 also keeps the counters, and runs the pass-end logic alone when the
// scan found nothing.  What other workgroups of the same launch read is never written in it: the scan's key is double-buffered
// by the round's parity (a launch argument) and the current-buffer flag changes hands in the next scan.
template <bool CHIP>
__global__ __launch_bounds__(kLkNT) void k_lk_control(LkArgs G)
{
    __shared__ uint32_t s_nseg;
    __shared__ LkSeg s_seg[2 * (kLkMaxDepth + 2)];
    __shared__ float s_part[kLkNT];
    LkState *S = G.state;
    if (S->finished) return;
    const uint32_t tid = threadIdx.x, n = G.n;
    uint32_t *tour = G.tour, *alt = G.alt, *pos = G.pos, *next = G.next, *prev = G.prev, *best = G.best;
    if (CHIP && S->flip) {
        tour = G.alt;
        alt = G.tour;
    }
    const float2 *__restrict__ xy = G.xy;

    auto rebuild = [&]() {
        for (uint32_t r = tid; r < n; r += kLkNT) {
            const uint32_t c = tour[r], cn = tour[r + 1u == n ? 0u : r + 1u];
            pos[c] = r;
            next[c] = cn;
            prev[c] = tour[r == 0u ? n - 1u : r - 1u];
            if (G.nx) {
                const float2 pn = xy[cn];
                G.nx[c] = make_float4(__uint_as_float(cn), pn.x, pn.y, dist(xy[c], pn));
            }
        }
    };
    auto tour_distance = [&](const uint32_t *t) -> float {  // lin_kernighan.rs:118-122
        float total = 0.0f;
        for (uint32_t base = 0; base < n; base += kLkNT) {
            const uint32_t r = base + tid;
            if (r < n) {
                const uint32_t a = t[r], b = t[r + 1u == n ? 0u : r + 1u];
                s_part[tid] = a == b ? 0.0f : dist(xy[a], xy[b]);
            }
            TL_SYNC();
            if (tid == 0) {
                const uint32_t cnt = (n - base) < (uint32_t)kLkNT ? (n - base) : (uint32_t)kLkNT;
                for (uint32_t q = 0; q < cnt; ++q) total += s_part[q];
            }
            TL_SYNC();
        }
        if (tid == 0) s_part[0] = total;
        TL_SYNC();
        const float r = s_part[0];
        TL_SYNC();
        return r;
    };

    // lin_kernighan.rs:71,90 send_progress(best_tour, best_dist): the tour just copied into `best` and its length, appended to
    // the caller's snapshot list (tl_lk_trace)
    auto snapshot = [&](const uint32_t *t, float d) {
        if (!G.snap) return;
        const uint32_t sidx = S->snaps;
        TL_SYNC();  // every thread has read the count
        if (G.snap_ring || sidx < G.snap_cap) {  // ring (tl_lk_live: the host drains it between polls) or a list of the first snap_cap
            const uint32_t at = G.snap_ring ? sidx % G.snap_cap : sidx;
            for (uint32_t r = tid; r < n; r += kLkNT) G.snap[(size_t)at * n + r] = t[r];
            if (tid == 0) G.snap_dist[at] = d;
        }
        if (tid == 0) S->snaps = sidx + 1u;
    };

    const uint32_t key = CHIP ? S->key2[G.parity] : S->key;
    if (!CHIP && tid == 0) S->applied = 0u;  // set again below if this round applies a move (k_lk_rebuild runs after every control)
    TL_SYNC();
    if (key != 0xFFFFFFFFu) {
        // ---- apply_lk_chain (:397-450), then rescan (lk_pass loop :468-478)
        const uint32_t *slot = G.chains + (size_t)key * kLkSlot;
        const uint32_t clen = slot[0];
        if (CHIP) {
            // the scan lane that validated this chain filed the move's segment table in the slot (lk_file_move)
            if (tid == 0) s_nseg = slot[kLkSlotSeg];
            if (tid < 4u * kLkMaxSeg) reinterpret_cast<uint32_t *>(s_seg)[tid] = slot[kLkSlotSeg + 1u + tid];
        } else if (tid == 0) {
            uint32_t chain[kLkMaxChain];
            for (uint32_t t = 0; t < clen; ++t) chain[t] = slot[1 + t];
            uint32_t cpos[kLkMaxChain];
            chain_positions(chain, clen, pos, cpos);
            s_nseg = lk_build_segments(cpos, clen, n, s_seg);
        }
        TL_SYNC();
        const uint32_t nseg = s_nseg;
        if (CHIP) {
            // this workgroup's slice of the new tour: the city at new position d comes from old position src(d)
            auto city_at = [&](uint32_t d) -> uint32_t {
                uint32_t sidx = 0;
                while (sidx + 1u < nseg && d >= s_seg[sidx + 1u].dst) ++sidx;  // segments are in dst order
                const LkSeg sg = s_seg[sidx];
                const uint32_t t = d - sg.dst;
                uint32_t sp = sg.dir > 0 ? sg.src + t : sg.src + n - t;
                if (sp >= n) sp -= n;
                return tour[sp];
            };
            for (uint32_t r = blockIdx.x * kLkNT + tid; r < n; r += gridDim.x * kLkNT) {
                const uint32_t c = city_at(r), cn = city_at(r + 1u == n ? 0u : r + 1u), cp = city_at(r == 0u ? n - 1u : r - 1u);
                alt[r] = c;
                pos[c] = r;
                next[c] = cn;
                prev[c] = cp;
                if (G.nx) {  // the packed view's record of c: its successor, the successor's point, the tour edge's length
                    const float2 pn = xy[cn];
                    G.nx[c] = make_float4(__uint_as_float(cn), pn.x, pn.y, dist(xy[c], pn));
                }
            }
            if (blockIdx.x == 0 && tid == 0) {
                S->scans += 1;
                S->searches += (uint64_t)key + 1u;
                S->moves += 1;
                S->exchanged += clen / 2u;
                if (++S->pass_moves > lk_pass_cap(n)) S->finished = 2u;  // a cycling lk_pass (see lk_pass_cap): the host reports it
                S->key2[G.parity ^ 1u] = 0xFFFFFFFFu;  // the next scan's key
                const uint32_t w = key + kLkWindowMargin;
                S->window = w < 2u * n ? w : 2u * n;
                S->flip_next = S->flip ^ 1u;
            }
            return;
        }
        for (uint32_t sidx = 0; sidx < nseg; ++sidx) {
            const LkSeg sg = s_seg[sidx];
            for (uint32_t t = tid; t < sg.len; t += kLkNT) {
                uint32_t sp = sg.dir > 0 ? sg.src + t : sg.src + n - t;
                if (sp >= n) sp -= n;
                alt[sg.dst + t] = tour[sp];
            }
        }
        // the new tour is in `alt`: copy it back and rebuild rank / successor / predecessor — here for small tours, in
        // k_lk_rebuild on all CUs for large ones (3n scattered stores are slow from one CU)
        if (n < kLkRebuildSplitN) {
            TL_SYNC();
            for (uint32_t r = tid; r < n; r += kLkNT) {
                const uint32_t c = alt[r];
                tour[r] = c;
                pos[c] = r;
                next[c] = alt[r + 1u == n ? 0u : r + 1u];
                prev[c] = alt[r == 0u ? n - 1u : r - 1u];
            }
        }
        if (tid == 0) {
            S->applied = n < kLkRebuildSplitN ? 0u : 1u;
            S->scans += 1;
            S->searches += (uint64_t)key + 1u;
            S->moves += 1;
            S->exchanged += clen / 2u;
            if (++S->pass_moves > lk_pass_cap(n)) S->finished = 2u;
            S->key = 0xFFFFFFFFu;
            // find_lk_move restarts at pair 0 after every move (:468-478) and keeps the LOWEST pair with a valid chain, so a
            // scan of a prefix that contains a hit is a complete scan.  The pairs before this hit had no valid chain a moment
            // ago; the next hit is most often near or behind this one: look at [0, key + margin) first.
            const uint32_t w = key + kLkWindowMargin;
            S->window = w < 2u * n ? w : 2u * n;
        }
        return;
    }
    if (CHIP) {  // no move: workgroup 0 is the state machine
        if (blockIdx.x != 0) return;
        if (tid == 0) {
            S->key2[G.parity ^ 1u] = 0xFFFFFFFFu;
            S->flip_next = S->flip;
        }
    }
    if (S->window < 2u * n) {  // nothing inside the prefix: the same find_lk_move goes on over all pairs
        TL_SYNC();
        if (tid == 0) S->window = 2u * n;
        return;
    }
    // ---- the scan found nothing: this lk_pass is over (:472-473)
    if (tid == 0) {
        S->scans += 1;
        S->searches += 2ull * n;
        S->pass_moves = 0u;
    }
    bool kick = false;
    if (S->stage == 0) {                       // initial pass done (:61-70)
        for (uint32_t r = tid; r < n; r += kLkNT) best[r] = tour[r];
        TL_SYNC();
        const float bd = tour_distance(best);
        if (tid == 0) {
            S->best_dist = bd;
            S->stage = 1;
            S->epoch = 0;
            S->platoo = 0;
        }
        snapshot(tour, bd);
        kick = G.epochs > 0;
    } else {                                    // an epoch's pass done (:85-96)
        const float dcur = tour_distance(tour);
        const bool better = dcur < S->best_dist;
        TL_SYNC();
        bool stop = false;
        if (better) {
            for (uint32_t r = tid; r < n; r += kLkNT) best[r] = tour[r];
            snapshot(tour, dcur);
        }
        uint32_t platoo = S->platoo, epoch = S->epoch;
        TL_SYNC();
        if (better) platoo = 0;
        else if (++platoo >= G.platoo_epochs) stop = true;
        ++epoch;
        if (tid == 0) {
            if (better) S->best_dist = dcur;
            S->platoo = platoo;
            S->epoch = epoch;
        }
        kick = !stop && epoch < G.epochs;
    }
    TL_SYNC();
    if (!kick) {
        if (tid == 0) S->finished = 1;
        return;
    }
    // ---- double_bridge (:485-499) into `tour`, then a fresh lk_pass: city_ids = tour (:466)
    if (n < 8) {
        for (uint32_t r = tid; r < n; r += kLkNT) tour[r] = best[r];
    } else {
        const uint64_t draws = S->draws;
        const uint32_t qn = n / 4u;
        const uint32_t r1 = (uint32_t)(splitmix64_at(G.seed, draws) % qn), r2 = (uint32_t)(splitmix64_at(G.seed, draws + 1) % qn),
                       r3 = (uint32_t)(splitmix64_at(G.seed, draws + 2) % qn);
        const uint32_t p1 = 1u + r1, p2 = p1 + 1u + r2, p3 = p2 + 1u + r3;
        for (uint32_t w = tid; w < n; w += kLkNT) {
            uint32_t src;
            if (w < p1) src = w;
            else if (w < p1 + (p3 - p2)) src = p2 + (w - p1);
            else if (w < p3) src = p1 + (w - p1 - (p3 - p2));
            else src = w;
            tour[w] = best[src];
        }
        TL_SYNC();
        if (tid == 0) S->draws = draws + 3;
    }
    TL_SYNC();
    for (uint32_t r = tid; r < n; r += kLkNT) G.city_ids[r] = tour[r];
    rebuild();
    if (tid == 0) {
        S->key = 0xFFFFFFFFu;
        S->window = kLkWindowFirst < 2u * n ? kLkWindowFirst : 2u * n;  // a fresh pass: hits start at the front
    }
}


// ------------------------------------------------------------------------------------------------
// k_lk_ils — the whole ILS of a SMALL instance (round 5): ONE persistent workgroup, every array the search
// touches in this CU's LDS as 16-bit city ids (coordinates, candidate lists, tour, scratch tour, rank, successor, predecessor,
// scan order, best tour: n (22 + 2k) bytes), no kernel boundary per round — at berlin52's size a round of the chip-wide form is
// 23.6 us of launch latency around a scan of 52 x 2 x 3 candidates.
//
// What made the round-2 single-workgroup form (k_lk_solve<NT, true>) slow was one LANE walking a pair's whole depth-first search
// (up to k + k^2 + ... + k^depth nodes, three dependent look-ups each).  Here find_lk_move (:345-389) runs LEVEL BY LEVEL over a
// chunk of consecutive (t1, orientation) pairs: a lane is (live node, candidate q) — one branch of find_lk_chain's loop (:290-337)
// — and files the child it creates, after the child's own closing test (:280-283), in the next level's queue.  A node's place in
// the reference's depth-first order is its PATH (q0, q1, ...): as a key of `bits` per level, digit q + 1, most significant first,
// zeros behind a node that closes — so a closing node sorts before its own descendants and after everything under an earlier
// sibling, exactly as the recursion meets them.  The pair's first chain is the minimum key that closed (ds_min_u64 per pair), and
// nodes behind an already posted key are not expanded.  Exactness of dropping the `break` at g1 <= EPS (:292-295): see lk_subsearch.
// Then, as in the reference (:373-381): only that first chain of a pair is checked for a single cycle; the lowest pair with a
// valid one wins; its lane builds the move's segment table; the workgroup copies the arcs.  Chunks grow geometrically from the
// front (hits of a running pass sit near the front), a queue overflow halves the chunk and repeats it (the plan guarantees that a
// chunk of one pair fits: k^(max_depth-1) <= queue capacity).
//
// The ILS state machine (first pass, kick, accept / plateau: :61-97) runs in the same workgroup; a launch is a SLICE of at most
// LkArgs::ils_slice scans — the state lives in LkState + the global tour / best / city_ids between slices — so that the host can
// drain progress snapshots (tl_lk_live) and a run of minutes is no single launch.
namespace {
__device__ __forceinline__ uint32_t readlane_u32(uint32_t v, uint32_t l)  // l: wave-uniform
{
    return (uint32_t)__builtin_amdgcn_readlane((int)v, __builtin_amdgcn_readfirstlane((int)l));
}
constexpr int kIlsNT = 1024;
#ifndef TL_ILS_SMALL_N
#define TL_ILS_SMALL_N 200u
#endif
constexpr uint32_t kIlsSmallN = TL_ILS_SMALL_N;  // up to here k_lk_ils runs on 4 waves
constexpr uint32_t kIlsHdr = 4u;  // record words in front of the chain: key lo, key hi, gain, pair (chunk-local)

__host__ __device__ __forceinline__ uint32_t ils_bits(uint32_t k)
{
    uint32_t b = 1u;
    while ((1u << b) < k + 1u) ++b;
    return b;
}
__host__ __device__ __forceinline__ size_t ils_up16(size_t v) { return (v + 15u) & ~(size_t)15u; }
// LDS image: xy | cand | dcand (the candidates' distances) | dnext (every city's tour edge to its successor) | 7 x u16[n] |
// pairmin u64[kIlsNT] | two queues of qcap records of (kIlsHdr + max_depth) words
__host__ __device__ __forceinline__ size_t ils_fixed_bytes(uint32_t n, uint32_t k)
{
    return ils_up16((size_t)n * 8) + ils_up16((size_t)n * k * 2) + ils_up16((size_t)n * k * 4) + ils_up16((size_t)n * 4) + 7u * ils_up16((size_t)n * 2) +
           (size_t)kIlsNT * 8;
}
// the kernel's static __shared__: per-wave segment tables + words
constexpr size_t kIlsStatic = (size_t)(kIlsNT / 64) * kLkMaxSeg * 16 + 1024;

}  // namespace

// the form applies iff this returns a queue capacity (records per level) > 0
uint32_t lk_ils_qcap(uint32_t n, uint32_t k, uint32_t max_depth, size_t lds_budget)
{
    if (n < 4 || n > 65535u || k == 0 || k > 16u || max_depth == 0 || max_depth > (uint32_t)kLkMaxDepth) return 0u;  // (k <= 16: the task -> (node, candidate) split of a level)
    if (ils_bits(k) * max_depth > 64u) return 0u;
    const size_t fixed = ils_fixed_bytes(n, k) + kIlsStatic;
    if (fixed >= lds_budget) return 0u;
    const size_t rec = (size_t)(kIlsHdr + max_depth) * 4;
    size_t qcap = (lds_budget - fixed) / (2 * rec);
    // a chunk of ONE pair must fit: its widest queued level holds k^(max_depth-1) nodes; the queues also serve as the n floats of
    // the ordered cost sum
    uint64_t need = 1;
    for (uint32_t d = 1; d < max_depth; ++d) {
        need *= k;
        if (need > qcap) return 0u;
    }
    // no more than the search uses: a level of a chunk of <= 1024 pairs rarely holds more than a few hundred nodes (an overflow
    // only halves the chunk), and a small image lets two epochs share a CU
    size_t cap = need > 512 ? need : 512;
    if ((size_t)n * 4 > 2 * cap * rec) cap = ((size_t)n * 4 + 2 * rec - 1) / (2 * rec);
    if (cap > 2048) cap = 2048;
    if (qcap > cap) qcap = cap;
    if (qcap < 64 || 2 * qcap * rec < (size_t)n * 4) return 0u;
    return (uint32_t)qcap;
}

size_t lk_ils_lds_bytes(uint32_t n, uint32_t k, uint32_t max_depth, uint32_t qcap)
{
    return ils_fixed_bytes(n, k) + 2 * (size_t)qcap * (kIlsHdr + max_depth) * 4 + kIlsStatic;
}

namespace {

template <int NT>
__global__ __launch_bounds__(NT) void k_lk_ils(LkArgs G)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char ils_smem[];
    __shared__ uint32_t s_cnt[3];
    __shared__ uint32_t s_nsegw[NT / 64], s_clenw[NT / 64], s_widx[NT / 64], s_wcnt[NT / 64];
    __shared__ unsigned long long s_wmask[NT / 64];
    __shared__ LkSeg s_segw[NT / 64][kLkMaxSeg];  // per wave: the segment table of its first valid chain
    __shared__ float s_bd;
    const uint32_t tid = threadIdx.x, n = G.n, k = G.k, D = G.max_depth, qcap = G.ils_qcap;
    const uint32_t bits = ils_bits(k), RW = kIlsHdr + D;
    LkState *S = G.state;
    // ---- LDS image
    unsigned char *sp = ils_smem;
    float2 *xy = reinterpret_cast<float2 *>(sp);
    sp += ils_up16((size_t)n * 8);
    uint16_t *cand = reinterpret_cast<uint16_t *>(sp);
    sp += ils_up16((size_t)n * k * 2);
    float *const dcand = reinterpret_cast<float *>(sp);  // d(t, cand[t][q]): the g1 term of a branch costs no square root
    sp += ils_up16((size_t)n * k * 4);
    float *const dnext = reinterpret_cast<float *>(sp);  // d(c, next[c]) of the current tour: the g2 term (t_break = next[t_next]) and g0
    sp += ils_up16((size_t)n * 4);
    const size_t a16 = ils_up16((size_t)n * 2);
    uint16_t *tour = reinterpret_cast<uint16_t *>(sp), *alt = reinterpret_cast<uint16_t *>(sp + a16);
    uint16_t *const pos = reinterpret_cast<uint16_t *>(sp + 2 * a16), *const next = reinterpret_cast<uint16_t *>(sp + 3 * a16),
                    *const prev = reinterpret_cast<uint16_t *>(sp + 4 * a16), *const ids = reinterpret_cast<uint16_t *>(sp + 5 * a16),
                    *const best = reinterpret_cast<uint16_t *>(sp + 6 * a16);
    sp += 7 * a16;
    unsigned long long *pairmin = reinterpret_cast<unsigned long long *>(sp);
    sp += (size_t)kIlsNT * 8;
    uint32_t *const fq0 = reinterpret_cast<uint32_t *>(sp);
    uint32_t *const fq1 = fq0 + (size_t)qcap * RW;
    float *fsum = reinterpret_cast<float *>(fq0);  // the ordered cost sum's edge lengths (the queues are idle then)

    // mode 0: the whole ILS in this one workgroup, slice by slice.  mode 1: the first lk_pass only (:61-70), then the state goes back.
    // mode 2: ONE EPOCH per workgroup — workgroup j of the launch kicks the current best tour with the draws of epoch S->epoch + j,
    // runs that epoch's lk_pass and files the tour it ends on, its length and its counters in slot j; k_lk_ils_commit then takes the
    // epochs in order up to the first one that is accepted (the later ones started from a best tour that is no longer the best).
    const uint32_t mode = G.ils_mode, slot = blockIdx.x;
    // (mode 2: the batch's width is S->applied — k_lk_ils_commit adapts it to how soon the last batches were decided: while every second
    //  epoch is accepted a batch of 512 only waits for its slowest epoch; the grid stays ils_P wide, the workgroups beyond the width return)
    if (mode == 2u && (S->finished != 0u || S->epoch + slot >= G.epochs || slot >= S->applied)) return;
    const bool fresh = mode != 2u && S->applied == 0u;
    for (uint32_t r = tid; r < n; r += NT) {
        xy[r] = G.xy[r];
        const uint32_t c = G.tour[r];
        tour[r] = (uint16_t)c;
        ids[r] = (uint16_t)(fresh ? c : G.city_ids[r]);    // :466 city_ids = tour at a pass's start
        best[r] = (uint16_t)(fresh ? c : G.best[r]);
    }
    if (mode == 2u && G.ils_dcand) {  // (written by the first-pass launch)
        for (uint32_t e = tid; e < n * k; e += NT) {
            cand[e] = (uint16_t)G.cand[e];
            dcand[e] = G.ils_dcand[e];
        }
    } else {
        for (uint32_t e = tid; e < n * k; e += NT) {
            const uint32_t c = G.cand[e];
            cand[e] = (uint16_t)c;
            const float dd = dist(G.xy[e / k], G.xy[c]);
            dcand[e] = dd;
            if (G.ils_dcand) G.ils_dcand[e] = dd;
        }
    }
    uint32_t stage = S->stage, epoch = S->epoch, platoo = S->platoo, snaps = S->snaps;
    float best_dist = S->best_dist;
    uint64_t draws = S->draws, scans = S->scans, searches = S->searches, moves = S->moves, exchanged = S->exchanged;
    TL_SYNC();

    auto rebuild = [&]() {  // make_pos + flat_to_next_prev (:110-145)
        for (uint32_t r = tid; r < n; r += NT) {
            const uint32_t c = tour[r];
            pos[c] = (uint16_t)r;
            const uint32_t cn = tour[r + 1u == n ? 0u : r + 1u];
            next[c] = (uint16_t)cn;
            prev[c] = tour[r == 0u ? n - 1u : r - 1u];
            dnext[c] = dist(xy[c], xy[cn]);
        }
        TL_SYNC();
    };

    // tour_distance (:118-122): sequential f32 sum from 0 in tour order, the closing edge last.  Only ever asked of the current tour
    // (behind a rebuild): its edge lengths are dnext.
    auto tour_distance = [&](const uint16_t *t) -> float {
        for (uint32_t r = tid; r < n; r += NT) fsum[r] = dnext[t[r]];
        TL_SYNC();
        if (tid == 0) {
            float total = 0.0f;
            uint32_t q = 0;
            const float4 *f4 = reinterpret_cast<const float4 *>(fsum);
            for (; q + 4 <= n; q += 4) {
                const float4 v = f4[q >> 2];
                total += v.x;
                total += v.y;
                total += v.z;
                total += v.w;
            }
            for (; q < n; ++q) total += fsum[q];
            s_bd = total;
        }
        TL_SYNC();
        const float r = s_bd;
        TL_SYNC();
        return r;
    };

    auto snapshot = [&](const uint16_t *t, float d) {  // :71,90 send_progress(best_tour, best_dist)
        if (G.snap) {
            if (G.snap_ring || snaps < G.snap_cap) {
                const uint32_t at = G.snap_ring ? snaps % G.snap_cap : snaps;
                for (uint32_t r = tid; r < n; r += NT) G.snap[(size_t)at * n + r] = t[r];
                if (tid == 0) G.snap_dist[at] = d;
            }
        }
        ++snaps;
    };

    // one branch of find_lk_chain's loop (:290-337) for the node (cw[0..d], key, gain) of pair `pl` at depth d: candidate q of its
    // open end; the child closes (:280-283: posted), is queued for the next level, or dies
    auto expand = [&](const uint32_t (&cw)[kLkMaxDepth + 1], const uint32_t d, const uint64_t key, const float gain, const uint32_t pl, const uint32_t q,
                      uint32_t *fnext, const uint32_t nxt) {
        const uint32_t t1 = cw[0] & 0xFFFFu;
        uint32_t t_open = 0;
#pragma unroll
        for (int t = 0; t < kLkMaxDepth + 1; ++t) t_open = (uint32_t)t == d ? cw[t] >> 16 : t_open;
        const uint32_t t_next = cand[t_open * k + q];
        const float g1 = gain - dcand[t_open * k + q];
        const uint32_t n_open = next[t_open];
        if (g1 <= kLkEps) return;                                   // :292-295
        const uint32_t t_break = next[t_next];
        if (n_open == t_next || t_break == t_open) return;          // is_tour_edge
        bool used = false;                                          // used[t_next] || used[t_break]
#pragma unroll
        for (int t = 0; t < kLkMaxDepth + 1; ++t) {
            if ((uint32_t)t <= d) {
                const uint32_t lo = cw[t] & 0xFFFFu, hi = cw[t] >> 16;
                used |= lo == t_next || hi == t_next || lo == t_break || hi == t_break;
            }
        }
        if (used) return;
        const float2 p_break = xy[t_break];
        const float g2 = g1 + dnext[t_next];
        const uint64_t ck = key | ((uint64_t)(q + 1u) << (64u - bits * (d + 1u)));
        if (g2 - dist(p_break, xy[t1]) > kLkEps) {                  // find_lk_chain(depth = d + 1) closes: :280-283
            atomicMin(&pairmin[pl], (unsigned long long)ck);
            return;
        }
        if (d + 1u >= D) return;                                    // :286-288
        const uint32_t slot = atomicAdd(&s_cnt[nxt], 1u);
        if (slot >= qcap) return;  // overflow: the count itself (> qcap) says so to everybody behind the level's barrier
        uint32_t *rec = fnext + (size_t)slot * RW;
        rec[0] = (uint32_t)ck;
        rec[1] = (uint32_t)(ck >> 32);
        rec[2] = __float_as_uint(g2);
        rec[3] = pl;
#pragma unroll
        for (int t = 0; t < kLkMaxDepth + 1; ++t)
            if ((uint32_t)t <= d) rec[kIlsHdr + t] = cw[t];
        rec[kIlsHdr + d + 1u] = t_next | (t_break << 16);
    };

    // find_lk_move (:345-389): true with the move's segment table in s_seg / s_nseg, its pair in s_bestpair, its length in s_clen
#ifdef TL_PROFILE_ILS
    uint64_t iq[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // cycles: chunk init, level 0, deeper levels, validation, segment table, apply + rebuild, pass end, levels run
    uint64_t it = __builtin_amdgcn_s_memtime();
#define TL_ISTAMP(k) do { const uint64_t t_ = __builtin_amdgcn_s_memtime(); iq[k] += t_ - it; it = t_; } while (0)
#else
#define TL_ISTAMP(k) do { } while (0)
#endif
    // task w of a level -> (node w / k, candidate w % k) without an integer division: k <= 16, w < 2^16
    const uint32_t kdiv = (1u << 20) / k + 1u;
    const uint32_t lane = tid & 63u, wave = tid >> 6;
    // find_lk_move (:345-389): true with the move's segment table in s_segw[s_win] / s_nsegw, its pair in s_bestpair, its length in s_clenw
    uint32_t chunk0 = 128u < (uint32_t)NT ? 128u : (uint32_t)NT;
    uint32_t hit_pair = 0u, hit_wave = 0u;  // the scan's answer: the pair, and the wave whose segment table is the move's
    auto scan = [&]() -> bool {
        uint32_t base = 0, chunk = chunk0;
        while (base < 2u * n) {
            const uint32_t CH = chunk < 2u * n - base ? chunk : 2u * n - base;
            for (uint32_t p = tid; p < CH; p += NT) pairmin[p] = ~0ull;
            if (tid == 0) {
                s_cnt[0] = 0u;
                s_cnt[1] = 0u;
            }
            TL_SYNC();
            TL_ISTAMP(0);
            // level 0: the pairs themselves
            for (uint32_t w = tid; w < CH * k; w += NT) {
                const uint32_t pl = (w * kdiv) >> 20, q = w - pl * k, idx = base + pl;
                const uint32_t t1 = ids[idx >> 1];
                const uint32_t t2 = (idx & 1u) ? prev[t1] : next[t1];  // :359 [next[t1], prev[t1]]
                uint32_t cw[kLkMaxDepth + 1];
#pragma unroll
                for (int t = 0; t < kLkMaxDepth + 1; ++t) cw[t] = 0u;
                cw[0] = t1 | (t2 << 16);
                const float g0 = (idx & 1u) ? dnext[t2] : dnext[t1];
                expand(cw, 0u, 0ull, g0, pl, q, fq0, 0u);
            }
            if (tid == 0) s_cnt[2] = 0u;  // the queue counters rotate (level L files into s_cnt[L % 3], resets s_cnt[(L + 2) % 3]): one barrier per level
            TL_SYNC();
            TL_ISTAMP(1);
            uint32_t d = 1;
            bool over = false;
            for (;;) {
                const uint32_t cslot = (d - 1u) % 3u, nslot = d % 3u;
                // (read behind the barrier that closed the level which filed into this slot: stable, the same for every thread — a
                //  separate overflow flag written by the level in progress would not be)
                const uint32_t cntp = s_cnt[cslot];
                over = cntp > qcap;
                if (over || cntp == 0u || d >= D) break;
                const uint32_t *fcur = ((d - 1u) & 1u) ? fq1 : fq0;
                uint32_t *fnext = ((d - 1u) & 1u) ? fq0 : fq1;
                for (uint32_t w = tid; w < cntp * k; w += NT) {
                    const uint32_t j = (w * kdiv) >> 20, q = w - j * k;
                    const uint32_t *rec = fcur + (size_t)j * RW;
                    const uint64_t key = (uint64_t)rec[0] | ((uint64_t)rec[1] << 32);
                    const uint32_t pl = rec[3];
                    if (key > pairmin[pl]) continue;  // behind the pair's first chain in depth-first order: the recursion never gets here
                    uint32_t cw[kLkMaxDepth + 1];
#pragma unroll
                    for (int t = 0; t < kLkMaxDepth + 1; ++t) cw[t] = (uint32_t)t <= d ? rec[kIlsHdr + t] : 0u;
                    expand(cw, d, key, __uint_as_float(rec[2]), pl, q, fnext, nslot);
                }
                if (tid == 0) s_cnt[(d + 1u) % 3u] = 0u;
                TL_SYNC();
                ++d;
#ifdef TL_PROFILE_ILS
                iq[7] += 1;
#endif
            }
            TL_ISTAMP(2);
            if (over) {  // a queue overflowed: the same pairs again in a smaller chunk (one pair always fits)
                chunk = CH > 1u ? CH / 2u : 1u;
                TL_SYNC();
                continue;
            }
            // The pairs' first chains (:373-381), lowest pair first: the chunk's pairs that hold a chain are numbered in pair order (a
            // ballot per wave, the waves' counts through LDS) and taken NW at a time, one WAVE per chain — lane e holds chain city e.
            // The chain is walked again from its key; removed edge m = (c[2m], c[2m+1]) sits behind position lo[m]; the arcs between the
            // removed edges are numbered by rank of lo; every chain city is the START or the END of an arc; the added edges (partner)
            // lead from arc to arc: one cycle <=> the walk from c[0] meets all k arcs (chain_is_valid_tour :181-250, what chain_valid_at
            // does per lane).  A valid chain's segment table (apply_lk_chain :397-450: the arcs in walk order from tour[0]) comes out of
            // the same tables.  The first batch with a valid chain holds the scan's answer: its lowest valid pair.
            constexpr uint32_t NW = (uint32_t)NT / 64u;
            {
                const bool has = tid < CH && pairmin[tid < CH ? tid : 0u] != ~0ull;
                const uint64_t bm = __builtin_amdgcn_ballot_w64(has);
                if (lane == 0u) {
                    s_wmask[wave] = bm;
                    s_wcnt[wave] = (uint32_t)__builtin_popcountll(bm);
                }
            }
            TL_SYNC();
            uint32_t tot = 0u;
            for (uint32_t w = 0; w < NW; ++w) tot += s_wcnt[w];
            uint32_t found_pair = 0xFFFFFFFFu, found_wave = 0u;
            for (uint32_t b0 = 0; b0 < tot && found_pair == 0xFFFFFFFFu; b0 += NW) {
                const uint32_t r = b0 + wave;  // this wave's chain: the r-th pair with one
                uint32_t myidx = 0xFFFFFFFFu;
                if (r < tot) {
                    uint32_t ws = 0u, rr = r;
                    while (rr >= s_wcnt[ws]) {
                        rr -= s_wcnt[ws];
                        ++ws;
                    }
                    uint64_t mm = s_wmask[ws];
                    for (uint32_t q = 0; q < rr; ++q) mm &= mm - 1ull;
                    const uint32_t pl = (ws << 6) + (uint32_t)__builtin_ffsll((long long)mm) - 1u, idx = base + pl;
                    const uint64_t key = pairmin[pl];
                    const uint32_t t1 = ids[idx >> 1];
                    uint32_t t_open = (idx & 1u) ? prev[t1] : next[t1];
                    uint32_t ce = lane == 0u ? t1 : t_open;  // lane e: chain city e
                    uint32_t clen = 2u;
                    for (uint32_t lv = 0; lv < D; ++lv) {
                        const uint32_t dg = (uint32_t)((key >> (64u - bits * (lv + 1u))) & ((1ull << bits) - 1ull));
                        if (dg == 0u) break;
                        const uint32_t t_next = cand[t_open * k + (dg - 1u)];
                        const uint32_t t_break = next[t_next];
                        ce = lane == clen ? t_next : (lane == clen + 1u ? t_break : ce);
                        t_open = t_break;
                        clen += 2u;
                    }
                    const uint32_t kk = clen >> 1;
                    const uint32_t P = pos[lane < clen ? ce : t1];
                    const uint32_t pu = (uint32_t)__shfl((int)P, (int)((2u * lane) & 63u)), pv = (uint32_t)__shfl((int)P, (int)((2u * lane + 1u) & 63u));
                    const uint32_t lo = ((pu + 1u == pv) || (pu == n - 1u && pv == 0u)) ? pu : pv;  // lanes < kk: removed edge m sits behind position lo
                    uint32_t rank = 0u;
                    for (uint32_t j = 0; j < kk; ++j) rank += readlane_u32(lo, j) < lo ? 1u : 0u;
                    const uint32_t lo_e = (uint32_t)__shfl((int)lo, (int)(lane >> 1)), rk_e = (uint32_t)__shfl((int)rank, (int)(lane >> 1));
                    const bool is_end = P == lo_e;  // the END of arc rank, else the START of the arc behind it
                    const uint32_t arc = is_end ? rk_e : (rk_e + 1u == kk ? 0u : rk_e + 1u);
                    uint32_t mate = 0u;  // the chain city at the other end of my arc
                    for (uint32_t j = 0; j < clen; ++j) mate = (readlane_u32(arc, j) == arc && j != lane) ? j : mate;
                    auto partner = [&](uint32_t t) -> uint32_t { return t == 0u ? clen - 1u : (t == clen - 1u ? 0u : ((t & 1u) ? t + 1u : t - 1u)); };
                    uint32_t ti = 0u, arcs = 0u;
                    bool ok = clen >= 4u;
                    if (ok) {
                        do {
                            const uint32_t tj = readlane_u32(mate, ti);
                            if (++arcs > kk) {
                                ok = false;
                                break;
                            }
                            ti = partner(tj);
                        } while (ti != 0u);
                        ok = ok && arcs == kk;
                    }
                    if (ok) {
                        // ---- valid: the segment table (lk_build_segments), by the same walk from tour position 0
                        const bool first_removed = __builtin_amdgcn_ballot_w64(lane < kk && lo == 0u) != 0ull;
                        int dir = first_removed ? -1 : +1;
                        // the first run stays inside arc 0: forward to its END, or (position 0 is its END) backward to its START
                        const uint64_t mb = __builtin_amdgcn_ballot_w64(lane < clen && arc == 0u && is_end == (dir > 0));
                        uint32_t eb = (uint32_t)__builtin_ffsll((long long)mb) - 1u;
                        uint32_t pp = 0u, emitted = 0u, nseg = 0u;
                        LkSeg *seg = s_segw[wave];
                        for (;;) {
                            const uint32_t endp = readlane_u32(P, eb);
                            uint32_t len = dir > 0 ? (endp >= pp ? endp - pp + 1u : endp + n - pp + 1u) : (pp >= endp ? pp - endp + 1u : pp + n - endp + 1u);
                            if (len > n - emitted) len = n - emitted;
                            if (lane == 0u) {
                                seg[nseg].src = pp;
                                seg[nseg].len = len;
                                seg[nseg].dst = emitted;
                                seg[nseg].dir = dir;
                            }
                            ++nseg;
                            emitted += len;
                            if (emitted >= n || nseg >= kLkMaxSeg) break;
                            const uint32_t e = partner(eb);
                            pp = readlane_u32(P, e);
                            dir = readlane_u32(is_end ? 1u : 0u, e) ? -1 : +1;
                            eb = readlane_u32(mate, e);
                        }
                        if (lane == 0u) {
                            s_nsegw[wave] = nseg;
                            s_clenw[wave] = clen;
                        }
                        myidx = idx;
                    }
                }
                if (lane == 0u) s_widx[wave] = myidx;
                TL_SYNC();
                for (uint32_t w = 0; w < NW; ++w) {
                    const uint32_t v = s_widx[w];
                    if (v < found_pair) {
                        found_pair = v;
                        found_wave = w;
                    }
                }
                TL_SYNC();  // (s_widx is written again by the next batch)
            }
            TL_ISTAMP(3);
            if (found_pair != 0xFFFFFFFFu) {
                hit_pair = found_pair;
                hit_wave = found_wave;
                chunk0 = found_pair + 64u < 128u ? 128u : found_pair + 64u;  // the next hit is most often near this one
                chunk0 = chunk0 > (uint32_t)NT ? (uint32_t)NT : chunk0;
                TL_ISTAMP(4);
                return true;
            }
            base += CH;
            chunk = CH * 2u < (uint32_t)NT ? CH * 2u : (uint32_t)NT;
        }
        chunk0 = 128u < (uint32_t)NT ? 128u : (uint32_t)NT;
        return false;
    };


    // apply_lk_chain (:397-450) of the scan's move: the old tour's arcs copied in walk order from tour[0]
    auto apply_move = [&]() {
        const uint32_t win = hit_wave, nseg = s_nsegw[win];
        const LkSeg *s_seg = s_segw[win];
        for (uint32_t r = tid; r < n; r += NT) {
            uint32_t sidx = 0;
            while (sidx + 1u < nseg && r >= s_seg[sidx + 1u].dst) ++sidx;
            const LkSeg sg = s_seg[sidx];
            const uint32_t t = r - sg.dst;
            uint32_t spos = sg.dir > 0 ? sg.src + t : sg.src + n - t;
            if (spos >= n) spos -= n;
            alt[r] = tour[spos];
        }
        searches += (uint64_t)hit_pair + 1u;
        ++moves;
        exchanged += (uint64_t)(s_clenw[win] / 2u);
        TL_SYNC();
        uint16_t *const tsw = tour;
        tour = alt;
        alt = tsw;
        rebuild();
    };
    // double_bridge (:485-499) of the best tour with the draws of epoch e, then a fresh pass: city_ids = tour (:466)
    auto kick_from_best = [&](uint64_t dr) {
        if (n < 8u) {
            for (uint32_t r = tid; r < n; r += NT) tour[r] = best[r];
        } else {
            const uint32_t qn = n / 4u;
            const uint32_t r1 = (uint32_t)(splitmix64_at(G.seed, dr) % qn), r2 = (uint32_t)(splitmix64_at(G.seed, dr + 1) % qn),
                           r3 = (uint32_t)(splitmix64_at(G.seed, dr + 2) % qn);
            const uint32_t p1 = 1u + r1, p2 = p1 + 1u + r2, p3 = p2 + 1u + r3;
            for (uint32_t w = tid; w < n; w += NT) {
                uint32_t src;
                if (w < p1) src = w;
                else if (w < p1 + (p3 - p2)) src = p2 + (w - p1);
                else if (w < p3) src = p1 + (w - p1 - (p3 - p2));
                else src = w;
                tour[w] = best[src];
            }
        }
        TL_SYNC();
        for (uint32_t r = tid; r < n; r += NT) ids[r] = tour[r];
        rebuild();
    };

    if (mode == 2u) {
        // ---- one epoch (:75-96 up to the comparison): kick, lk_pass, tour_distance; the verdict is k_lk_ils_commit's
        // A speculative epoch has a budget of scans.  The reference's lk_pass can cycle — a chain's gain is a sum of rounded f32
        // terms, and on instances with many equal distances a few moves of "gain" 1e-5 lead back to the same tour (a280: the kick of
        // epoch 170 from the best tour of epoch 10, found by this very kernel; the oracle does not return from it either).  The
        // sequential loop never meets such an epoch when an earlier one is accepted first, so a speculative one must not hang
        // the launch: out of budget it files "unfinished" and the verdict stops in front of it; only an epoch that is really next
        // (slot 0) is run again with a larger budget (S->window counts the doublings).
        const uint32_t e = S->epoch + slot;
        const uint32_t lvl = S->window < 6u ? S->window : 6u;
        const uint64_t cap = lk_pass_cap(n);
        const uint64_t ep_budget = ((uint64_t)G.ils_slice << (3u * lvl)) < cap ? ((uint64_t)G.ils_slice << (3u * lvl)) : cap;
        scans = searches = moves = exchanged = 0ull;
        kick_from_best(3ull * e);
        bool unfinished = false;
        for (;;) {
            if (scans >= ep_budget) {
                unfinished = true;
                break;
            }
            const bool found = scan();
            ++scans;
            if (!found) break;
            apply_move();
        }
        searches += 2ull * n;
        const float dcur = tour_distance(tour);
        for (uint32_t r = tid; r < n; r += NT) G.ils_ep_tour[(size_t)slot * n + r] = tour[r];
        if (tid == 0) {
            G.ils_ep_dist[slot] = dcur;
            uint64_t *cn = G.ils_ep_cnt + (size_t)slot * 4;
            cn[0] = unfinished ? ~0ull : scans;
            cn[1] = searches;
            cn[2] = moves;
            cn[3] = exchanged;
        }
        return;
    }

    // ---- lin_kernighan::solve (:35-100), from wherever the previous slice stopped
    rebuild();
    uint32_t budget = G.ils_slice;
    bool fin = S->finished != 0u;
    uint64_t pass_scans = S->pass_moves;  // (a pass may span slices)
    bool cycling = false;
    while (!fin && budget != 0u) {
        --budget;
        const bool found = scan();
        ++scans;
        if (found) {
            apply_move();
            TL_ISTAMP(5);
            if (++pass_scans > lk_pass_cap(n)) {  // a pass longer than this is a cycle — see the epochs' budget
                cycling = true;
                break;
            }
            continue;
        }
        pass_scans = 0;
        searches += 2ull * n;
        // ---- this lk_pass is over (:472-473)
        bool kick;
        if (stage == 0u) {                          // the initial pass (:61-70)
            for (uint32_t r = tid; r < n; r += NT) best[r] = tour[r];
            TL_SYNC();
            best_dist = tour_distance(best);
            stage = 1u;
            epoch = 0u;
            platoo = 0u;
            snapshot(tour, best_dist);
            kick = G.epochs > 0u;
            if (mode == 1u) {  // the epochs are the speculative launches' (mode 2)
                fin = !kick;
                break;
            }
        } else {                                    // an epoch's pass (:85-96)
            const float dcur = tour_distance(tour);
            bool stop = false;
            if (dcur < best_dist) {
                for (uint32_t r = tid; r < n; r += NT) best[r] = tour[r];
                snapshot(tour, dcur);
                best_dist = dcur;
                platoo = 0u;
            } else if (++platoo >= G.platoo_epochs) {
                stop = true;
            }
            ++epoch;
            kick = !stop && epoch < G.epochs;
        }
        TL_SYNC();
        if (!kick) {
            fin = true;
            break;
        }
        kick_from_best(draws);
        if (n >= 8u) draws += 3;
        TL_ISTAMP(6);
        // a full ring of undelivered snapshots ends the slice: the host drains it and starts the next one
        if (G.snap && G.snap_ring && snaps - G.snap_delivered >= G.snap_cap) break;
    }
#ifdef TL_PROFILE_ILS
    if (tid == 0)
        printf("k_lk_ils n=%u: scans %lu; cycles per scan: chunk init %lu, level 0 %lu, deeper levels %lu (%lu levels per 100 scans), validation %lu, segment table %lu, apply + rebuild %lu, pass end %lu\n",
               n, (unsigned long)(scans - S->scans), (unsigned long)(iq[0] / (scans - S->scans + 1)), (unsigned long)(iq[1] / (scans - S->scans + 1)), (unsigned long)(iq[2] / (scans - S->scans + 1)),
               (unsigned long)(iq[7] * 100 / (scans - S->scans + 1)), (unsigned long)(iq[3] / (scans - S->scans + 1)), (unsigned long)(iq[4] / (scans - S->scans + 1)),
               (unsigned long)(iq[5] / (scans - S->scans + 1)), (unsigned long)(iq[6] / (scans - S->scans + 1)));
#endif
    // ---- the state goes back to HBM (the next slice, or the result)
    TL_SYNC();
    for (uint32_t r = tid; r < n; r += NT) {
        G.tour[r] = tour[r];
        G.city_ids[r] = ids[r];
        G.best[r] = best[r];
    }
    if (tid == 0) {
        S->applied = mode == 1u ? 16u : 1u;  // (mode 1 hands over to the epochs: their first batch is 16 wide)
        S->pass_moves = (uint32_t)pass_scans;
        S->finished = fin ? 1u : (cycling ? 2u : 0u);
        S->stage = stage;
        S->epoch = epoch;
        S->platoo = platoo;
        S->best_dist = best_dist;
        S->draws = draws;
        S->scans = scans;
        S->searches = searches;
        S->moves = moves;
        S->exchanged = exchanged;
        S->snaps = snaps;
    }
}

}  // namespace

namespace {
// The verdict over a batch of speculative epochs (k_lk_ils mode 2), in epoch order, exactly as the sequential loop gives it
// (lin_kernighan.rs:75-97): an epoch's counters count; shorter than the best -> it becomes the best (PathUpdate), the plateau
// count restarts, and the batch ends here — the later epochs of the batch were kicked from a tour that is no longer the best and
// are run again; not shorter -> the plateau count grows and may end the search.
__global__ __launch_bounds__(256) void k_lk_ils_commit(LkArgs G)
{
    __shared__ uint32_t s_acc;
    __shared__ float s_d;
    LkState *S = G.state;
    if (S->finished) return;
    const uint32_t tid = threadIdx.x, n = G.n;
    const uint32_t width = S->applied < G.ils_P ? S->applied : G.ils_P;
    const uint32_t left = G.epochs - S->epoch, P = width < left ? width : left;
    if (tid == 0) {
        uint32_t epoch = S->epoch, platoo = S->platoo, acc = 0xFFFFFFFFu;
        float best_dist = S->best_dist;
        uint64_t scans = S->scans, searches = S->searches, moves = S->moves, exchanged = S->exchanged;
        bool fin = false;
        uint32_t window = S->window;
        for (uint32_t j = 0; j < P; ++j) {
            const uint64_t *cn = G.ils_ep_cnt + (size_t)j * 4;
            if (cn[0] == ~0ull) {  // out of its scan budget: the verdict stops in front of it; if it is the next epoch it runs again, longer
                if (j == 0u) {  // the next epoch: again with 8 x the budget — or, if that was lk_pass_cap already, it does not terminate
                    const uint32_t lvl = window < 6u ? window : 6u;
                    if (((uint64_t)G.ils_slice << (3u * lvl)) >= lk_pass_cap(n)) S->key = 0xDEAD0001u;
                    ++window;
                }
                break;
            }
            if (j == 0u) window = 0u;
            scans += cn[0];
            searches += cn[1];
            moves += cn[2];
            exchanged += cn[3];
            const float d = G.ils_ep_dist[j];
            bool stop = false;
            if (d < best_dist) {      // :86
                best_dist = d;
                platoo = 0u;
                acc = j;
            } else if (++platoo >= G.platoo_epochs) {  // :92-95
                stop = true;
            }
            ++epoch;
            fin = stop || epoch >= G.epochs;
            if (fin || acc != 0xFFFFFFFFu) break;
        }
        S->epoch = epoch;
        S->platoo = platoo;
        S->best_dist = best_dist;
        S->draws = n >= 8u ? 3ull * epoch : 0ull;
        S->scans = scans;
        S->searches = searches;
        S->moves = moves;
        S->exchanged = exchanged;
        S->finished = fin ? 1u : (S->key == 0xDEAD0001u ? 2u : 0u);
        S->window = window;
        // the next batch's width: decided at epoch j -> a few times j; a batch that was rejected throughout -> four times as wide
        {
            uint32_t w2 = acc != 0xFFFFFFFFu ? 4u * (acc + 1u) + 8u : 4u * width;
            w2 = w2 < 16u ? 16u : w2;
            S->applied = w2 > G.ils_P ? G.ils_P : w2;
        }
        s_acc = acc;
        s_d = best_dist;
#ifdef TL_DEBUG_ILS
        printf("commit: P %u -> epoch %u platoo %u acc %u best %.5f scans %lu fin %d\n", P, epoch, platoo, acc, (double)best_dist, (unsigned long)scans, (int)fin);
#endif
    }
    TL_SYNC();
    const uint32_t acc = s_acc;
    if (acc == 0xFFFFFFFFu) return;
    const uint32_t *src = G.ils_ep_tour + (size_t)acc * n;
    const uint32_t sidx = S->snaps;
    const bool keep = G.snap && (G.snap_ring || sidx < G.snap_cap);
    const uint32_t at = G.snap_ring ? sidx % (G.snap_cap ? G.snap_cap : 1u) : sidx;
    for (uint32_t r = tid; r < n; r += 256u) {
        const uint32_t c = src[r];
        G.best[r] = c;
        if (keep) G.snap[(size_t)at * n + r] = c;  // :90 send_progress(best_tour, best_dist)
    }
    TL_SYNC();
    if (tid == 0) {
        if (keep) G.snap_dist[at] = s_d;
        S->snaps = sidx + 1u;
    }
}
}  // namespace

hipError_t launch_lk_ils_commit(const LkArgs &G, hipStream_t s)
{
    hipLaunchKernelGGL(k_lk_ils_commit, dim3(1), dim3(256), 0, s, G);
    return hipGetLastError();
}

hipError_t launch_lk_ils(const LkArgs &G, hipStream_t s)
{
    const size_t lds = ils_fixed_bytes(G.n, G.k) + 2 * (size_t)G.ils_qcap * (kIlsHdr + G.max_depth) * 4;
    auto go = [&](auto kern, int nt) -> hipError_t {
        hipError_t e = allow_max_lds(reinterpret_cast<const void *>(kern));
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3(G.ils_mode == 2u ? G.ils_P : 1u), dim3(nt), lds, s, G);
        return hipGetLastError();
    };
    // a level of a small instance is a few hundred lanes: four waves meet at a barrier faster than sixteen
    const uint32_t nt = G.ils_threads ? G.ils_threads : (G.n <= kIlsSmallN ? 256u : 1024u);
    if (nt == 256u) return go(k_lk_ils<256>, 256);
    if (nt == 512u) return go(k_lk_ils<512>, 512);
    return go(k_lk_ils<kIlsNT>, kIlsNT);
}

// after a move: tour = alt, rank / successor / predecessor of every city — spread over the chip
__global__ __launch_bounds__(256) void k_lk_rebuild(LkArgs G)
{
    LkState *S = G.state;
    if (S->finished || !S->applied) return;
    const uint32_t n = G.n, r = blockIdx.x * 256u + threadIdx.x;
    if (r >= n) return;
    const uint32_t c = G.alt[r];
    G.tour[r] = c;
    G.pos[c] = r;
    G.next[c] = G.alt[r + 1u == n ? 0u : r + 1u];
    G.prev[c] = G.alt[r == 0u ? n - 1u : r - 1u];
}

// first lk_pass of solve(): city_ids = tour, next/prev/pos from tour, state reset
__global__ __launch_bounds__(kLkNT) void k_lk_begin(LkArgs G)
{
    const uint32_t tid = threadIdx.x, n = G.n;
    for (uint32_t r = tid; r < n; r += kLkNT) {
        const uint32_t c = G.tour[r], cn = G.tour[r + 1u == n ? 0u : r + 1u];
        G.city_ids[r] = c;
        G.best[r] = c;
        G.pos[c] = r;
        G.next[c] = cn;
        G.prev[c] = G.tour[r == 0u ? n - 1u : r - 1u];
        if (G.nx) {
            const float2 pn = G.xy[cn];
            G.nx[c] = make_float4(__uint_as_float(cn), pn.x, pn.y, dist(G.xy[c], pn));
        }
    }
    if (G.candd) {  // the candidates with their distances, once per call
        for (size_t e = tid; e < (size_t)n * G.k; e += kLkNT) {
            const uint32_t c = G.cand[e];
            G.candd[e] = make_uint2(c, __float_as_uint(dist(G.xy[e / G.k], G.xy[c])));
        }
    }
    if (tid == 0) {
        LkState *S = G.state;
        S->snaps = 0u;
        S->key = 0xFFFFFFFFu;
        S->key2[0] = S->key2[1] = 0xFFFFFFFFu;
        S->flip = S->flip_next = 0u;
        S->finished = n < 4 ? 1u : 0u;  // :57-59
        S->stage = 0;
        S->epoch = 0;
        S->platoo = 0;
        S->best_dist = 0.0f;
        S->draws = 0;
        S->scans = S->searches = S->moves = S->exchanged = 0;
        S->window = kLkWindowFirst < 2u * n ? kLkWindowFirst : 2u * n;
        S->applied = 0u;
        S->pass_moves = 0u;
    }
}

hipError_t launch_lk_begin(const LkArgs &G, hipStream_t s)
{
    hipLaunchKernelGGL(k_lk_begin, dim3(1), dim3(kLkNT), 0, s, G);
    return hipGetLastError();
}

hipError_t launch_lk_round(const LkArgs &G0, hipStream_t s, uint32_t round)
{
    LkArgs G = G0;
    G.parity = round & 1u;
    if (G.pairmin) {  // split scan
        const uint64_t lanes = (uint64_t)2u * G.n * G.k * (G.k + 1u) * (G.split_levels == 3u ? G.k + 1u : 1u);
        const uint32_t k1 = G.k + 1u, per_pair = lk_subs(G);
#ifdef TL_TUNE
        if (per_pair <= 1024u && G.persist_blocks && G.fused_pick && G.split_levels == 3u) {
            // persistent form: a fixed grid strides over the window's pairs (k_lk_scan_persist)
            const uint32_t grid = G.persist_blocks < 2u * G.n ? G.persist_blocks : 2u * G.n;
            hipLaunchKernelGGL(k_lk_scan_persist, dim3(grid), dim3(k1, k1, G.k), 0, s, G);
        } else
#endif
        if (per_pair <= 1024u) {  // one workgroup per pair, thread coordinates = sub-search digits
            const dim3 blk = G.split_levels == 3u ? dim3(k1, k1, G.k) : dim3(k1, G.k, 1);
            if (G.nx && G.fused_pick && G.split_levels == 3u) hipLaunchKernelGGL((k_lk_scan_sub<true, true>), dim3(2u * G.n), blk, 0, s, G);
            else hipLaunchKernelGGL(k_lk_scan_sub<true>, dim3(2u * G.n), blk, 0, s, G);
        } else {
            hipLaunchKernelGGL(k_lk_scan_sub<false>, dim3((uint32_t)((lanes + 255u) / 256u)), dim3(256), 0, s, G);
        }
        if (!(G.fused_pick && per_pair <= 1024u)) hipLaunchKernelGGL(k_lk_scan_pick, dim3((2u * G.n + 255u) / 256u), dim3(256), 0, s, G);
    } else {
        const size_t lds = (size_t)G.n * 10;
        if (G.n < 65536u && lds <= (size_t)G.lds_budget) {
            hipError_t e = allow_max_lds(reinterpret_cast<const void *>(k_lk_scan<true>));
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(k_lk_scan<true>, dim3((2u * G.n + 255u) / 256u), dim3(256), lds, s, G);
        } else {
            hipLaunchKernelGGL(k_lk_scan<false>, dim3((2u * G.n + 255u) / 256u), dim3(256), 0, s, G);
        }
    }
    if (G.chip_step) {
        hipLaunchKernelGGL(k_lk_control<true>, dim3((G.n + kLkNT - 1u) / kLkNT), dim3(kLkNT), 0, s, G);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(k_lk_control<false>, dim3(1), dim3(kLkNT), 0, s, G);
    if (G.n >= kLkRebuildSplitN) hipLaunchKernelGGL(k_lk_rebuild, dim3((G.n + 255u) / 256u), dim3(256), 0, s, G);
    return hipGetLastError();
}

size_t lk_chain_slot_words() { return (size_t)kLkSlot; }
size_t lk_sub_slot_words() { return (size_t)kLkSubSlot; }
uint32_t lk_max_depth() { return (uint32_t)kLkMaxDepth; }

size_t lk_small_lds_bytes(uint32_t n, uint32_t k) { return (size_t)n * (36u + 4u * (size_t)k) + 16u; }

hipError_t launch_lk_solve(const LkArgs &G, hipStream_t s, bool small, int threads)
{
    if (!small) {
        hipLaunchKernelGGL((k_lk_solve<kLkNT, false>), dim3(1), dim3(kLkNT), 0, s, G);
        return hipGetLastError();
    }
#ifndef TL_TUNE  // the LDS-resident form loses to the chip-wide scans at every size: tuning build only
    (void)threads;
    return hipErrorInvalidValue;
#else
    const size_t lds = lk_small_lds_bytes(G.n, G.k);
    auto go = [&](auto kern, int nt) -> hipError_t {
        hipError_t e = allow_max_lds(reinterpret_cast<const void *>(kern));
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3(1), dim3(nt), lds, s, G);
        return hipGetLastError();
    };
    (void)threads;  // 64 / 256 / 1024 threads measured (scripts/timing_lk.py): 256 is the best of the three up to n ~ 300
    return go(k_lk_solve<256, true>, 256);
#endif
}

}  // namespace TL_LK_NS
