// two_opt_common.h — device helpers shared by the 2-opt kernels: wave reductions on integer keys (DPP), the per-tile
// bounding-box metadata of the exact L0 bound and the f32 lower bound itself (DESIGN.md "Exact decision cascade").
#pragma once
#include "tl_device.h"

#pragma clang fp contract(off)

namespace tl {

// ---- tour-ordered points.  The kernels take them either as plain float2 (8 B per city) or, for the LDS descent kernel, as
// grid coordinates (5 B per city): TSPLIB-style inputs lie on a decimal grid — x = k / S with an integer k < 2^20 — and the f32
// coordinate is recovered EXACTLY as fl32(fl64(k) * fl64(1/S)).  "Exactly" is not argued but checked: k_fx_encode runs this
// very decode on every coordinate of the instance before the form is chosen (two_opt_ref.hip).  7 B per city with the u16 id
// is what lets two tours of n = 10^4 share one CU's LDS.
struct PtsFx {
    uint32_t *lo;   // x (20 bits) | y's low 12 bits << 20
    uint8_t *hi;    // y's high 8 bits
    double inv;     // fl64(1 / S)
    struct Raw {
        uint32_t lo, hi;
    };
};
__device__ __forceinline__ float2 fx_decode(uint32_t lo, uint32_t hi, double inv)
{
    const uint32_t kx = lo & 0xFFFFFu, ky = (lo >> 20) | (hi << 12);
    return make_float2((float)((double)kx * inv), (float)((double)ky * inv));
}
__device__ __forceinline__ float2 pt_get(const float2 *P, uint32_t k) { return P[k]; }
__device__ __forceinline__ float2 pt_get(const PtsFx &P, uint32_t k) { return fx_decode(P.lo[k], P.hi[k], P.inv); }
__device__ __forceinline__ float2 pt_raw(const float2 *P, uint32_t k) { return P[k]; }
__device__ __forceinline__ PtsFx::Raw pt_raw(const PtsFx &P, uint32_t k) { return PtsFx::Raw{P.lo[k], P.hi[k]}; }
__device__ __forceinline__ void pt_put(float2 *P, uint32_t k, float2 r) { P[k] = r; }
__device__ __forceinline__ void pt_put(const PtsFx &P, uint32_t k, PtsFx::Raw r)
{
    P.lo[k] = r.lo;
    P.hi[k] = (uint8_t)r.hi;
}

// ---- whole-wave maximum of u32 keys in 6 DPP steps, result in lane 63 (no LDS, no readlane): row_shr 1, 2, 4, 8 leave each row's
// maximum in its lane 15 (max is idempotent, so the overlapping windows do no harm), row_bcast15 / row_bcast31 carry it on to the
// rows above.  Lanes without a DPP source read 0, the identity of an unsigned maximum.
template <int CTRL, int ROWMASK>
__device__ __forceinline__ uint32_t dpp_max_step(uint32_t v)
{
    const uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROWMASK, 0xf, true);
    return v > o ? v : o;
}
__device__ __forceinline__ uint32_t wave_max_key_lane63(uint32_t v)
{
    v = dpp_max_step<0x111, 0xf>(v);
    v = dpp_max_step<0x112, 0xf>(v);
    v = dpp_max_step<0x114, 0xf>(v);
    v = dpp_max_step<0x118, 0xf>(v);
    v = dpp_max_step<0x142, 0xa>(v);  // row_bcast15 into rows 1 and 3
    v = dpp_max_step<0x143, 0xc>(v);  // row_bcast31 into rows 2 and 3
    return v;
}
// f32 <-> u32 key with the same order (negative floats: all bits flipped; others: sign bit set)
__device__ __forceinline__ uint32_t fkey(float f)
{
    const uint32_t b = __builtin_bit_cast(uint32_t, f);
    return b ^ ((uint32_t)((int32_t)b >> 31) | 0x80000000u);
}
__device__ __forceinline__ float fkey_inv(uint32_t k)
{
    return __builtin_bit_cast(float, k ^ ((k & 0x80000000u) ? 0x80000000u : 0xFFFFFFFFu));
}

// L0 metadata of tile t: bounding box of P[64t .. 64t+64] over the positions that take part in a
// candidate (j <= n-2 as c, j+1 as e) and the largest squared tour-edge sq(P[j],P[j+1]) in it.
// Five wave reductions on order-preserving integer keys (minima as maxima of the inverted key), each 6 DPP-fused v_max_u32;
// lane 63 ends up with all five and stores them.  (The float form — v_min/v_max_f32 with their NaN canonicalisation, four
// readlanes and three scalar min/max per reduction — was ~150 instructions per tile; this is ~70.)  A NaN coordinate is left out
// of the box as before (v_min/v_max_f32 return the other operand).
template <typename PT>
__device__ __forceinline__ void build_tile_meta(const PT &P, uint32_t n, uint32_t t, int lane, float4 *tbox, float *tmsq)
{
    const uint32_t j = (t << 6) + (uint32_t)lane;
    const bool valid = j + 2u <= n;  // j <= n-2
    const float2 c = pt_get(P, j), e = pt_get(P, j + 1u);
    const float inf = __builtin_inff();
    // invalid lanes: +inf for the minima, -inf for the maxima, 0 for the largest edge (all-invalid tile: an empty box, never live)
    const uint32_t kmnx = ~fkey(valid ? fminf(fminf(c.x, e.x), inf) : inf), kmny = ~fkey(valid ? fminf(fminf(c.y, e.y), inf) : inf);
    const uint32_t kmxx = fkey(valid ? fmaxf(fmaxf(c.x, e.x), -inf) : -inf), kmxy = fkey(valid ? fmaxf(fmaxf(c.y, e.y), -inf) : -inf);
    const float sq = sqdist(c, e);
    const uint32_t kmsq = (valid && sq == sq) ? __builtin_bit_cast(uint32_t, sq) : 0u;  // squares are >= +0: their bits order like u32
    const uint32_t rmnx = wave_max_key_lane63(kmnx), rmny = wave_max_key_lane63(kmny);
    const uint32_t rmxx = wave_max_key_lane63(kmxx), rmxy = wave_max_key_lane63(kmxy);
    const uint32_t rmsq = wave_max_key_lane63(kmsq);
    if (lane == 63) {
        tbox[t] = make_float4(fkey_inv(~rmnx), fkey_inv(~rmny), fkey_inv(rmxx), fkey_inv(rmxy));
        tmsq[t] = __builtin_bit_cast(float, rmsq);
    }
}

// f32 lower bound of sqdist(p, q) over every q inside the box (monotone ops only -> valid in f32).
// d = max(lo - p, p - hi, 0) is taken on the bit patterns as signed ints: negative floats are negative
// ints and non-negative floats order like ints, so one v_max3_i32 does it without NaN canonicalisation.
// Written on (x, y) pairs — lo - p and p - hi are the two halves of the float4 as it is loaded — so that it becomes two
// v_pk_add_f32, two v_max3_i32, one v_pk_mul_f32 and one add with no operand shuffling (paired the other way, a's and b's bounds
// together, every box component is needed twice: 36 v_mov per pruned step and wave).
__device__ __forceinline__ float box_lb(float px, float py, float4 box)
{
    const v2f p = {px, py}, lo = {box.x, box.y}, hi = {box.z, box.w};
    const v2f u = lo - p, v = p - hi;
    // (through float temporaries: __builtin_bit_cast applied to an ext-vector ELEMENT, `bit_cast(int, u.y)`, reads element 0 with
    //  hipcc 7.2 — the y half of the bound silently became a copy of the x half)
    const float uxf = u.x, uyf = u.y, vxf = v.x, vyf = v.y;
    const int ux = __builtin_bit_cast(int, uxf), vx = __builtin_bit_cast(int, vxf);
    const int uy = __builtin_bit_cast(int, uyf), vy = __builtin_bit_cast(int, vyf);
    const int mx = ux > vx ? ux : vx, my = uy > vy ? uy : vy;
    v2f d = {__builtin_bit_cast(float, mx > 0 ? mx : 0), __builtin_bit_cast(float, my > 0 ? my : 0)};
    d = d * d;
    return add_f32_nopack(d.x, d.y);
}

#ifndef TL_DENSE_LEAD
#define TL_DENSE_LEAD 4
#endif
static constexpr uint32_t kDenseLead = TL_DENSE_LEAD;  // waves active in round 1 of a dense step
#ifndef TL_MAX_CHAIN
#define TL_MAX_CHAIN 16
#endif
static constexpr uint32_t kMaxChainHits = TL_MAX_CHAIN;  // hits one wave may chain inside its tile before handing back

// Work the cascade really does, counted per wave with wave-uniform (SALU) adds: candidates (lanes) that entered L1, that
// survived L1 into L2, that needed the exact L3, and tile bounds evaluated by L0 (lanes = tiles).  bench.py reports them
// as "candidates touched" next to the algorithmic candidate count.
struct TileCounts {
    uint32_t l0 = 0, l1 = 0, l2 = 0, l3 = 0;
    uint32_t prow = 0, ptile = 0;  // pruned mode: rows bounded by L0, tile passes those rows ran (live tiles)
};
// The timed kernels carry NoCounts: the four live SGPR counters cost the LDS descent kernel 8 % (it is SGPR-bound: 22 -> 44
// spilled SGPRs), so counting is a separate instantiation that bench.py launches once, untimed (TL_FLAG_COUNT_WORK).
struct NoCounts {
    struct Sink {
        __device__ __forceinline__ void operator+=(uint32_t) {}
    } l0, l1, l2, l3, prow, ptile;
};

// The improving columns of row (a, b) inside one 64-wide j tile (lane l holds c = P[tb+l], e = P[tb+l+1]) as a lane mask,
// decided by the L1 -> L2 -> L3 cascade.  Straight-line code, every branch wave-uniform, the result in SGPRs.
// sqab = sq(a, b) is a row constant the caller already holds.
// (The logic runs on 64-bit lane masks — every ballot below is taken of a plain comparison, i.e. is the v_cmp itself, and the
// and / or / not are scalar instructions; a ballot of a boolean expression costs a v_cndmask + v_cmp_ne pair on top.)
// (tile_mask_terms: the same with the two squares that do not depend on the row's b — s1 = sq(a, c), sqce = sq(c, e) — given: a
//  wave that chains hits inside a tile decides the tile again and again against a new b and keeps them in registers)
// (ROOTS: the chain's copy — the approximate roots of s1, sqce and sqab come with them: r1 and rce were taken once when the chain
//  began, and the new sqab's root is r1 of the hit lane)
template <bool PRUNE, bool ROOTS = false, typename TC>
__device__ __forceinline__ uint64_t tile_mask_terms(float s1, float sqce, float2 e, uint32_t j, uint32_t n, uint32_t jmin,
                                                    float bx, float by, float sqab, TC &tc, float r1 = 0.0f, float rce = 0.0f, float rab = 0.0f)
{
    const float s2 = sqdist(make_float2(bx, by), e);
    const uint64_t m_rng = __builtin_amdgcn_ballot_w64((j - jmin) <= (n - 2u - jmin));  // jmin <= j <= n-2 in one unsigned compare (callers keep jmin <= n-2)
    if (PRUNE) {
        tc.l1 += (uint32_t)__builtin_popcountll(m_rng);
        const uint64_t m1 = m_rng & (__builtin_amdgcn_ballot_w64(s1 < sqab) | __builtin_amdgcn_ballot_w64(s2 < sqce));  // L1
        if (!m1) return 0;                                               // the common case late in a sweep
        tc.l2 += (uint32_t)__builtin_popcountll(m1);
        const float neu_a = (ROOTS ? r1 : __builtin_amdgcn_sqrtf(s1)) + __builtin_amdgcn_sqrtf(s2);  // L2
        const float cur_a = (ROOTS ? rab : __builtin_amdgcn_sqrtf(sqab)) + (ROOTS ? rce : __builtin_amdgcn_sqrtf(sqce));
        const float margin = cur_a * 1.9073486e-6f;                      // 2^-19
        uint64_t m_imp = m1 & __builtin_amdgcn_ballot_w64(neu_a < cur_a - margin);
        // near-ties, tiny squares (v_sqrt_f32 loses accuracy on denormals) and non-finite sums go to L3.  Squares are
        // >= +0, so their bit patterns order like unsigned ints (a NaN compares high and is caught through cur_a, or
        // makes neu_a NaN, which L2 and the reference both read as "not improving").
        const uint32_t smin = min(min(__builtin_bit_cast(uint32_t, s1), __builtin_bit_cast(uint32_t, sqce)),
                                  min(__builtin_bit_cast(uint32_t, s2), __builtin_bit_cast(uint32_t, sqab)));
        const uint64_t mt = (m1 & ~m_imp) & (__builtin_amdgcn_ballot_w64(neu_a <= cur_a + margin) | __builtin_amdgcn_ballot_w64(smin < 0x0DA24260u /* 1e-30f */) |
                                             ~__builtin_amdgcn_ballot_w64(cur_a < 3.0e38f));
        if (mt) {                                                        // L3
            tc.l3 += (uint32_t)__builtin_popcountll(mt);
            // the opaque copies keep the compiler from hoisting the loop-invariant exact roots into a tile or row
            // prologue, where every tile (row) would pay ~40 VALU for a path that almost never runs
            float s1v = s1, scev = sqce, sabv = sqab;
            asm volatile("" : "+v"(s1v), "+v"(scev), "+v"(sabv));
            const float neu = sqrt_rn(s1v) + sqrt_rn(s2);
            const float cur = sqrt_rn(sabv) + sqrt_rn(scev);
            m_imp = (m_imp & ~mt) | (mt & __builtin_amdgcn_ballot_w64(neu < cur));
        }
        return m_imp;
    } else {
        tc.l3 += (uint32_t)__builtin_popcountll(m_rng);
        const float neu = sqrt_rn(s1) + sqrt_rn(s2);
        const float cur = sqrt_rn(sqab) + sqrt_rn(sqce);
        return m_rng & __builtin_amdgcn_ballot_w64(neu < cur);  // two_opt.rs:35-49
    }
}

template <bool PRUNE, typename TC>
__device__ __forceinline__ uint64_t tile_mask_core(float2 c, float2 e, uint32_t j, uint32_t n, uint32_t jmin,
                                                   float ax, float ay, float bx, float by, float sqab, TC &tc)
{
    return tile_mask_terms<PRUNE>(sqdist(make_float2(ax, ay), c), sqdist(c, e), e, j, n, jmin, bx, by, sqab, tc);
}

template <bool PRUNE, typename TC, typename PT>
__device__ __forceinline__ uint64_t tile_improving_mask(const PT &P, uint32_t n, uint32_t tb, uint32_t jmin,
                                                        float ax, float ay, float bx, float by, float sqab, int lane,
                                                        TC &tc)
{
    const uint32_t j = tb + (uint32_t)lane;
    return tile_mask_core<PRUNE>(pt_get(P, j), pt_get(P, j + 1u), j, n, jmin, ax, ay, bx, by, sqab, tc);
}

template <bool PRUNE, typename PT>
__device__ __forceinline__ bool tile_first_hit(const PT &P, uint32_t n, uint32_t i, uint32_t tb, uint32_t jmin,
                                               float ax, float ay, float bx, float by, float sqab,
                                               uint32_t *keyslot, int lane)
{
    NoCounts tc;
    const uint64_t m = tile_improving_mask<PRUNE>(P, n, tb, jmin, ax, ay, bx, by, sqab, lane, tc);
    if (m == 0) return false;
    if (lane == 0) atomicMin(keyslot, (i << 16) | (tb + (uint32_t)(__builtin_ffsll((long long)m) - 1)));
    return true;
}

// Dense mode: the improving moves of the reference's scan from this tile on, chained without leaving the wave.  After a hit
// at lane l the row's b becomes the old P[j] (two_opt.rs:50 reverses p[i+1..=j], so p[i+1] := p[j]) and positions > j
// are untouched, hence the lanes > l are simply decided again against the new b.  When the tile is exhausted and the wave
// owns the row's first hit, it walks on alone through the next tiles for as long as hits keep coming (at most
// kChainTiles tiles without one): early in a descent moves are a few candidates apart, and a further tile costs this wave
// ~500 cycles where handing back costs the workgroup a whole step (two barriers and the step's accounting, ~3.6k).
// Measured on the 256-restart batch (n = 10^4): 0 / 1 / 2 / 4 / 8 / 16 tiles -> 43.5k / 37.2k / 35.4k / 33.5k / 31.6k /
// 29.6k steps per descent and 128.1 / 123.2 / 123.2 / 127.3 / 133.2 / 143.6 ms (the chance of a hit in the next tile falls
// from ~18 % over the first two to ~5 % and ~3 % further out); choosing the count from the recent gap between moves, or
// requesting the next tile's coordinates ahead, measured no better.
// The chain is recorded in `hl` (hl[0] = count, hl[1] = column at which the scan resumes, hl[2], hl[3] = the row's b after the
// last hit) and in *hits_out (lane h = h-th hit column); the first hit is posted to the key slot.  Only the list of the wave that
// owns the globally first hit is used afterwards.  Returns the number of hits.
#ifndef TL_CHAIN_TILES
#define TL_CHAIN_TILES 1
#endif
static constexpr uint32_t kChainTiles = TL_CHAIN_TILES;
// The key posted for the first hit is (column << 16) | wtag: the row is implied, and the tag (the posting wave) finds the owner's
// hit list.  STOPCHK (the waves of the second round): the key slot is read beside the tile's points — one LDS round trip per
// tile instead of two; a wave that finds an earlier column already posted returns kTileStopped.
constexpr uint32_t kTileStopped = 0xFFFFFFFFu;
template <bool PRUNE, bool STOPCHK, typename TC, typename PT>
__device__ __forceinline__ uint32_t dense_tile(const PT &P, uint32_t n, uint32_t wtag, uint32_t tb, uint32_t jmin,
                                               float ax, float ay, float bx, float by, float sqab,
                                               uint32_t *hl, uint32_t *keyslot, int lane, TC &tc, uint32_t *hits_out)
{
    uint32_t j = tb + (uint32_t)lane;
    float2 c = pt_get(P, j), e = pt_get(P, j + 1u);
    uint32_t kbv = 0;
    if (STOPCHK) kbv = *keyslot;
    if (STOPCHK) {  // an earlier column already improves (kNoKey reads as column 65535)
        if (((uint32_t)__builtin_amdgcn_readfirstlane((int)kbv) >> 16) < tb) return kTileStopped;
    }
    float s1 = sqdist(make_float2(ax, ay), c), sqce = sqdist(c, e);  // the tile's terms without b: kept for the chain's re-decisions
    uint64_t m = tile_mask_terms<PRUNE>(s1, sqce, e, j, n, jmin, bx, by, sqab, tc);
    if (m == 0) return 0;  // the common case: no chain state was ever set up
    uint32_t from = jmin, nh = 0, hitv = 0, mykey = 0;  // lane h of hitv holds the h-th hit column
    bool capped = false;
    float r1 = 0.0f, rce = 0.0f;  // the chain's roots of s1 and sqce (PRUNE: L2 works on approximate roots)
    if (PRUNE) {
        r1 = __builtin_amdgcn_sqrtf(s1);
        rce = __builtin_amdgcn_sqrtf(sqce);
    }
    for (;;) {
        const int l = __builtin_ffsll((long long)m) - 1;
        const uint32_t jh = tb + (uint32_t)l;
        hitv = ((uint32_t)lane == nh) ? jh : hitv;
        if (nh == 0) {
            mykey = (jh << 16) | wtag;
            if (lane == 0) lds_min_u32(keyslot, mykey);  // post at once: it stops the other waves' scans
        }
        ++nh;
        from = jh + 1u;
        bx = readlane_f(c.x, l);  // new p[i+1] = old p[j]
        by = readlane_f(c.y, l);
        if (nh >= kMaxChainHits) {
            capped = true;
            break;
        }
        if (from > n - 2u) break;
        sqab = readlane_f(s1, l);  // the new sq(a, b) is sq(a, c) of the hit lane: the same four roundings
        const float rab = PRUNE ? readlane_f(r1, l) : 0.0f;
        m = (l == 63) ? 0ull : tile_mask_terms<PRUNE, PRUNE>(s1, sqce, e, j, n, from, bx, by, sqab, tc, r1, rce, rab);
        uint32_t idle = 0;
        bool stop = false;
        while (m == 0) {  // tile exhausted: the next one, alone
            // (a wave that does not own the first hit any more stops: its list will not be read)
            if (idle >= kChainTiles || tb + 64u > n - 2u || (uint32_t)__builtin_amdgcn_readfirstlane((int)*keyslot) != mykey) {
                stop = true;
                break;
            }
            ++idle;
            tb += 64u;
            j += 64u;
            c = pt_get(P, j);
            e = pt_get(P, j + 1u);
            s1 = sqdist(make_float2(ax, ay), c);
            sqce = sqdist(c, e);
            if (PRUNE) {
                r1 = __builtin_amdgcn_sqrtf(s1);
                rce = __builtin_amdgcn_sqrtf(sqce);
            }
            m = tile_mask_terms<PRUNE, PRUNE>(s1, sqce, e, j, n, from, bx, by, sqab, tc, r1, rce, rab);
        }
        if (stop) break;
    }
    // the list carries the row's b after these hits (the other waves go on from it); the hit columns stay in this wave's
    // register (lane h = h-th hit) — the owner files them itself after the barrier
    *hits_out = hitv;
    if (lane == 0)
        *reinterpret_cast<uint4 *>(hl) = make_uint4(nh, capped ? from : (tb + 64u), __builtin_bit_cast(uint32_t, bx), __builtin_bit_cast(uint32_t, by));
    return nh;
}

}  // namespace tl
