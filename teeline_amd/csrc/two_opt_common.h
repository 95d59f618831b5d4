// two_opt_common.h — device helpers shared by the 2-opt kernels: wave min/max reductions (DPP), the per-tile
// bounding-box metadata of the exact L0 bound and the f32 lower bound itself (DESIGN.md "Exact decision cascade").
#pragma once
#include "tl_device.h"

#pragma clang fp contract(off)

namespace tl {

template <int CTRL>
__device__ __forceinline__ float dpp_shr(float v)
{
    // row_shr within rows of 16 lanes; lanes without a source keep their own value (old = v)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_min(float v)
{
    v = fminf(v, dpp_shr<0x111>(v));
    v = fminf(v, dpp_shr<0x112>(v));
    v = fminf(v, dpp_shr<0x114>(v));
    v = fminf(v, dpp_shr<0x118>(v));  // lane 15 of each row: min of the row (min is idempotent)
    return fminf(fminf(readlane_f(v, 15), readlane_f(v, 31)), fminf(readlane_f(v, 47), readlane_f(v, 63)));
}
__device__ __forceinline__ float wave_max(float v)
{
    v = fmaxf(v, dpp_shr<0x111>(v));
    v = fmaxf(v, dpp_shr<0x112>(v));
    v = fmaxf(v, dpp_shr<0x114>(v));
    v = fmaxf(v, dpp_shr<0x118>(v));
    return fmaxf(fmaxf(readlane_f(v, 15), readlane_f(v, 31)), fmaxf(readlane_f(v, 47), readlane_f(v, 63)));
}

// L0 metadata of tile t: bounding box of P[64t .. 64t+64] over the positions that take part in a
// candidate (j <= n-2 as c, j+1 as e) and the largest squared tour-edge sq(P[j],P[j+1]) in it.
__device__ __forceinline__ void build_tile_meta(const float2 *P, uint32_t n, uint32_t t, int lane, float4 *tbox, float *tmsq)
{
    const uint32_t j = (t << 6) + (uint32_t)lane;
    const bool valid = j + 2u <= n;  // j <= n-2
    const float2 c = P[j], e = P[j + 1u];
    const float inf = __builtin_inff();
    const float mnx = wave_min(valid ? fminf(c.x, e.x) : inf), mny = wave_min(valid ? fminf(c.y, e.y) : inf);
    const float mxx = wave_max(valid ? fmaxf(c.x, e.x) : -inf), mxy = wave_max(valid ? fmaxf(c.y, e.y) : -inf);
    const float msq = wave_max(valid ? sqdist(c, e) : -1.0f);
    if (lane == 0) {
        tbox[t] = make_float4(mnx, mny, mxx, mxy);
        tmsq[t] = msq;
    }
}

// f32 lower bound of sqdist(p, q) over every q inside the box (monotone ops only -> valid in f32).
// d = max(lo - p, p - hi, 0) is taken on the bit patterns as signed ints: negative floats are negative
// ints and non-negative floats order like ints, so one v_max3_i32 does it without NaN canonicalisation.
__device__ __forceinline__ float box_lb(float px, float py, float4 box)
{
    const int ux = __builtin_bit_cast(int, box.x - px), vx = __builtin_bit_cast(int, px - box.z);
    const int uy = __builtin_bit_cast(int, box.y - py), vy = __builtin_bit_cast(int, py - box.w);
    const int mx = ux > vx ? ux : vx, my = uy > vy ? uy : vy;
    const float dx = __builtin_bit_cast(float, mx > 0 ? mx : 0);
    const float dy = __builtin_bit_cast(float, my > 0 ? my : 0);
    return dx * dx + dy * dy;
}

}  // namespace tl
