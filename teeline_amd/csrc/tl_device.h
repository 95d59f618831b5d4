// tl_device.h — device-side helpers shared by every kernel of libteeline_gpu (gfx950 only).
//
// Numerics contract (reference: src/tsp/kdtree.rs:291-295 `KDPoint::distance`):
//   d = sqrt(dx*dx + dy*dy) in f32, separate roundings for both products and the sum (Rust never
//   contracts to FMA), correctly rounded sqrt.  This file is compiled with -ffp-contract=off and
//   also pins the pragma below so a stray build flag cannot change results.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#pragma clang fp contract(off)

namespace tl {

constexpr int kWave = 64;  // CDNA wavefront width

typedef float v2f __attribute__((ext_vector_type(2)));

// fl(fl(dx*dx) + fl(dy*dy)): written on 2-vectors so that it becomes v_pk_add_f32, v_pk_mul_f32, v_add_f32 — the same
// three roundings per component as the scalar form, no operand shuffling.
// The final add is a plain v_add_f32 written as inline asm: left to itself the SLP vectoriser packs the adds of two neighbouring
// sqdist calls into one v_pk_add_f32 and pays three v_mov to line the operands up (seen in every tile pass of the 2-opt kernels).
__device__ __forceinline__ float add_f32_nopack(float x, float y)
{
    float r;
    asm("v_add_f32_e32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
    return r;
}
__device__ __forceinline__ float sqdist(float2 p, float2 q)
{
    const v2f a = {p.x, p.y}, b = {q.x, q.y};
    v2f d = a - b;
    d = d * d;
    return add_f32_nopack(d.x, d.y);
}

// Correctly rounded f32 sqrt, the compiler's way: hipcc (-fhip-fp32-correctly-rounded-divide-sqrt, default on)
// expands this to a 2^32 pre-scale for tiny inputs, v_sqrt_f32, a +-1 ulp FMA fix-up, the un-scale and a zero/inf
// select (~17 VALU).
__device__ __forceinline__ float sqrt_rn_ref(float x) { return __builtin_sqrtf(x); }

// The same algorithm without the scaling: v_sqrt_f32 (<= 1 ulp), then pick among y-1ulp, y, y+1ulp with two exact FMA
// residuals (r = x - y'*y).  Valid for x = 0, +inf and every x >= 2^-96 (where v_sqrt_f32 needs no denormal help);
// smaller non-zero inputs take the compiler's full expansion under a wave-uniform branch.  ~9 VALU.
// tl_selftest_sqrt compares it with sqrt_rn_ref over whole bit-pattern ranges on the device (tests/test_gpu_numerics.py
// sweeps all 2^31 non-negative floats).
__device__ __forceinline__ float sqrt_rn_fast(float x)  // the fix-up alone: x = 0, +inf or x >= 2^-96 (callers check)
{
    const float y = __builtin_amdgcn_sqrtf(x);
    const int yi = __builtin_bit_cast(int, y);
    const float yd = __builtin_bit_cast(float, yi - 1), yu = __builtin_bit_cast(float, yi + 1);
    const float rd = __builtin_fmaf(-yd, y, x), ru = __builtin_fmaf(-yu, y, x);
    float r = (rd <= 0.0f) ? yd : y;
    r = (ru > 0.0f) ? yu : r;
    return r;
}

__device__ __forceinline__ float sqrt_rn(float x)
{
    if (__builtin_amdgcn_ballot_w64((x < 1.2621774e-29f) & (x != 0.0f))) return sqrt_rn_ref(x);  // 2^-96
    return sqrt_rn_fast(x);
}

__device__ __forceinline__ float dist(float2 p, float2 q) { return sqrt_rn(sqdist(p, q)); }

// distance_matrix.rs:59-75 geo_distance (f64 trig -> floor -> f32)
__device__ __forceinline__ double geo_to_rad(float x)
{
    const double PI = 3.14159265358979323846264338327950288;
    double deg = (double)truncf(x);
    double min = (double)(x - truncf(x));
    return PI * (deg + 5.0 * min / 3.0) / 180.0;
}

__device__ __forceinline__ float geo_dist(float2 p, float2 q)
{
    double lat1 = geo_to_rad(p.x), lon1 = geo_to_rad(p.y);
    double lat2 = geo_to_rad(q.x), lon2 = geo_to_rad(q.y);
    double q1 = cos(lon1 - lon2);
    double q2 = cos(lat1 - lat2);
    double q3 = cos(lat1 + lat2);
    const double RRR = 6378.388;
    return (float)floor(RRR * acos(0.5 * ((1.0 + q1) * q2 - (1.0 - q1) * q3)) + 1.0);
}

// distance_matrix.rs:177-191 distance_by_pos on the packed strict lower triangle
__device__ __forceinline__ float dm_lookup(const float *__restrict__ packed, uint32_t p, uint32_t q)
{
    if (p == q) return 0.0f;
    uint64_t from = p > q ? p : q;
    uint64_t to = p > q ? q : p;
    return packed[from * (from - 1) / 2 + to];
}

__device__ __forceinline__ uint64_t splitmix64_at(uint64_t seed, uint64_t k)
{
    // k-th output (k = 0,1,...) of the splitmix64 stream started at `seed`: the state is a counter,
    // so draws are independent of each other and can be computed in parallel.
    uint64_t z = seed + (k + 1) * 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

// Workgroup barrier.  The race-stress build (-DTL_JITTER, libteeline_gpu_jitter.so, tests/test_gpu_race_stress.py) parks
// pseudo-randomly chosen waves for 3-30 us right after every barrier, so that waves leave it far apart: an exchange through
// LDS that is only safe because the waves happen to run in step shows up as a result that differs from the oracle's.
// (Found this way: the LDS 2-opt kernel's "some key is posted -> skip round 2" test, two_opt_ref.hip.)
#ifdef TL_JITTER
__device__ __forceinline__ void jitter(uint32_t salt)
{
    uint32_t h = (uint32_t)__builtin_amdgcn_s_memtime() ^ (salt * 0x9E3779B1u) ^ ((threadIdx.x >> 6) * 0x85EBCA6Bu) ^
                 (blockIdx.x * 0xC2B2AE35u);
    h ^= h >> 15;
    h *= 0x2C1B3C6Du;
    h ^= h >> 12;
    h = (uint32_t)__builtin_amdgcn_readfirstlane((int)h);
    if ((h & 7u) == 0u) {
        const uint32_t reps = (h >> 3) & 7u;
        for (uint32_t k = 0; k <= reps; ++k) __builtin_amdgcn_s_sleep(127);  // 127 x 64 clocks
    }
}
#define TL_SYNC()           \
    do {                    \
        __syncthreads();    \
        tl::jitter(__LINE__); \
    } while (0)
#else
#define TL_SYNC() __syncthreads()
#endif

// One lane's ds_min_u32 exactly as written.  atomicMin on a wave-uniform LDS address goes through the compiler's atomic optimiser,
// which wraps it in a scan over the active lanes (~20 scalar instructions and a v_mbcnt pair) even where the caller has already
// narrowed exec to one lane — on the path every wave of a descent waits for.
// An LDS operation issued through inline asm is NOT counted by the compiler's s_waitcnt insertion: a barrier that follows it
// gets its `s_waitcnt lgkmcnt(0)` only if some tracked LDS operation of the wave sits in between (LDS operations of a wave
// complete in order, so waiting for a later one covers this one).  Callers therefore either follow the post with a tracked LDS
// write before the step's barrier (the hit list of a dense step, two_opt_common.h / two_opt_dm.hip) or use lds_min_u32_fenced.
// teeline_amd/build.py disassembles the linked library and refuses it unless every path from a ds_min_u32 to the next
// s_barrier passes an `s_waitcnt lgkmcnt(0)` (verify_ds_min_waits; tests/test_build_asm.py).
__device__ __forceinline__ void lds_min_u32(uint32_t *p, uint32_t v)
{
    const uint32_t a = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)p;
    asm volatile("ds_min_u32 %0, %1" : : "v"(a), "v"(v) : "memory");
}
// ... with its own wait: for a post that is followed by a barrier with no tracked LDS operation of this wave in between (the
// pruned step's hit, two_opt_ref.hip — rare there, so the ~100 cycles of the wait are not on a hot path).
__device__ __forceinline__ void lds_min_u32_fenced(uint32_t *p, uint32_t v)
{
    const uint32_t a = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)p;
    asm volatile("ds_min_u32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : : "v"(a), "v"(v) : "memory");
}

__device__ __forceinline__ float readlane_f(float v, int l)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}

}  // namespace tl
