// dm_build.hip — all-pairs distance matrix (reference: DistanceMatrix::build,
// src/tsp/distance_matrix.rs:122-153) and the ordered tour-cost sum (:235-245).
//
// The build is a pure streaming-store kernel: 4 B written per distance, 8n B of coordinates that
// stay L2-resident (80 KB at n = 10^4).  Roofline: HBM write bandwidth — 200 MB at n = 10^4 for the
// reference's packed strict lower triangle (idx(i>j) = i(i-1)/2 + j), 400 MB for the full n x n
// layout.  Lanes run along the flat output index so every wave-instruction stores 256 contiguous
// bytes; a thread converts its first flat index to (i,j) once (f64 sqrt + integer fix-up) and then
// walks rows incrementally.
#include "tl_kernels.h"

#pragma clang fp contract(off)

namespace tl {

namespace {

constexpr int kDmThreads = 256;
constexpr int kDmPerThread = 16;  // 4096 distances (16 KB) per workgroup; 512x8, 256x8, 256x32, 1024x4 all measured slower

// Packed strict lower triangle (distance_matrix.rs:122-153): workgroup (i, s) writes columns [4096 s, 4096 s + 4096) of
// row i, 256 contiguous bytes per wave store, coordinates L2-resident.  No index arithmetic per element (a flat-index
// form with aligned stores was 25 % slower: it is the VALU work per element that counts here); rows shorter than a slab
// leave lanes idle (0.06 % of the elements at n = 10^4).
template <bool GEO>
__global__ __launch_bounds__(kDmThreads) void k_dm_build_packed_rows(const float2 *__restrict__ xy, uint32_t n, float *__restrict__ out)
{
    const uint32_t i = blockIdx.x + 1u;  // row 0 of the strict lower triangle is empty
    uint32_t j = blockIdx.y * (kDmThreads * kDmPerThread) + threadIdx.x;
    if (j >= i) return;
    const float2 a = xy[i];
    float *__restrict__ row = out + (size_t)i * (i - 1u) / 2u;
    for (int k = 0; k < kDmPerThread; ++k, j += kDmThreads) {
        if (j >= i) return;
        const float2 c = xy[j];
        row[j] = GEO ? geo_dist(a, c) : dist(a, c);  // cities[i].distance(cities[j]), j < i (a nontemporal store: 1.5x slower)
    }
}

// EUC_2D form of the same build: a workgroup takes kDmRows consecutive rows x (kDmThreads * kDmCols) columns.  The column
// coordinates are loaded ONCE into registers and reused for every row of the block (the row's own point is a wave-uniform
// scalar load), so the store loop holds no vector load at all.  tests/probes/dm_store_probe.hip takes the row kernel
// apart at n = 10^4: the bare store pattern runs at the fill ceiling (33 us, 6.0 TB/s); the per-element coordinate load
// costs +7.5 us (each iteration's load sits behind the previous iteration's store on the shared vmcnt counter, and 400 MB of
// L2 -> L1 reads ride along with the 200 MB of writes); all the arithmetic of the correctly rounded distance only +3.5 us.
// Blocking over rows removes the load term: 44.7 -> 34.6-37.5 us in the probe (4 rows x 1024 columns; 2, 3, 6, 8, 16 rows,
// 512 / 2048 / 4096 columns, 16-byte stores and other workgroup sizes all measured, none faster).
constexpr int kDmRows = 4, kDmCols = 4;
__global__ __launch_bounds__(kDmThreads) void k_dm_build_packed_blocked(const float2 *__restrict__ xy, uint32_t n, float *__restrict__ out)
{
    const uint32_t i0 = blockIdx.x * kDmRows + 1u;  // row 0 of the strict lower triangle is empty
    const uint32_t jb = blockIdx.y * (kDmThreads * kDmCols) + threadIdx.x;
    const uint32_t ilast = (i0 + kDmRows - 1u < n - 1u) ? i0 + kDmRows - 1u : n - 1u;
    if (blockIdx.y * (kDmThreads * kDmCols) >= ilast) return;  // the slab lies on or above the diagonal of every row of the block
    float2 c[kDmCols];
#pragma unroll
    for (int p = 0; p < kDmCols; ++p) {
        const uint32_t j = jb + (uint32_t)p * kDmThreads;
        c[p] = xy[j < n ? j : n - 1u];  // clamped, never stored: no divergent load
    }
#pragma unroll 1
    for (uint32_t i = i0; i <= ilast; ++i) {
        const float2 a = xy[i];
        float *__restrict__ row = out + (size_t)i * (i - 1u) / 2u;
#pragma unroll
        for (int p = 0; p < kDmCols; ++p) {
            const uint32_t j = jb + (uint32_t)p * kDmThreads;
            if (j < i) row[j] = dist(a, c[p]);  // cities[i].distance(cities[j]), j < i
        }
    }
}

// Does a caller's packed matrix hold exactly the EUC_2D distances of xy?  (The reference's DistanceMatrix does not
// remember its DistanceType, distance_matrix.rs:86-93: a drop-in caller that only has `problem.distances` learns here
// whether the on-the-fly coordinate kernels apply — bit for bit, NaN payloads aside — or the matrix kernels must run.)
__global__ __launch_bounds__(kDmThreads) void k_dm_compare_packed_rows(const float2 *__restrict__ xy, uint32_t n,
                                                                      const float *__restrict__ dm, uint32_t *__restrict__ differs)
{
    const uint32_t i = blockIdx.x + 1u;
    uint32_t j = blockIdx.y * (kDmThreads * kDmPerThread) + threadIdx.x;
    if (j >= i) return;
    const float2 a = xy[i];
    const float *__restrict__ row = dm + (size_t)i * (i - 1u) / 2u;
    bool bad = false;
    for (int k = 0; k < kDmPerThread; ++k, j += kDmThreads) {
        if (j >= i) break;
        const float want = dist(a, xy[j]), got = row[j];
        bad |= __builtin_bit_cast(uint32_t, want) != __builtin_bit_cast(uint32_t, got) && !(want != want && got != got);
    }
    if (bad) atomicOr(differs, 1u);
}

template <bool GEO>
__global__ __launch_bounds__(kDmThreads) void k_dm_build_full(const float2 *__restrict__ xy, uint32_t n,
                                                             float *__restrict__ out)
{
    // one workgroup per (row, 4096-column slab)
    const uint32_t i = blockIdx.x;  // rows on x: gridDim.y is capped at 65535
    const float2 a = xy[i];
    uint32_t j = blockIdx.y * (kDmThreads * kDmPerThread) + threadIdx.x;
    float *__restrict__ row = out + (size_t)i * n;
    for (int k = 0; k < kDmPerThread; ++k, j += kDmThreads) {
        if (j >= n) return;
        const float2 c = xy[j];
        float v;
        if (j == i) v = 0.0f;  // distance_by_pos returns 0 for equal positions (:178-180)
        else if (GEO) v = (i > j) ? geo_dist(a, c) : geo_dist(c, a);
        else v = dist(a, c);
        row[j] = v;
    }
}

// tour_length_by_pos (distance_matrix.rs:235-245): total = d(last, first); then += d(w0, w1) in order.
// Edge lengths are produced in parallel, the f32 sum strictly in the reference's order by one lane.
__global__ __launch_bounds__(1024) void k_tour_length(const float2 *__restrict__ xy,
                                                      const float *__restrict__ dm, uint32_t n,
                                                      const uint32_t *__restrict__ perm,
                                                      float *__restrict__ out_cost)
{
    __shared__ __attribute__((aligned(16))) float scratch[1024];
    const uint32_t tid = threadIdx.x;
    auto D = [&](uint32_t p, uint32_t q) -> float {
        if (p == q) return 0.0f;
        return dm ? dm_lookup(dm, p, q) : dist(xy[p], xy[q]);
    };
    float total = 0.0f;
    if (n < 2) {
        if (tid == 0) *out_cost = 0.0f;  // :236-238
        return;
    }
    if (tid == 0) total = D(perm[n - 1], perm[0]);
    for (uint32_t base = 0; base + 1 < n; base += 1024) {
        const uint32_t k = base + tid;
        scratch[tid] = (k + 1 < n) ? D(perm[k], perm[k + 1]) : 0.0f;
        TL_SYNC();
        if (tid == 0) {
            const uint32_t cnt = (n - 1 - base) < 1024u ? (n - 1 - base) : 1024u;
            for (uint32_t q = 0; q < cnt; ++q) total += scratch[q];
        }
        TL_SYNC();
    }
    if (tid == 0) *out_cost = total;
}

}  // namespace

hipError_t launch_dm_build(const float2 *xy, uint32_t n, int dist_kind, int layout, float *out, hipStream_t s)
{
    const bool geo = dist_kind == 1;
    if (layout == 0) {
        if (n < 2) return hipSuccess;
        const uint32_t per_row = kDmThreads * kDmPerThread;
        const dim3 grid(n - 1, (n - 1 + per_row - 1) / per_row);
        if (geo) hipLaunchKernelGGL(k_dm_build_packed_rows<true>, grid, dim3(kDmThreads), 0, s, xy, n, out);
        else {
            const uint32_t cols = kDmThreads * kDmCols;
            const dim3 gb((n - 1 + kDmRows - 1) / kDmRows, (n - 1 + cols - 1) / cols);
            hipLaunchKernelGGL(k_dm_build_packed_blocked, gb, dim3(kDmThreads), 0, s, xy, n, out);
        }
    } else {
        const uint32_t per = kDmThreads * kDmPerThread;
        dim3 grid(n, (n + per - 1) / per);
        if (geo) hipLaunchKernelGGL(k_dm_build_full<true>, grid, dim3(kDmThreads), 0, s, xy, n, out);
        else hipLaunchKernelGGL(k_dm_build_full<false>, grid, dim3(kDmThreads), 0, s, xy, n, out);
    }
    return hipGetLastError();
}

hipError_t launch_dm_compare(const float2 *xy, uint32_t n, const float *dm, uint32_t *differs, hipStream_t s)
{
    if (n < 2) return hipSuccess;
    const uint32_t per_row = kDmThreads * kDmPerThread;
    const dim3 grid(n - 1, (n - 1 + per_row - 1) / per_row);
    hipLaunchKernelGGL(k_dm_compare_packed_rows, grid, dim3(kDmThreads), 0, s, xy, n, dm, differs);
    return hipGetLastError();
}

hipError_t launch_tour_length(const float2 *xy, const float *dm, uint32_t n, const uint32_t *perm,
                              float *out_cost, hipStream_t s)
{
    hipLaunchKernelGGL(k_tour_length, dim3(1), dim3(1024), 0, s, xy, dm, n, perm, out_cost);
    return hipGetLastError();
}

}  // namespace tl

// ---- self-test of the numerics contract: sqrt_rn (fast path) vs the compiler's correctly rounded expansion ----
namespace tl {
__global__ __launch_bounds__(256) void k_selftest_sqrt(uint32_t first_bits, uint64_t count, unsigned long long *mismatches, uint32_t *first_bad)
{
    const uint64_t stride = (uint64_t)gridDim.x * 256u;
    unsigned long long bad = 0;
    for (uint64_t k = (uint64_t)blockIdx.x * 256u + threadIdx.x; k < count; k += stride) {
        const uint32_t bits = first_bits + (uint32_t)k;
        const float x = __builtin_bit_cast(float, bits);
        const float a = sqrt_rn(x), b = sqrt_rn_ref(x);
        const uint32_t ab = __builtin_bit_cast(uint32_t, a), bb = __builtin_bit_cast(uint32_t, b);
        if (ab != bb && !(a != a && b != b)) {  // NaN payloads aside
            ++bad;
            atomicMin(first_bad, bits);
        }
    }
    if (bad) atomicAdd(mismatches, bad);
}

hipError_t launch_selftest_sqrt(uint32_t first_bits, uint64_t count, unsigned long long *mismatches, uint32_t *first_bad, hipStream_t s)
{
    hipLaunchKernelGGL(k_selftest_sqrt, dim3(4096), dim3(256), 0, s, first_bits, count, mismatches, first_bad);
    return hipGetLastError();
}
}  // namespace tl
