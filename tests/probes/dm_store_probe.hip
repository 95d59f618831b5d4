// dm_store_probe — where do the 47 us of k_dm_build_packed_rows (n = 10^4, 200 MB) go?  Same grid and store pattern, work
// removed piece by piece:  0 = store only (value from registers), 1 = + coordinate load, 2 = + squared distance,
// 3 = + v_sqrt_f32 (no fix-up), 4 = the full correctly rounded distance;  5 = a plain float4 fill of the same bytes.
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I teeline_amd/csrc -o dm_store_probe tests/probes/dm_store_probe.hip
#include "tl_device.h"

#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace tl;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

constexpr int kT = 256, kP = 16;

template <typename F>
static float time_us_fn(F launch)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 6; ++rep) {
        CHECK(hipEventRecord(e0, 0));
        launch();
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best) best = ms;
    }
    return best * 1e3f;
}

template <int MODE, int PER, int THREADS>
__global__ __launch_bounds__(THREADS) void k_rows(const float2 *__restrict__ xy, uint32_t n, float *__restrict__ out)
{
    const uint32_t i = blockIdx.x + 1u;
    uint32_t j = blockIdx.y * (THREADS * PER) + threadIdx.x;
    if (j >= i) return;
    const float2 a = xy[i];
    float *__restrict__ row = out + (size_t)i * (i - 1u) / 2u;
    for (int k = 0; k < PER; ++k, j += THREADS) {
        if (j >= i) return;
        float v = a.x + (float)j;
        if (MODE >= 1) {
            const float2 c = xy[j];
            if (MODE == 1) v = c.x + c.y;
            if (MODE == 2) v = sqdist(a, c);
            if (MODE == 3) v = __builtin_amdgcn_sqrtf(sqdist(a, c));
            if (MODE == 4) v = dist(a, c);
        }
        row[j] = v;
    }
}

// R consecutive rows x (THREADS * P) columns per workgroup: the column coordinates are loaded ONCE into registers and reused
// for R rows (the row's own point is a wave-uniform scalar load); the inner loops hold no vector load at all.
template <int R, int P, int THREADS>
__global__ __launch_bounds__(THREADS) void k_rows_blocked(const float2 *__restrict__ xy, uint32_t n, float *__restrict__ out)
{
    const uint32_t i0 = blockIdx.x * R + 1u;
    const uint32_t jb = blockIdx.y * (THREADS * P) + threadIdx.x;
    const uint32_t ilast = (i0 + R - 1u < n - 1u) ? i0 + R - 1u : n - 1u;
    if (blockIdx.y * (THREADS * P) >= ilast) return;  // slab wholly on or above the diagonal of every row of the block
    float2 c[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const uint32_t j = jb + (uint32_t)p * THREADS;
        c[p] = xy[j < n ? j : n - 1u];
    }
#pragma unroll 1
    for (uint32_t i = i0; i <= ilast; ++i) {
        const float2 a = xy[i];
        float *__restrict__ row = out + (size_t)i * (i - 1u) / 2u;
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const uint32_t j = jb + (uint32_t)p * THREADS;
            if (j < i) row[j] = dist(a, c[p]);
        }
    }
}

// the same with 4 CONSECUTIVE columns per lane: one 16-byte store per lane and row (dword-aligned only: rows start anywhere)
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
template <int R, int THREADS>
__global__ __launch_bounds__(THREADS) void k_rows_blocked_x4(const float2 *__restrict__ xy, uint32_t n, float *__restrict__ out)
{
    const uint32_t i0 = blockIdx.x * R + 1u;
    const uint32_t jb = (blockIdx.y * THREADS + threadIdx.x) * 4u;
    const uint32_t ilast = (i0 + R - 1u < n - 1u) ? i0 + R - 1u : n - 1u;
    if (blockIdx.y * (THREADS * 4) >= ilast) return;
    float2 c[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) c[p] = xy[jb + p < n ? jb + p : n - 1u];
#pragma unroll 1
    for (uint32_t i = i0; i <= ilast; ++i) {
        const float2 a = xy[i];
        float *__restrict__ row = out + (size_t)i * (i - 1u) / 2u;
        if (jb + 3u < i) {
            f4u v;
            v.x = dist(a, c[0]); v.y = dist(a, c[1]); v.z = dist(a, c[2]); v.w = dist(a, c[3]);
            *reinterpret_cast<f4u *>(row + jb) = v;
        } else {
#pragma unroll
            for (int p = 0; p < 4; ++p)
                if (jb + p < i) row[jb + p] = dist(a, c[p]);
        }
    }
}
template <int R, int THREADS>
static float run_blocked_x4(const float2 *xy, uint32_t n, float *out)
{
    const dim3 grid((n - 1 + R - 1) / R, (n - 1 + THREADS * 4 - 1) / (THREADS * 4));
    return time_us_fn([&] { hipLaunchKernelGGL((k_rows_blocked_x4<R, THREADS>), grid, dim3(THREADS), 0, 0, xy, n, out); });
}

// blocked, with the column slab on blockIdx.x (consecutive workgroups write adjacent 4 KB pieces of the same rows)
template <int R, int P, int THREADS>
__global__ __launch_bounds__(THREADS) void k_rows_blocked_swapped(const float2 *__restrict__ xy, uint32_t n, float *__restrict__ out)
{
    const uint32_t i0 = blockIdx.y * R + 1u;
    const uint32_t jb = blockIdx.x * (THREADS * P) + threadIdx.x;
    const uint32_t ilast = (i0 + R - 1u < n - 1u) ? i0 + R - 1u : n - 1u;
    if (blockIdx.x * (THREADS * P) >= ilast) return;
    float2 c[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const uint32_t j = jb + (uint32_t)p * THREADS;
        c[p] = xy[j < n ? j : n - 1u];
    }
#pragma unroll 1
    for (uint32_t i = i0; i <= ilast; ++i) {
        const float2 a = xy[i];
        float *__restrict__ row = out + (size_t)i * (i - 1u) / 2u;
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const uint32_t j = jb + (uint32_t)p * THREADS;
            if (j < i) row[j] = dist(a, c[p]);
        }
    }
}
template <int R, int P, int THREADS>
static float run_blocked_swapped(const float2 *xy, uint32_t n, float *out)
{
    const dim3 grid((n - 1 + THREADS * P - 1) / (THREADS * P), (n - 1 + R - 1) / R);
    return time_us_fn([&] { hipLaunchKernelGGL((k_rows_blocked_swapped<R, P, THREADS>), grid, dim3(THREADS), 0, 0, xy, n, out); });
}

// blocked, launched over the (row block, column slab) pairs of the lower triangle only (no workgroup that exits at once):
// linear id -> slab by a short scalar walk over the slabs (<= n / (THREADS * P) steps)
template <int R, int P, int THREADS>
__global__ __launch_bounds__(THREADS) void k_rows_blocked_tri(const float2 *__restrict__ xy, uint32_t n, float *__restrict__ out, uint32_t nrb)
{
    uint32_t id = blockIdx.x, slab = 0;
    for (;;) {  // slab s holds the row blocks rb with (rb + 1) * R > s * THREADS * P, i.e. rb >= first(s)
        const uint32_t first = (slab * (THREADS * P)) / R;
        const uint32_t cnt = nrb - (first < nrb ? first : nrb);
        if (id < cnt) break;
        id -= cnt;
        ++slab;
    }
    const uint32_t rb = (slab * (THREADS * P)) / R + id;
    const uint32_t i0 = rb * R + 1u;
    const uint32_t jb = slab * (THREADS * P) + threadIdx.x;
    const uint32_t ilast = (i0 + R - 1u < n - 1u) ? i0 + R - 1u : n - 1u;
    float2 c[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const uint32_t j = jb + (uint32_t)p * THREADS;
        c[p] = xy[j < n ? j : n - 1u];
    }
#pragma unroll 1
    for (uint32_t i = i0; i <= ilast; ++i) {
        const float2 a = xy[i];
        float *__restrict__ row = out + (size_t)i * (i - 1u) / 2u;
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const uint32_t j = jb + (uint32_t)p * THREADS;
            if (j < i) row[j] = dist(a, c[p]);
        }
    }
}
template <int R, int P, int THREADS>
static float run_blocked_tri(const float2 *xy, uint32_t n, float *out)
{
    const uint32_t nrb = (n - 1 + R - 1) / R, nslab = (n - 1 + THREADS * P - 1) / (THREADS * P);
    uint32_t total = 0;
    for (uint32_t s = 0; s < nslab; ++s) {
        const uint32_t first = (s * (THREADS * P)) / R;
        total += nrb - (first < nrb ? first : nrb);
    }
    return time_us_fn([&] { hipLaunchKernelGGL((k_rows_blocked_tri<R, P, THREADS>), dim3(total), dim3(THREADS), 0, 0, xy, n, out, nrb); });
}

template <int R, int P, int THREADS>
static float run_blocked(const float2 *xy, uint32_t n, float *out)
{
    const dim3 grid((n - 1 + R - 1) / R, (n - 1 + THREADS * P - 1) / (THREADS * P));
    return time_us_fn([&] { hipLaunchKernelGGL((k_rows_blocked<R, P, THREADS>), grid, dim3(THREADS), 0, 0, xy, n, out); });
}

__global__ __launch_bounds__(256) void k_fill4(float4 *__restrict__ out, size_t n4)
{
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < n4; k += stride) out[k] = make_float4(1.f, 2.f, 3.f, 4.f);
}

template <typename F>
static float time_us(F launch)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 6; ++rep) {
        CHECK(hipEventRecord(e0, 0));
        launch();
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best) best = ms;
    }
    return best * 1e3f;
}

template <int MODE, int PER, int THREADS>
static float run_rows(const float2 *xy, uint32_t n, float *out)
{
    const uint32_t per_row = THREADS * PER;
    const dim3 grid(n - 1, (n - 1 + per_row - 1) / per_row);
    return time_us([&] { hipLaunchKernelGGL((k_rows<MODE, PER, THREADS>), grid, dim3(THREADS), 0, 0, xy, n, out); });
}

int main()
{
    const uint32_t n = 10000;
    const size_t elems = (size_t)n * (n - 1) / 2;
    std::vector<float2> h(n);
    uint64_t s = 88172645463325252ull;
    for (auto &p : h) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17; p.x = (float)(s % 1000000) / 1000.0f;
        s ^= s << 13; s ^= s >> 7; s ^= s << 17; p.y = (float)(s % 1000000) / 1000.0f;
    }
    float2 *xy;
    float *out;
    CHECK(hipMalloc(&xy, n * sizeof(float2)));
    CHECK(hipMalloc(&out, elems * 4 + 256));
    CHECK(hipMemcpy(xy, h.data(), n * sizeof(float2), hipMemcpyHostToDevice));
    const double gb = elems * 4 / 1e9;
    auto line = [&](const char *what, float us) { std::printf("  {\"variant\": \"%s\", \"us\": %.1f, \"GBps\": %.0f},\n", what, us, gb / (us * 1e-6)); };
    std::printf("{\"probe\": \"dm_store\", \"n\": %u, \"bytes\": %zu, \"results\": [\n", n, elems * 4);
    line("rows 256x16: store only", run_rows<0, kP, kT>(xy, n, out));
    line("rows 256x16: + coordinate load", run_rows<1, kP, kT>(xy, n, out));
    line("rows 256x16: + squared distance", run_rows<2, kP, kT>(xy, n, out));
    line("rows 256x16: + v_sqrt_f32", run_rows<3, kP, kT>(xy, n, out));
    line("rows 256x16: full correctly rounded distance", run_rows<4, kP, kT>(xy, n, out));
    line("rows 256x4: full", run_rows<4, 4, 256>(xy, n, out));
    line("rows 256x40: full (one workgroup per row)", run_rows<4, 40, 256>(xy, n, out));
    line("rows 1024x10: full (one workgroup per row)", run_rows<4, 10, 1024>(xy, n, out));
    line("rows 512x8: full", run_rows<4, 8, 512>(xy, n, out));
    line("tri R=4 P=4 T=256", run_blocked_tri<4, 4, 256>(xy, n, out));
    line("tri R=8 P=4 T=256", run_blocked_tri<8, 4, 256>(xy, n, out));
    line("tri R=4 P=2 T=256", run_blocked_tri<4, 2, 256>(xy, n, out));
    line("tri R=4 P=4 T=256 (again)", run_blocked_tri<4, 4, 256>(xy, n, out));
    line("blocked R=4  P=4 T=256 (ref0)", run_blocked<4, 4, 256>(xy, n, out));
    line("swapped R=4 P=4 T=256", run_blocked_swapped<4, 4, 256>(xy, n, out));
    line("swapped R=8 P=4 T=256", run_blocked_swapped<8, 4, 256>(xy, n, out));
    line("swapped R=4 P=2 T=256", run_blocked_swapped<4, 2, 256>(xy, n, out));
    line("swapped R=4 P=8 T=256", run_blocked_swapped<4, 8, 256>(xy, n, out));
    line("blocked R=4  P=4 T=256 (ref)", run_blocked<4, 4, 256>(xy, n, out));
    line("swapped R=4 P=4 T=256 (again)", run_blocked_swapped<4, 4, 256>(xy, n, out));
    line("blocked R=2  P=4 T=256", run_blocked<2, 4, 256>(xy, n, out));
    line("blocked R=2  P=8 T=256", run_blocked<2, 8, 256>(xy, n, out));
    line("blocked R=3  P=4 T=256", run_blocked<3, 4, 256>(xy, n, out));
    line("blocked R=4  P=2 T=256", run_blocked<4, 2, 256>(xy, n, out));
    line("blocked R=4  P=8 T=256", run_blocked<4, 8, 256>(xy, n, out));
    line("blocked R=4  P=16 T=256", run_blocked<4, 16, 256>(xy, n, out));
    line("blocked R=6  P=4 T=256", run_blocked<6, 4, 256>(xy, n, out));
    line("blocked R=4  P=4 T=512", run_blocked<4, 4, 512>(xy, n, out));
    line("blocked R=4  P=4 T=1024", run_blocked<4, 4, 1024>(xy, n, out));
    line("blocked x4 R=2 T=256", run_blocked_x4<2, 256>(xy, n, out));
    line("blocked x4 R=4 T=256", run_blocked_x4<4, 256>(xy, n, out));
    line("blocked x4 R=8 T=256", run_blocked_x4<8, 256>(xy, n, out));
    line("blocked x4 R=4 T=64", run_blocked_x4<4, 64>(xy, n, out));
    line("blocked R=4  P=4 T=256", run_blocked<4, 4, 256>(xy, n, out));
    line("blocked R=8  P=4 T=256", run_blocked<8, 4, 256>(xy, n, out));
    line("blocked R=16 P=4 T=256", run_blocked<16, 4, 256>(xy, n, out));
    line("blocked R=8  P=8 T=256", run_blocked<8, 8, 256>(xy, n, out));
    line("blocked R=16 P=2 T=256", run_blocked<16, 2, 256>(xy, n, out));
    line("blocked R=32 P=2 T=256", run_blocked<32, 2, 256>(xy, n, out));
    line("blocked R=16 P=1 T=256", run_blocked<16, 1, 256>(xy, n, out));
    line("blocked R=64 P=1 T=256", run_blocked<64, 1, 256>(xy, n, out));
    line("blocked R=16 P=4 T=128", run_blocked<16, 4, 128>(xy, n, out));
    line("blocked R=16 P=2 T=512", run_blocked<16, 2, 512>(xy, n, out));
    line("rows 256x16: store only (again)", run_rows<0, kP, kT>(xy, n, out));
    const size_t n4 = elems / 4;
    line("float4 fill, 4096 workgroups", time_us([&] { hipLaunchKernelGGL(k_fill4, dim3(4096), dim3(256), 0, 0, (float4 *)out, n4); }));
    line("float4 fill, 16384 workgroups", time_us([&] { hipLaunchKernelGGL(k_fill4, dim3(16384), dim3(256), 0, 0, (float4 *)out, n4); }));
    std::printf("  {}]}\n");
    return 0;
}
