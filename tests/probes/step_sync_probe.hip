// step_sync_probe — what do the pieces of ONE step of the LDS 2-opt descent kernel cost on a CU of MI355X?
//
// The descent (two_opt_ref.hip) is a chain of ~35 k steps; a step is "every wave learns the step, scans, the workgroup agrees on
// the first hit".  This probe times the synchronisation skeletons such a step can be built from, on ONE workgroup per CU
// (NW = 16 / 8 / 4 waves), in shader cycles per iteration (s_memtime on wave 0, 4096 iterations):
//   bar        s_barrier only
//   bar_lds    s_barrier ; ds_read (uniform) ; v_readfirstlane ; dependent use                (what a worker does per step)
//   sym K      every wave runs K dependent SALU instructions, then s_barrier                   (classic form: state machine in all waves)
//   ctl K      wave 0: ds_read ; K dependent SALU ; ds_write ; B0 ; B2     workers: B0 ; ds_read ; readfirstlane ; B2   (role split)
//   poll K     wave 0: K SALU ; ds_write seq ; then spins on an arrival counter      workers: spin on seq (s_sleep 1) ; ds_add arrival
//              (no s_barrier at all)
// build:  hipcc --offload-arch=gfx950 -O3 -o step_sync_probe tests/probes/step_sync_probe.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                         \
    do {                                                                 \
        hipError_t e_ = (x);                                             \
        if (e_ != hipSuccess) {                                          \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); \
            std::exit(1);                                                \
        }                                                                \
    } while (0)

template <int K>
__device__ __forceinline__ unsigned salu_chain(unsigned x)
{
#pragma unroll
    for (int k = 0; k < K; ++k) asm volatile("s_add_u32 %0, %0, 1" : "+s"(x) : : "scc");
    return x;
}

constexpr int ITERS = 4096;

template <int MODE, int K>
__global__ __launch_bounds__(1024) void k_probe(unsigned long long *out, unsigned *sink)
{
    __shared__ unsigned sh[64];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nw = blockDim.x >> 6;
    if (tid < 64) sh[tid] = 0;
    __syncthreads();
    unsigned acc = (unsigned)__builtin_amdgcn_readfirstlane((int)sh[1]);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (MODE == 0) {
        for (int it = 0; it < ITERS; ++it) __builtin_amdgcn_s_barrier();
    } else if (MODE == 1) {
        for (int it = 0; it < ITERS; ++it) {
            __builtin_amdgcn_s_barrier();
            const unsigned v = *(volatile unsigned *)&sh[acc & 3u];
            acc += (unsigned)__builtin_amdgcn_readfirstlane((int)v) + 1u;
        }
    } else if (MODE == 2) {
        for (int it = 0; it < ITERS; ++it) {
            acc = salu_chain<K>(acc);
            __builtin_amdgcn_s_barrier();
        }
    } else if (MODE == 3) {
        if (wave == 0) {
            for (int it = 0; it < ITERS; ++it) {
                const unsigned v = *(volatile unsigned *)&sh[8 + (acc & 3u)];
                acc += (unsigned)__builtin_amdgcn_readfirstlane((int)v);
                acc = salu_chain<K>(acc);
                if ((tid & 63) == 0) *(volatile unsigned *)&sh[0] = acc;
                __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
                __builtin_amdgcn_s_barrier();  // B0
                __builtin_amdgcn_s_barrier();  // B2
            }
        } else {
            for (int it = 0; it < ITERS; ++it) {
                __builtin_amdgcn_s_barrier();  // B0
                const unsigned v = *(volatile unsigned *)&sh[0];
                acc += (unsigned)__builtin_amdgcn_readfirstlane((int)v);
                __builtin_amdgcn_s_barrier();  // B2
            }
        }
    } else if (MODE == 4) {
        // sequence word sh[0], arrival counter sh[16]; every spin is bounded
        if (wave == 0) {
            for (int it = 1; it <= ITERS; ++it) {
                acc = salu_chain<K>(acc);
                if ((tid & 63) == 0) *(volatile unsigned *)&sh[0] = (unsigned)it;
                const unsigned want = (unsigned)it * (unsigned)(nw - 1);
                unsigned spins = 0;
                while ((unsigned)__builtin_amdgcn_readfirstlane((int)*(volatile unsigned *)&sh[16]) < want && ++spins < (1u << 14)) __builtin_amdgcn_s_sleep(1);
                if (spins >= (1u << 14)) break;  // give up (reported as an absurd cycle count)
            }
            if ((tid & 63) == 0) *(volatile unsigned *)&sh[0] = 0x7FFFFFFFu;  // release every worker
        } else {
            for (int it = 1; it <= ITERS; ++it) {
                unsigned spins = 0;
                while ((unsigned)__builtin_amdgcn_readfirstlane((int)*(volatile unsigned *)&sh[0]) < (unsigned)it && ++spins < (1u << 14)) __builtin_amdgcn_s_sleep(1);
                if (spins >= (1u << 14)) break;
                if ((tid & 63) == 0) atomicAdd(&sh[16], 1u);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) out[blockIdx.x] = t1 - t0;
    if (acc == 0xdeadbeefu) sink[0] = acc;
}

template <int MODE, int K>
static void run(const char *name, int nw, unsigned long long *d_out, unsigned *d_sink, int blocks)
{
    std::vector<unsigned long long> h((size_t)blocks);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((k_probe<MODE, K>), dim3(blocks), dim3(nw * 64), 100 * 1024, 0, d_out, d_sink);
        CHECK(hipGetLastError());
        CHECK(hipDeviceSynchronize());
    }
    CHECK(hipMemcpy(h.data(), d_out, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost));
    double s = 0;
    for (auto v : h) s += (double)v;
    std::printf("{\"pattern\": \"%s\", \"K\": %d, \"waves\": %d, \"blocks\": %d, \"cycles_per_iter\": %.1f}\n", name, K, nw, blocks, s / blocks / ITERS);
    std::fflush(stdout);
}

int main()
{
    unsigned long long *d_out;
    unsigned *d_sink;
    const int blocks = 256;
    CHECK(hipMalloc(&d_out, sizeof(unsigned long long) * blocks));
    CHECK(hipMalloc(&d_sink, 64));
    for (int nw : {16, 8, 4}) {
        run<0, 0>("bar", nw, d_out, d_sink, blocks);
        run<1, 0>("bar_lds", nw, d_out, d_sink, blocks);
        run<2, 0>("sym", nw, d_out, d_sink, blocks);
        run<2, 50>("sym", nw, d_out, d_sink, blocks);
        run<2, 150>("sym", nw, d_out, d_sink, blocks);
        run<3, 0>("ctl", nw, d_out, d_sink, blocks);
        run<3, 50>("ctl", nw, d_out, d_sink, blocks);
        run<3, 150>("ctl", nw, d_out, d_sink, blocks);
        run<4, 0>("poll", nw, d_out, d_sink, blocks);
        run<4, 150>("poll", nw, d_out, d_sink, blocks);
    }
    return 0;
}
