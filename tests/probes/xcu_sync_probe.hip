// xcu_sync_probe — what does ONE cross-CU exchange step cost on MI355X?
//
// VERDICT r01 item 3 proposed running one REF_ORDER 2-opt descent on G CUs: tour replicated in each CU's LDS, disjoint row
// blocks per CU, and per step one global atomicMin of the first-hit key + a spin until all G CUs have posted, after which
// every CU applies the same reversal.  A step of the single-CU kernel costs 2.0 us (dense) / 4.0 us (pruned, with a hit);
// the cooperative form can only win if the exchange is much cheaper than that.  This probe measures exactly the exchange:
// G workgroups (one per CU: each asks for 150 KB of LDS), per round  atomicMin(slot) ; atomicAdd(arrived) ; spin on
// `arrived` with agent-scope loads ; read slot  — K rounds, time / K.  Every spin is bounded (the probe cannot hang).
//
// build:  hipcc --offload-arch=gfx950 -O3 -o xcu_sync_probe tests/probes/xcu_sync_probe.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                         \
    do {                                                                                 \
        hipError_t e_ = (x);                                                             \
        if (e_ != hipSuccess) {                                                          \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                 \
            std::exit(1);                                                                \
        }                                                                                \
    } while (0)

struct Shared {
    unsigned int slot[4];
    unsigned int arrived;
    unsigned int failed;
};

__global__ __launch_bounds__(1024) void k_probe(Shared *sh, unsigned int G, unsigned int rounds, unsigned long long *cycles, unsigned int *xcc)
{
    extern __shared__ unsigned char lds[];  // only there to force one workgroup per CU
    const unsigned int tid = threadIdx.x;
    if (tid == 0) lds[0] = 0;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned int acc = 0;
    for (unsigned int r = 0; r < rounds; ++r) {
        if (tid == 0) {
            const unsigned int key = (r << 8) | blockIdx.x;
            atomicMin(&sh->slot[r & 3u], key);
            if (blockIdx.x == 0) __hip_atomic_store(&sh->slot[(r + 2u) & 3u], 0xFFFFFFFFu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(&sh->arrived, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned int want = G * (r + 1u);
            unsigned int spins = 0;
            while (__hip_atomic_load(&sh->arrived, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) {
                if (++spins > (1u << 22)) {  // ~seconds: give up, report
                    atomicAdd(&sh->failed, 1u);
                    break;
                }
            }
            acc += __hip_atomic_load(&sh->slot[r & 3u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            lds[0] = (unsigned char)acc;
        }
        __syncthreads();  // the other 15 waves wait for the exchange, as they would in the descent kernel
        if (lds[0] == 255 && tid == 1023) acc += 1;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (tid == 0) {
        cycles[blockIdx.x] = t1 - t0 + (acc == 0xFFFFFFFFu ? 1 : 0);
        unsigned int id = 0;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        xcc[blockIdx.x] = id & 0xF;
    }
}

int main()
{
    Shared *sh;
    unsigned long long *cyc;
    unsigned int *xcc;
    CHECK(hipMalloc(&sh, sizeof(Shared)));
    CHECK(hipMalloc(&cyc, 256 * sizeof(unsigned long long)));
    CHECK(hipMalloc(&xcc, 256 * sizeof(unsigned int)));
    const size_t lds = 150 * 1024;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_probe), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const unsigned int rounds = 20000;
    std::printf("{\"probe\": \"xcu_sync\", \"rounds\": %u, \"results\": [", rounds);
    bool first = true;
    for (unsigned int G : {1u, 2u, 4u, 8u, 16u, 32u}) {
        for (int rep = 0; rep < 2; ++rep) {
            Shared h{};
            for (auto &s : h.slot) s = 0xFFFFFFFFu;
            CHECK(hipMemcpy(sh, &h, sizeof(h), hipMemcpyHostToDevice));
            hipLaunchKernelGGL(k_probe, dim3(G), dim3(1024), lds, 0, sh, G, rounds, cyc, xcc);
            CHECK(hipGetLastError());
            CHECK(hipDeviceSynchronize());
            if (rep == 0) continue;  // warm-up
            std::vector<unsigned long long> c(G);
            std::vector<unsigned int> x(G);
            CHECK(hipMemcpy(c.data(), cyc, G * sizeof(unsigned long long), hipMemcpyDeviceToHost));
            CHECK(hipMemcpy(x.data(), xcc, G * sizeof(unsigned int), hipMemcpyDeviceToHost));
            CHECK(hipMemcpy(&h, sh, sizeof(h), hipMemcpyDeviceToHost));
            unsigned long long mx = 0;
            unsigned int xmask = 0;
            for (unsigned int g = 0; g < G; ++g) {
                mx = c[g] > mx ? c[g] : mx;
                xmask |= 1u << x[g];
            }
            std::printf("%s{\"workgroups\": %u, \"us_per_round\": %.3f, \"xcc_mask\": %u, \"spin_timeouts\": %u}", first ? "" : ", ", G,
                        (double)mx / 100.0 / rounds, xmask, h.failed);
            first = false;
        }
    }
    std::printf("]}\n");
    return 0;
}
