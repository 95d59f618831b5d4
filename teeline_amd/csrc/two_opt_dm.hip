// two_opt_dm.hip — REF_ORDER 2-opt (src/tsp/two_opt.rs:26-61) with distances gathered from the
// reference's packed lower-triangle matrix (distance_matrix.rs:177-191) resident in HBM/L2.
//
// Used when the caller supplies `dm_packed` (EXPLICIT / GEO problems, and north-star config 2:
// "fp32 distance matrix in HBM").  One persistent workgroup per descent; the tour (u32 positions)
// lives in LDS.  A step speculatively decides 16 rows (one per wave, lanes along j) under "no move yet",
// reduces the lexicographically first improving (i,j) (ballot/ffs per wave, ds_min_u32 on i<<16|j), applies
// the reversal cooperatively and resumes at (i, j+1) — exactly the reference's loop order.
// Algorithmic bytes per candidate: perm[j+1] 4 B + D[a][c] 4 B + D[b][e] 4 B + D[c][e] 4 B = 16 B
// (SURVEY.md §8(d)); the row terms a, b, D[a][b] are amortised over the row.
#include "tl_kernels.h"

#pragma clang fp contract(off)

namespace tl {

namespace {
constexpr uint32_t kNoKey = 0xFFFFFFFFu;
constexpr int kDmNT = 1024;
}

__global__ __launch_bounds__(kDmNT) void k_two_opt_ref_dm(TwoOptBatchArgs A)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t n = A.n;
    uint32_t *perm = reinterpret_cast<uint32_t *>(smem);            // n + 1 (pad)
    uint32_t *keys = perm + ((n + 1u + 3u) & ~3u);                  // 4 slots
    float *scratch = reinterpret_cast<float *>(keys + 4);           // kDmNT floats
    const float *__restrict__ dm = A.dm;
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
    __syncthreads();

    const uint32_t nrows = n - 3;
    uint32_t i0 = 0, j0 = 2, step = 0, sweeps = 1, status = 0;
    bool improved = false;
    uint64_t moves = 0, reversed = 0;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    constexpr uint32_t NWv = kDmNT / 64;
    uint32_t since_rows = 0;  // rows scanned since the last move

    while (n >= 4) {
        const uint32_t slot = step % 3u;
        if (tid == 0) keys[(step + 1u) % 3u] = kNoKey;
        ++step;
        // Two block shapes, chosen from the observed gap between moves (like the coordinate kernel):
        //  wide  — moves are rare: 16 rows per step, one per wave, lanes along j;
        //  dense — moves every few rows: one row per step, its columns dealt to the 16 waves.
        // Either way the lexicographically first improving (i, j) wins (ds_min_u32 on i << 16 | j) and the scan
        // resumes at (i, j+1) like the reference.
        const bool wide = since_rows >= 4u;
        const uint32_t R = wide ? NWv : 1u;
        if (wide) {
            const uint32_t i = i0 + wave;
            if (i < nrows) {
                const uint32_t a = perm[i], b = perm[i + 1u];
                const float dab = dm_lookup(dm, a, b);
                const uint32_t jmin = wave == 0u ? j0 : i + 2u;
                for (uint32_t jb = jmin - (jmin & 63u); jb <= n - 2u; jb += 64u) {
                    const uint32_t kb = (uint32_t)__builtin_amdgcn_readfirstlane((int)keys[slot]);
                    if (kb != kNoKey && ((kb >> 16) < i || ((kb >> 16) == i && (kb & 0xFFFFu) < jb))) break;  // an earlier hit exists
                    const uint32_t j = jb + lane;
                    bool imp = false;
                    if (j >= jmin && j <= n - 2u) {
                        const uint32_t c = perm[j], e = perm[j + 1u];
                        const float cur = dab + dm_lookup(dm, c, e);                       // two_opt.rs:35-40
                        const float neu = dm_lookup(dm, a, c) + dm_lookup(dm, b, e);       // two_opt.rs:42-47
                        imp = neu < cur;                                                   // :49
                    }
                    const uint64_t m = __builtin_amdgcn_ballot_w64(imp);
                    if (m) {
                        if (lane == 0) atomicMin(&keys[slot], (i << 16) | (jb + (uint32_t)(__builtin_ffsll((long long)m) - 1)));
                        break;
                    }
                }
            }
        } else {
            const uint32_t i = i0;
            const uint32_t a = perm[i], b = perm[i + 1u];
            const float dab = dm_lookup(dm, a, b);
            for (uint32_t jb = j0 - (j0 & 63u) + (wave << 6); jb <= n - 2u; jb += kDmNT) {
                const uint32_t kb = (uint32_t)__builtin_amdgcn_readfirstlane((int)keys[slot]);
                if (kb != kNoKey && (kb & 0xFFFFu) < jb) break;
                const uint32_t j = jb + lane;
                bool imp = false;
                if (j >= j0 && j <= n - 2u) {
                    const uint32_t c = perm[j], e = perm[j + 1u];
                    const float cur = dab + dm_lookup(dm, c, e);
                    const float neu = dm_lookup(dm, a, c) + dm_lookup(dm, b, e);
                    imp = neu < cur;
                }
                const uint64_t m = __builtin_amdgcn_ballot_w64(imp);
                if (m) {
                    if (lane == 0) atomicMin(&keys[slot], (i << 16) | (jb + (uint32_t)(__builtin_ffsll((long long)m) - 1)));
                    break;
                }
            }
        }
        __syncthreads();
        const uint32_t key = (uint32_t)__builtin_amdgcn_readfirstlane((int)keys[slot]);
        if (key == kNoKey) {
            i0 += R;
            j0 = i0 + 2u;
            since_rows += R;
        } else {
            since_rows = 0;
            const uint32_t is = key >> 16, js = key & 0xFFFFu;
            const uint32_t lo = is + 1u, hi = js;  // swap_2opt(path, i+1, j), two_opt.rs:69-79
            const uint32_t half = (hi - lo + 1u) >> 1;
            for (uint32_t t = tid; t < half; t += kDmNT) {
                const uint32_t u = perm[lo + t], v = perm[hi - t];
                perm[lo + t] = v;
                perm[hi - t] = u;
            }
            __syncthreads();
            improved = true;
            ++moves;
            reversed += (uint64_t)(js - is);
            i0 = is;
            j0 = js + 1u;
            if (j0 > n - 2u) {
                ++i0;
                j0 = i0 + 2u;
            }
        }
        if (i0 >= nrows) {
            if (!improved) break;
            if (sweeps >= A.max_sweeps) {
                status = 1;
                break;
            }
            improved = false;
            ++sweeps;
            i0 = 0;
            j0 = 2;
        }
    }

    uint32_t *__restrict__ out = A.out_pos + (size_t)d * n;
    for (uint32_t k = tid; k < n; k += kDmNT) out[k] = perm[k];

    // tour_length_by_pos (distance_matrix.rs:235-245), sequential f32 sum in tour order
    float total = 0.0f;
    if (n >= 2 && tid == 0) total = dm_lookup(dm, perm[n - 1], perm[0]);
    for (uint32_t base = 0; base + 1 < n; base += kDmNT) {
        const uint32_t k = base + tid;
        scratch[tid] = (k + 1 < n) ? dm_lookup(dm, perm[k], perm[k + 1]) : 0.0f;
        __syncthreads();
        if (tid == 0) {
            const uint32_t cnt = (n - 1 - base) < (uint32_t)kDmNT ? (n - 1 - base) : (uint32_t)kDmNT;
            for (uint32_t q = 0; q < cnt; ++q) total += scratch[q];
        }
        __syncthreads();
    }
    if (tid == 0) {
        A.out_cost[d] = total;
        uint64_t *st = A.out_stats + (size_t)d * TL_STATS_STRIDE;
        st[0] = sweeps;
        st[1] = moves;
        st[2] = reversed;
        st[3] = status;
        st[4] = step;
    }
}

size_t two_opt_ref_dm_lds_bytes(uint32_t n)
{
    return (size_t)((n + 1u + 3u) & ~3u) * 4 + 16 + (size_t)kDmNT * 4;
}

hipError_t launch_two_opt_ref_dm(const TwoOptBatchArgs &A, uint32_t count, hipStream_t s)
{
    const size_t lds = two_opt_ref_dm_lds_bytes(A.n);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_two_opt_ref_dm),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_two_opt_ref_dm, dim3(count), dim3(kDmNT), lds, s, A);
    return hipGetLastError();
}

}  // namespace tl
