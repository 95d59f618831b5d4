// three_opt.hip — best-improvement 3-opt (reference: src/tsp/three_opt.rs).
//
//   find_best_move (three_opt.rs:58-131): scan every i<j<k (i in [0,n-2), j in [i+1,n-1), k in [j+1,n),
//   skip i==0 && k==n-1, F = path[(k+1)%n] wraps), 7 reconnection costs (:170-180), per-triple first minimum
//   among the costs < orig (:113-117), global strict-max of savings = orig - cost in loop order (:119-125).
//   apply_3opt (:186-218), solve loop (:16-51): one move per pass.
//
// This one IS best-improvement in the reference, so the whole chip works on one pass:
//   k_three_opt_prepare   Pt[k] = xy[perm[k]] (tour order, Pt[n] = Pt[0]) and E[k] = d(path[k], path[(k+1)%n])
//   k_three_opt_scan      one workgroup per (i, chunk of JC consecutive j).  Distance ROWS are staged in LDS:
//                         Da[k] = d(a, path[k]), Db[k] (per i), and a rolling pair Dc / Dn with Dn(j) = Dc(j+1)
//                         because D = path[j+1] is the next j's C — so each (i,j) costs ONE new row of n-j
//                         correctly rounded distances, i.e. one sqrt per triple instead of seven.  Lanes run
//                         along k: 7 stride-1 LDS reads + ~35 VALU per triple; every f32 sum is associated
//                         exactly as the reference writes it, (x + y) + z.
//   k_three_opt_pick      reduces the per-workgroup bests (max savings, lowest (i,j,k) on ties == the reference's
//                         strict `>` in loop order) and applies the move (apply_3opt) in place.
// Roofline: VALU (one correctly rounded sqrt + 7 three-term sums per triple); algorithmic bytes in matrix form
// would be 48 B/triple — served here from LDS rows.
#include "tl_kernels.h"

#pragma clang fp contract(off)

namespace tl {

namespace {

constexpr int kT3 = 256;  // threads per scan workgroup

__device__ __forceinline__ bool better(float sa, uint32_t ija, uint32_t kca, float sb, uint32_t ijb, uint32_t kcb)
{
    return sa > sb || (sa == sb && (ija < ijb || (ija == ijb && kca < kcb)));
}

template <bool DM>
__device__ __forceinline__ float Dpos(const float2 *__restrict__ Pt, const float *__restrict__ dm,
                                      const uint32_t *__restrict__ perm, uint32_t kp, uint32_t kq, uint32_t n)
{
    // distance between tour positions kp and kq (kq may be n == wrap to 0)
    const uint32_t q = kq == n ? 0u : kq;
    if (DM) return dm_lookup(dm, perm[kp], perm[q]);
    return dist(Pt[kp], Pt[q]);
}

}  // namespace

template <bool DM>
__global__ __launch_bounds__(256) void k_three_opt_prepare(ThreeOptArgs A)
{
    const uint32_t n = A.n;
    for (uint32_t k = blockIdx.x * 256u + threadIdx.x; k <= n; k += gridDim.x * 256u) {
        const uint32_t kk = k == n ? 0u : k;
        if (!DM) A.Pt[k] = A.xy[A.perm[kk]];
    }
    for (uint32_t k = blockIdx.x * 256u + threadIdx.x; k < n; k += gridDim.x * 256u) {
        const uint32_t p = A.perm[k], q = A.perm[(k + 1u) == n ? 0u : (k + 1u)];
        A.E[k] = DM ? dm_lookup(A.dm, p, q) : dist(A.xy[p], A.xy[q]);
    }
}

template <bool DM>
__global__ __launch_bounds__(kT3) void k_three_opt_scan(ThreeOptArgs A)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t n = A.n, stride = n + 2u;
    float *Da = reinterpret_cast<float *>(smem);
    float *Db = Da + stride;
    float *Dr0 = Db + stride;
    float *Dr1 = Dr0 + stride;
    float *El = Dr1 + stride;
    __shared__ uint32_t s_i;
    __shared__ float r_s[kT3 / 64];
    __shared__ uint32_t r_ij[kT3 / 64], r_kc[kT3 / 64];
    const uint32_t tid = threadIdx.x;
    const float2 *__restrict__ Pt = A.Pt;
    const uint32_t *__restrict__ perm = A.perm;

    // block -> (i, chunk): prefix[i] = number of chunks of rows < i
    if (tid == 0) {
        uint32_t lo = 0, hi = n - 2u;  // i in [0, n-2)
        const uint32_t b = blockIdx.x;
        while (hi - lo > 1u) {
            const uint32_t mid = (lo + hi) >> 1;
            if (A.chunk_prefix[mid] <= b) lo = mid;
            else hi = mid;
        }
        s_i = lo;
    }
    __syncthreads();
    const uint32_t i = s_i;
    const uint32_t jlo = i + 1u + (blockIdx.x - A.chunk_prefix[i]) * A.jc;
    uint32_t jhi = jlo + A.jc;
    if (jhi > n - 1u) jhi = n - 1u;  // j in [i+1, n-1)

    // rows for k in [jlo, n]  (index n == wrap to position 0)
    for (uint32_t k = jlo + tid; k <= n; k += kT3) {
        Da[k] = Dpos<DM>(Pt, A.dm, perm, i, k, n);
        Db[k] = Dpos<DM>(Pt, A.dm, perm, i + 1u, k, n);
        Dr0[k] = Dpos<DM>(Pt, A.dm, perm, jlo, k, n);
        if (k < n) El[k] = A.E[k];
    }
    __syncthreads();
    const float d_ab = A.E[i];

    float bs = 0.0f;  // three_opt.rs:61 best_savings = 0.0
    uint32_t bij = 0xFFFFFFFFu, bkc = 0xFFFFFFFFu;
    float *Dc = Dr0, *Dn = Dr1;
    for (uint32_t j = jlo; j < jhi; ++j) {
        // Dn = distances from D = path[j+1] (the next j's C row)
        for (uint32_t k = j + 1u + tid; k <= n; k += kT3) Dn[k] = Dpos<DM>(Pt, A.dm, perm, j + 1u, k, n);
        __syncthreads();
        const float d_c_dt = El[j], d_ac = Da[j], d_b_dt = Db[j + 1u], d_a_dt = Da[j + 1u];
        const float s_orig = d_ab + d_c_dt;   // (d_ab + d_c_dt) + d_ef
        const float s_c0 = d_ac + d_b_dt;     // (d_ac + d_b_dt) + d_ef
        for (uint32_t k = j + 1u + tid; k < n; k += kT3) {
            if (i == 0u && k == n - 1u) continue;  // :81-83
            const float d_ef = El[k], d_ae = Da[k], d_be = Db[k], d_bf = Db[k + 1u];
            const float d_ce = Dc[k], d_cf = Dc[k + 1u], d_dt_f = Dn[k + 1u];
            const float orig = s_orig + d_ef;
            float cmin = orig;
            int ci = -1;
            float c;
            c = s_c0 + d_ef;              if (c < cmin) { cmin = c; ci = 0; }   // case 1
            c = (d_ab + d_ce) + d_dt_f;   if (c < cmin) { cmin = c; ci = 1; }   // case 2
            c = (d_ac + d_be) + d_dt_f;   if (c < cmin) { cmin = c; ci = 2; }   // case 3
            c = (d_a_dt + d_be) + d_cf;   if (c < cmin) { cmin = c; ci = 3; }   // case 4
            c = (d_a_dt + d_ce) + d_bf;   if (c < cmin) { cmin = c; ci = 4; }   // case 5
            c = (d_ae + d_b_dt) + d_cf;   if (c < cmin) { cmin = c; ci = 5; }   // case 6
            c = (d_ae + d_c_dt) + d_bf;   if (c < cmin) { cmin = c; ci = 6; }   // case 7
            if (ci >= 0) {
                const float sav = orig - cmin;  // :120
                if (sav > bs) {                  // :121 strict; a thread visits (j,k) in ascending order
                    bs = sav;
                    bij = (i << 16) | j;
                    bkc = (k << 3) | (uint32_t)(ci + 1);
                }
            }
        }
        __syncthreads();
        float *t = Dc;
        Dc = Dn;
        Dn = t;
    }

    // workgroup reduction: wave (shuffles) then LDS
    for (int off = 32; off > 0; off >>= 1) {
        const float os = __shfl_down(bs, off);
        const uint32_t oij = __shfl_down(bij, off), okc = __shfl_down(bkc, off);
        if (better(os, oij, okc, bs, bij, bkc)) { bs = os; bij = oij; bkc = okc; }
    }
    if ((tid & 63u) == 0u) { r_s[tid >> 6] = bs; r_ij[tid >> 6] = bij; r_kc[tid >> 6] = bkc; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < kT3 / 64; ++w)
            if (better(r_s[w], r_ij[w], r_kc[w], bs, bij, bkc)) { bs = r_s[w]; bij = r_ij[w]; bkc = r_kc[w]; }
        ThreeOptBest *o = A.partials + blockIdx.x;
        o->sav = bs;
        o->ij = bij;
        o->kc = bkc;
    }
}

// one workgroup: reduce the partials, publish the move, optionally apply it (apply_3opt, three_opt.rs:186-218)
__global__ __launch_bounds__(1024) void k_three_opt_pick(ThreeOptArgs A, uint32_t nblocks, int apply)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *tmp = reinterpret_cast<uint32_t *>(smem);  // n entries
    __shared__ float r_s[16];
    __shared__ uint32_t r_ij[16], r_kc[16];
    const uint32_t tid = threadIdx.x;
    float bs = 0.0f;
    uint32_t bij = 0xFFFFFFFFu, bkc = 0xFFFFFFFFu;
    for (uint32_t b = tid; b < nblocks; b += 1024u) {
        const ThreeOptBest p = A.partials[b];
        if (better(p.sav, p.ij, p.kc, bs, bij, bkc)) { bs = p.sav; bij = p.ij; bkc = p.kc; }
    }
    for (int off = 32; off > 0; off >>= 1) {
        const float os = __shfl_down(bs, off);
        const uint32_t oij = __shfl_down(bij, off), okc = __shfl_down(bkc, off);
        if (better(os, oij, okc, bs, bij, bkc)) { bs = os; bij = oij; bkc = okc; }
    }
    if ((tid & 63u) == 0u) { r_s[tid >> 6] = bs; r_ij[tid >> 6] = bij; r_kc[tid >> 6] = bkc; }
    __syncthreads();
    bs = r_s[0]; bij = r_ij[0]; bkc = r_kc[0];
    for (int w = 1; w < 16; ++w)
        if (better(r_s[w], r_ij[w], r_kc[w], bs, bij, bkc)) { bs = r_s[w]; bij = r_ij[w]; bkc = r_kc[w]; }
    const bool found = bkc != 0xFFFFFFFFu;  // savings > 0 was required to record a move
    if (tid == 0) {
        A.best->sav = bs;
        A.best->ij = bij;
        A.best->kc = bkc;
        A.best->found = found ? 1u : 0u;
        if (apply) {
            A.counters[0] += 1;                  // passes
            if (found) A.counters[1] += 1;       // moves
        }
    }
    if (!found || !apply) return;
    const uint32_t i = bij >> 16, j = bij & 0xFFFFu, k = bkc >> 3, kase = bkc & 7u;
    uint32_t *path = A.perm;
    const uint32_t l1 = j - i, l2 = k - j, L = l1 + l2;  // seg1 = path[i+1..=j], seg2 = path[j+1..=k]
    for (uint32_t t = tid; t < L; t += 1024u) tmp[t] = path[i + 1u + t];
    __syncthreads();
    for (uint32_t t = tid; t < L; t += 1024u) {
        uint32_t src;  // index into tmp (0..l1-1 = seg1, l1.. = seg2)
        switch (kase) {
        case 1: src = t < l1 ? (l1 - 1u - t) : t; break;                                  // rev(s1) + s2
        case 2: src = t < l1 ? t : (l1 + (L - 1u - t)); break;                            // s1 + rev(s2)
        case 3: src = t < l1 ? (l1 - 1u - t) : (l1 + (L - 1u - t)); break;                // rev(s1) + rev(s2)
        case 4: src = t < l2 ? (l1 + t) : (t - l2); break;                                // s2 + s1
        case 5: src = t < l2 ? (l1 + t) : (l1 - 1u - (t - l2)); break;                    // s2 + rev(s1)
        case 6: src = t < l2 ? (l1 + (l2 - 1u - t)) : (t - l2); break;                    // rev(s2) + s1
        default: src = t < l2 ? (l1 + (l2 - 1u - t)) : (l1 - 1u - (t - l2)); break;       // 7: rev(s2) + rev(s1)
        }
        path[i + 1u + t] = tmp[src];
    }
}

size_t three_opt_scan_lds_bytes(uint32_t n) { return (size_t)5 * (n + 2u) * 4; }

hipError_t launch_three_opt_pass(const ThreeOptArgs &A, uint32_t nblocks, bool dm, int apply, hipStream_t s)
{
    const uint32_t pg = (A.n + 256u) / 256u;
    const size_t lds = three_opt_scan_lds_bytes(A.n);
    hipError_t e;
    if (dm) {
        hipLaunchKernelGGL(k_three_opt_prepare<true>, dim3(pg), dim3(256), 0, s, A);
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_three_opt_scan<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_three_opt_scan<true>, dim3(nblocks), dim3(kT3), lds, s, A);
    } else {
        hipLaunchKernelGGL(k_three_opt_prepare<false>, dim3(pg), dim3(256), 0, s, A);
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_three_opt_scan<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_three_opt_scan<false>, dim3(nblocks), dim3(kT3), lds, s, A);
    }
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_three_opt_pick), hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)A.n * 4));
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_three_opt_pick, dim3(1), dim3(1024), (size_t)A.n * 4, s, A, nblocks, apply);
    return hipGetLastError();
}

}  // namespace tl
