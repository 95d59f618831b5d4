// three_opt.hip — best-improvement 3-opt (reference: src/tsp/three_opt.rs).
//
//   find_best_move (three_opt.rs:58-131): scan every i<j<k (i in [0,n-2), j in [i+1,n-1), k in [j+1,n),
//   skip i==0 && k==n-1, F = path[(k+1)%n] wraps), 7 reconnection costs (:170-180), per-triple first minimum
//   among the costs < orig (:113-117), global strict-max of savings = orig - cost in loop order (:119-125).
//   apply_3opt (:186-218), solve loop (:16-51): one move per pass.
//
// This one IS best-improvement in the reference, so the whole chip works on one pass:
//   k_three_opt_prepare   Pt[k] = xy[perm[k]] (tour order, Pt[n] = Pt[0]) and E[k] = d(path[k], path[(k+1)%n])
//   k_three_opt_build_dt  the n x n matrix of distances between tour POSITIONS (coordinates: one correctly rounded
//                         distance each; explicit matrices: one gather each), rebuilt every pass — O(n^2) next to
//                         the pass's O(n^3).
//   k_three_opt_scan      one workgroup per (i, chunk of JC consecutive j).  Distance ROWS are staged in LDS:
//                         Da[k] = d(a, path[k]), Db[k] (per i), and a rolling pair Dc / Dn with Dn(j) = Dc(j+1)
//                         because D = path[j+1] is the next j's C — so each (i,j) costs ONE new row of n-j
//                         coalesced loads from Dt.  Lanes run along k: 7 stride-1 LDS reads + ~25 VALU per triple;
//                         every f32 sum is associated exactly as the reference writes it, (x + y) + z.
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
    if (A.run && A.run->done) return;  // a later pass of a batch whose descent is over
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
__global__ __launch_bounds__(256) void k_three_opt_build_dt(ThreeOptArgs A)
{
    if (A.run && A.run->done) return;
    const uint32_t n = A.n, p = blockIdx.x, q = blockIdx.y * 256u + threadIdx.x;
    if (q > n) return;
    A.Dt[(size_t)p * (n + 1u) + q] = Dpos<DM>(A.Pt, A.dm, A.perm, p, q, n);
}

__global__ __launch_bounds__(kT3) void k_three_opt_scan(ThreeOptArgs A)
{
    if (A.run && A.run->done) return;
    const uint32_t n = A.n;
    __shared__ uint32_t s_i;
    __shared__ float r_s[kT3 / 64];
    __shared__ uint32_t r_ij[kT3 / 64], r_kc[kT3 / 64];
    const uint32_t tid = threadIdx.x;
    const float *__restrict__ Dt = A.Dt;
    const float *__restrict__ E = A.E;
    const size_t rs = (size_t)n + 1u;  // row stride of Dt; column n == column 0 (F = path[(k+1) % n], :85)

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
    TL_SYNC();
    const uint32_t i = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_i);  // provably wave-uniform: scalar loads for the (i, j) terms
    const uint32_t jlo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(i + 1u + (blockIdx.x - A.chunk_prefix[i]) * A.jc));
    uint32_t jhi = jlo + A.jc;
    if (jhi > n - 1u) jhi = n - 1u;  // j in [i+1, n-1)
    const float *__restrict__ Ra = Dt + i * rs, *__restrict__ Rb = Ra + rs;  // rows of a = path[i], b = path[i+1]
    const float d_ab = E[i];

    float bs = 0.0f;  // three_opt.rs:61 best_savings = 0.0
    uint32_t bij = 0xFFFFFFFFu, bkc = 0xFFFFFFFFu;
    // A lane owns a column k and walks the chunk's j in registers: everything that depends on (i, k) only is loaded
    // once, the row of C = path[j] rolls into the row of the next j (D = path[j+1] is the next j's C), so a triple
    // costs two coalesced loads and the seven sums.  No LDS, no barrier.
    for (uint32_t k = jlo + 1u + tid; k < n; k += kT3) {
        if (i == 0u && k == n - 1u) continue;  // :81-83
        const float d_ef = E[k], d_ae = Ra[k], d_be = Rb[k], d_bf = Rb[k + 1u];
        const float *__restrict__ Rc = Dt + jlo * rs;
        float d_ce = Rc[k], d_cf = Rc[k + 1u];
        // software pipeline: the row of D = path[j+1] and the j terms of the NEXT iteration are in flight during this one
        float n_de = Rc[rs + k], n_dtf = Rc[rs + k + 1u];
        float n_c_dt = E[jlo], n_ac = Ra[jlo], n_a_dt = Ra[jlo + 1u], n_b_dt = Rb[jlo + 1u];
        for (uint32_t j = jlo; j < jhi; ++j) {  // wave-uniform trip count (scalar loads for the j terms); j >= k is masked below
            const float d_de = n_de, d_dt_f = n_dtf;
            const float d_c_dt = n_c_dt, d_ac = n_ac, d_a_dt = n_a_dt, d_b_dt = n_b_dt;
            {   // rows up to jhi+1 <= n exist (row n-1 is the last; jhi <= n-1, so j+2 <= n needs a clamp at the very end)
                const uint32_t jn = j + 1u < jhi ? j + 1u : j;  // last iteration: reload the same (unused) values
                const float *__restrict__ Rn = Dt + (jn + 1u) * rs;
                n_de = Rn[k];
                n_dtf = Rn[k + 1u];
                n_c_dt = E[jn];
                n_ac = Ra[jn];
                n_a_dt = Ra[jn + 1u];
                n_b_dt = Rb[jn + 1u];
            }
            const float orig = (d_ab + d_c_dt) + d_ef;
            const float c0 = (d_ac + d_b_dt) + d_ef;   // case 1
            const float c1 = (d_ab + d_ce) + d_dt_f;   // case 2
            const float c2 = (d_ac + d_be) + d_dt_f;   // case 3
            const float c3 = (d_a_dt + d_be) + d_cf;   // case 4
            const float c4 = (d_a_dt + d_ce) + d_bf;   // case 5
            const float c5 = (d_ae + d_b_dt) + d_cf;   // case 6
            const float c6 = (d_ae + d_c_dt) + d_bf;   // case 7
            // :113-117 leaves cmin = min(orig, c0..c6) (NaN costs never pass `c < cmin`; fminf drops them the same way),
            // and a triple matters only if it beats this thread's best so far — rare, so the case index is worked out
            // under a wave-uniform branch
            const float cm = fminf(fminf(fminf(orig, c0), fminf(c1, c2)), fminf(fminf(c3, c4), fminf(c5, c6)));
            const float sav = orig - cm;  // :120
            if (__builtin_amdgcn_ballot_w64((sav >= bs) & (sav > 0.0f) & (j < k))) {
                float cmin = orig;
                int ci = -1;
                if (c0 < cmin) { cmin = c0; ci = 0; }
                if (c1 < cmin) { cmin = c1; ci = 1; }
                if (c2 < cmin) { cmin = c2; ci = 2; }
                if (c3 < cmin) { cmin = c3; ci = 3; }
                if (c4 < cmin) { cmin = c4; ci = 4; }
                if (c5 < cmin) { cmin = c5; ci = 5; }
                if (c6 < cmin) { cmin = c6; ci = 6; }
                // :119-125 strict `>` in (i, j, k) loop order; this thread meets its triples k-major, so order by key
                if (ci >= 0 && j < k && better(orig - cmin, (i << 16) | j, (k << 3) | (uint32_t)(ci + 1), bs, bij, bkc)) {
                    bs = orig - cmin;
                    bij = (i << 16) | j;
                    bkc = (k << 3) | (uint32_t)(ci + 1);
                }
            }
            d_ce = d_de;
            d_cf = d_dt_f;
        }
    }

    // workgroup reduction: wave (shuffles) then LDS
    for (int off = 32; off > 0; off >>= 1) {
        const float os = __shfl_down(bs, off);
        const uint32_t oij = __shfl_down(bij, off), okc = __shfl_down(bkc, off);
        if (better(os, oij, okc, bs, bij, bkc)) { bs = os; bij = oij; bkc = okc; }
    }
    if ((tid & 63u) == 0u) { r_s[tid >> 6] = bs; r_ij[tid >> 6] = bij; r_kc[tid >> 6] = bkc; }
    TL_SYNC();
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
__global__ __launch_bounds__(1024) void k_three_opt_pick(ThreeOptArgs A, uint32_t nblocks, int apply, int stage_lds)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // the two segments of the move: in LDS where n entries fit, else in the workspace (one workgroup: its own barrier orders the copy)
    uint32_t *tmp = stage_lds ? reinterpret_cast<uint32_t *>(smem) : A.scratch;
    __shared__ float r_s[16];
    __shared__ uint32_t r_ij[16], r_kc[16];
    const uint32_t tid = threadIdx.x;
    if (A.run && A.run->done) return;
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
    TL_SYNC();
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
        if (A.run) {  // three_opt.rs:36-45: count the pass, file the move or end the descent
            A.run->passes += 1u;
            if (!found) {
                A.run->done = 1u;
            } else {
                const uint32_t m = A.run->moves;
                if (m < A.run->log_cap) {
                    A.log[4u * m + 0u] = bij >> 16;
                    A.log[4u * m + 1u] = bij & 0xFFFFu;
                    A.log[4u * m + 2u] = bkc >> 3;
                    A.log[4u * m + 3u] = bkc & 7u;
                }
                A.run->moves = m + 1u;
            }
        }
    }
    if (!found || !apply) return;
    const uint32_t i = bij >> 16, j = bij & 0xFFFFu, k = bkc >> 3, kase = bkc & 7u;
    uint32_t *path = A.perm;
    const uint32_t l1 = j - i, l2 = k - j, L = l1 + l2;  // seg1 = path[i+1..=j], seg2 = path[j+1..=k]
    for (uint32_t t = tid; t < L; t += 1024u) tmp[t] = path[i + 1u + t];
    TL_SYNC();
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

size_t three_opt_scan_lds_bytes(uint32_t) { return 0; }  // the scan keeps its rows in registers

hipError_t launch_three_opt_pass(const ThreeOptArgs &A, uint32_t nblocks, bool dm, int apply, hipStream_t s, int lds_budget)
{
    const uint32_t pg = (A.n + 256u) / 256u;
    const size_t lds = three_opt_scan_lds_bytes(A.n);
    hipError_t e;
    const dim3 dg(A.n, (A.n + 1u + 255u) / 256u);
    if (dm) {
        hipLaunchKernelGGL(k_three_opt_prepare<true>, dim3(pg), dim3(256), 0, s, A);
        hipLaunchKernelGGL(k_three_opt_build_dt<true>, dg, dim3(256), 0, s, A);
    } else {
        hipLaunchKernelGGL(k_three_opt_prepare<false>, dim3(pg), dim3(256), 0, s, A);
        hipLaunchKernelGGL(k_three_opt_build_dt<false>, dg, dim3(256), 0, s, A);
    }
    e = allow_max_lds(reinterpret_cast<const void *>(k_three_opt_scan));
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_three_opt_scan, dim3(nblocks), dim3(kT3), lds, s, A);
    e = allow_max_lds(reinterpret_cast<const void *>(k_three_opt_pick));
    if (e != hipSuccess) return e;
    // LDS up to 256 cities, the workspace beyond: both forms run in the parity tests at sizes the oracle affords (the copy is one
    // workgroup's and L2-resident either way; the scan is what a pass costs)
    const bool stage_lds = A.n <= 256u && (size_t)A.n * 4 + 2048 <= (size_t)lds_budget;
    hipLaunchKernelGGL(k_three_opt_pick, dim3(1), dim3(1024), stage_lds ? (size_t)A.n * 4 : 0, s, A, nblocks, apply, stage_lds ? 1 : 0);
    return hipGetLastError();
}

}  // namespace tl
