// nn_knn.hip — candidate lists by brute force and the nearest-neighbour seed (split off lk.hip in round 4).
//
//   k_knn        build_candidates (lin_kernighan.rs:12-27) and DistanceMatrix::nearest (distance_matrix.rs:259-280):
//                k nearest cities per city, ascending f32 distance, stable on ties in position order — the
//                k-buffer rule of NearestResult::add (mod.rs:1839-1860).  Brute force, coordinates streamed through LDS
//                tiles; exact (correctly rounded) distances because ties are decided on the rounded values.  (The product's
//                LK lists come from the reference's kd-tree, kdtree.hip; these scans feed the NN seed and TL_FLAG_KNN_BRUTE.)
//   k_nn_seed    nearest_neighbor::solve (nearest_neighbor.rs:8-76): first unvisited among the n_nearest
//                closest, else the globally nearest unvisited (tie: lowest position).  One workgroup; lane 0
//                walks the candidate lists, the whole group does the O(n) fallback scans.
//   k_nn_seed_dm the same walk over an EXPLICIT / GEO problem's packed matrix.
#include "tl_kernels.h"
#include <type_traits>

#pragma clang fp contract(off)

namespace tl {

namespace {

constexpr int kLkNT = 1024;  // threads of the NN-seed workgroup (the name it had inside lk.hip)

// ---------------------------------------------------------------------------------------------- k-NN
template <int KMAX>
__global__ __launch_bounds__(256) void k_knn(const float2 *__restrict__ xy, uint32_t n, uint32_t k, uint32_t *__restrict__ cand)
{
    __shared__ float2 tile[256];
    const uint32_t c = blockIdx.x * 256u + threadIdx.x;
    const float2 pc = xy[c < n ? c : 0u];
    float bd[KMAX];
    uint32_t bp[KMAX];
#pragma unroll
    for (int t = 0; t < KMAX; ++t) { bd[t] = __builtin_inff(); bp[t] = 0xFFFFFFFFu; }
    float radius = __builtin_inff(), rlim = __builtin_inff();
    for (uint32_t base = 0; base < n; base += 256u) {
        TL_SYNC();
        if (base + threadIdx.x < n) tile[threadIdx.x] = xy[base + threadIdx.x];
        TL_SYNC();
        const uint32_t lim = (n - base) < 256u ? (n - base) : 256u;
        for (uint32_t t = 0; t < lim; ++t) {
            const uint32_t p = base + t;
            // sqrt is monotone: a squared distance above (radius (1 + 2^-20))^2 cannot give d < radius.  Almost every candidate
            // fails that for all 64 cities of the wave, and the correctly rounded root is only taken for the rest.
            const float sq = sqdist(tile[t], pc);
            if (!__builtin_amdgcn_ballot_w64((sq <= rlim) & (p != c))) continue;
            if (p == c) continue;                    // self excluded (mod.rs:1840-1842)
            const float d = sqrt_rn(sq);             // kdtree.rs:194 self.point.distance(target)
            // insert iff d < search_radius (INF until the buffer holds k, then the k-th kept distance), at
            // partition_point(r.distance <= d): AFTER equal distances, then truncate to k
            if (d < radius) {
                float cd = d;
                uint32_t cp = p;
                bool shifting = false;
#pragma unroll
                for (int s = 0; s < KMAX; ++s) {
                    if ((uint32_t)s < k && (shifting || cd < bd[s])) {
                        const float td = bd[s];
                        const uint32_t tp = bp[s];
                        bd[s] = cd;
                        bp[s] = cp;
                        cd = td;
                        cp = tp;
                        shifting = true;
                    }
                }
                radius = __builtin_inff();
#pragma unroll
                for (int s = 0; s < KMAX; ++s)
                    if ((uint32_t)s + 1u == k) radius = bd[s];
                rlim = radius * radius * 1.000002f;  // inf while the buffer is not full (and on overflow: no filtering)
            }
        }
    }
    if (c < n) {
#pragma unroll
        for (int t = 0; t < KMAX; ++t)
            if ((uint32_t)t < k) cand[(size_t)c * k + t] = bp[t];
    }
}

template <int CTRL>
__device__ __forceinline__ uint32_t dpp_shr_keep(uint32_t v)
{
    // row_shr within rows of 16 lanes; lanes without a source keep their own value
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xf, 0xf, false);
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v)  // the minimum over the wave, in every lane (as SGPR values)
{
    uint32_t o;
    o = dpp_shr_keep<0x111>(v); v = o < v ? o : v;
    o = dpp_shr_keep<0x112>(v); v = o < v ? o : v;
    o = dpp_shr_keep<0x114>(v); v = o < v ? o : v;
    o = dpp_shr_keep<0x118>(v); v = o < v ? o : v;  // lane 15 of each row: min of the row
    const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)v, 15), b = (uint32_t)__builtin_amdgcn_readlane((int)v, 31);
    const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)v, 47), d = (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
    const uint32_t ab = a < b ? a : b, cd = c < d ? c : d;
    return ab < cd ? ab : cd;
}

// Several lanes per city (a DPP quad or a row of 16): lane q takes the candidates p = q, q+G, ... of every tile with its own k-buffer (the
// same insertion rule; positions ascend inside a lane, so ties stay stable), then the four sorted buffers are merged by k
// rounds of "smallest head of the quad" on the packed key (d bits << 32 | position) — the k smallest in (distance,
// position) order, exactly what one lane scanning everything keeps.  Four times the lanes of k_knn: at n ~ 10^4 one lane
// per city leaves most of the chip idle.
__device__ __forceinline__ unsigned long long quad_min_u64(unsigned long long v)
{
    const uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    {
        const uint32_t olo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)lo, 0xB1 /* quad_perm:[1,0,3,2] */, 0xf, 0xf, false);
        const uint32_t ohi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hi, 0xB1, 0xf, 0xf, false);
        const unsigned long long o = ((unsigned long long)ohi << 32) | olo;
        v = o < v ? o : v;
    }
    {
        const uint32_t lo2 = (uint32_t)v, hi2 = (uint32_t)(v >> 32);
        const uint32_t olo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)lo2, 0x4E /* quad_perm:[2,3,0,1] */, 0xf, 0xf, false);
        const uint32_t ohi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hi2, 0x4E, 0xf, 0xf, false);
        const unsigned long long o = ((unsigned long long)ohi << 32) | olo;
        v = o < v ? o : v;
    }
    return v;
}

// the same over a row of 16 lanes: butterfly of row rotations, every lane ends with the minimum
template <int CTRL>
__device__ __forceinline__ unsigned long long row_min_step_u64(unsigned long long v)
{
    const uint32_t olo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)v, CTRL, 0xf, 0xf, false);
    const uint32_t ohi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(v >> 32), CTRL, 0xf, 0xf, false);
    const unsigned long long o = ((unsigned long long)ohi << 32) | olo;
    return o < v ? o : v;
}
__device__ __forceinline__ unsigned long long row16_min_u64(unsigned long long v)
{
    v = row_min_step_u64<0x128>(v);  // row_ror:8
    v = row_min_step_u64<0x124>(v);  // row_ror:4
    v = row_min_step_u64<0x122>(v);  // row_ror:2
    v = row_min_step_u64<0x121>(v);  // row_ror:1
    return v;
}

// GROUP lanes per city (4 = a DPP quad, 16 = a DPP row)
template <int KMAX, int GROUP>
__global__ __launch_bounds__(256) void k_knn_quad(const float2 *__restrict__ xy, uint32_t n, uint32_t k, uint32_t *__restrict__ cand)
{
    __shared__ float2 tile[256];
    static_assert(GROUP == 4 || GROUP == 16, "lanes per city");
    constexpr uint32_t G = GROUP, CPB = 256u / G;  // cities per block
    const uint32_t c = blockIdx.x * CPB + (threadIdx.x / G), q = threadIdx.x % G;
    const float2 pc = xy[c < n ? c : 0u];
    float bd[KMAX];
    uint32_t bp[KMAX];
#pragma unroll
    for (int t = 0; t < KMAX; ++t) { bd[t] = __builtin_inff(); bp[t] = 0xFFFFFFFFu; }
    float radius = __builtin_inff(), rlim = __builtin_inff();
    for (uint32_t base = 0; base < n; base += 256u) {
        TL_SYNC();
        if (base + threadIdx.x < n) tile[threadIdx.x] = xy[base + threadIdx.x];
        TL_SYNC();
        const uint32_t lim = (n - base) < 256u ? (n - base) : 256u;
        for (uint32_t t = q; t < ((lim + G - 1u) & ~(G - 1u)); t += G) {  // the group walks together (wave-uniform trip count)
            const uint32_t p = base + t;
            const bool in = t < lim && p != c;
            const float sq = sqdist(tile[t < lim ? t : 0u], pc);
            if (!__builtin_amdgcn_ballot_w64((sq <= rlim) & in)) continue;
            const float d = sqrt_rn(sq);
            if (in && d < radius) {
                float cd = d;
                uint32_t cp = p;
                bool shifting = false;
#pragma unroll
                for (int s = 0; s < KMAX; ++s) {
                    if ((uint32_t)s < k && (shifting || cd < bd[s])) {
                        const float td = bd[s];
                        const uint32_t tp = bp[s];
                        bd[s] = cd;
                        bp[s] = cp;
                        cd = td;
                        cp = tp;
                        shifting = true;
                    }
                }
                radius = __builtin_inff();
#pragma unroll
                for (int s = 0; s < KMAX; ++s)
                    if ((uint32_t)s + 1u == k) radius = bd[s];
                rlim = radius * radius * 1.000002f;
            }
        }
    }
    // merge the quad's four sorted buffers
    uint32_t head = 0;
    for (uint32_t r = 0; r < k; ++r) {
        float hd = __builtin_inff();
        uint32_t hp = 0xFFFFFFFFu;
#pragma unroll
        for (int s = 0; s < KMAX; ++s) {
            if ((uint32_t)s == head) { hd = bd[s]; hp = bp[s]; }
        }
        const unsigned long long key = (head < k && hp != 0xFFFFFFFFu) ? (((unsigned long long)__builtin_bit_cast(uint32_t, hd) << 32) | hp) : ~0ULL;
        const unsigned long long best = GROUP == 4 ? quad_min_u64(key) : row16_min_u64(key);
        if (key == best && best != ~0ULL) ++head;
        if (q == 0u && c < n) cand[(size_t)c * k + r] = best == ~0ULL ? 0xFFFFFFFFu : (uint32_t)best;
    }
}

// ---------------------------------------------------------------------------------------------- NN seed
// nearest_neighbor::solve (nearest_neighbor.rs:8-76) is one sequential walk: n steps, each "first unvisited among the
// n_nearest closest" (:44-49) or, failing that, the globally nearest unvisited city (:50-63; ties -> lowest position, the
// oracle's rule where the reference iterates a HashSet).  One workgroup; what the walk touches per step lives in LDS so a
// step costs LDS latencies, not HBM/L2 round trips: the visited flags (n bytes), the candidate lists as u16 positions
// (2kn bytes, when n < 65536 and they fit) and, if there is room left, the coordinates for the fallback scans
// (8n bytes).  Wave 0 walks (lane t checks the t-th nearest); the fallback scan is the whole workgroup in two u32 passes (smallest
// squared distance, then the lowest position at the rounded minimum), DPP wave reductions, one LDS atomicMin per wave.
// NQ: cities per thread whose coordinates stay in registers for the fallback scans (n <= NQ * 1024; 0 = none, loop form)

template <bool LDS_CAND, bool LDS_XY, int NQ>
__global__ __launch_bounds__(kLkNT) void k_nn_seed(const float2 *__restrict__ xy, uint32_t n, const uint32_t *__restrict__ cand,
                                                   uint32_t k, uint32_t *__restrict__ path)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char nn_smem[];
    __shared__ uint32_t s_len, s_cur, s_minsq, s_minpos;
    unsigned char *visited = nn_smem;                                                      // n bytes
    const size_t o_cand = ((size_t)n + 15u) & ~(size_t)15u;
    uint16_t *lc = reinterpret_cast<uint16_t *>(nn_smem + o_cand);                         // k*n u16 (LDS_CAND)
    const size_t o_xy = o_cand + (LDS_CAND ? ((((size_t)n * k * 2u) + 15u) & ~(size_t)15u) : 0u);
    float2 *lxy = reinterpret_cast<float2 *>(nn_smem + o_xy);                              // n float2 (LDS_XY)
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    for (uint32_t p = tid; p < n; p += kLkNT) visited[p] = 0;
    if (LDS_CAND)
        for (size_t e = tid; e < (size_t)n * k; e += kLkNT) lc[e] = (uint16_t)cand[e];    // 0xFFFFFFFF (no neighbour) -> 0xFFFF
    if (LDS_XY)
        for (uint32_t p = tid; p < n; p += kLkNT) lxy[p] = xy[p];
    constexpr int kNnRegs = NQ > 0 ? NQ : 1;
    const bool regs = NQ > 0;
    float2 rxy[kNnRegs];
#pragma unroll
    for (int m = 0; m < kNnRegs; ++m) {
        const uint32_t p = tid + (uint32_t)m * kLkNT;
        rxy[m] = (regs && p < n) ? xy[p] : make_float2(0.f, 0.f);
    }
    TL_SYNC();
    if (tid == 0) {
        path[0] = 0;  // :28 start = cities[0]
        visited[0] = 1;
        s_len = 1;
        s_cur = 0;
        s_minsq = 0xFFFFFFFFu;
        s_minpos = 0xFFFFFFFFu;
    }
    TL_SYNC();
    while (true) {
        if (tid < 64u) {
            // the walk (:44-49), wave 0: lane t looks at the t-th nearest of the current city, the first lane whose city is
            // unvisited wins — two LDS latencies per step
            uint32_t len = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_len), cur = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_cur);
            const uint32_t fb = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_minpos);
            if (fb != 0xFFFFFFFFu) {  // the city the last fallback scan found
                if (lane == 0) {
                    path[len] = fb;
                    visited[fb] = 1;
                    s_minsq = 0xFFFFFFFFu;
                    s_minpos = 0xFFFFFFFFu;
                }
                ++len;
                cur = fb;
            }
            while (len < n) {
                uint32_t q = 0xFFFFFFFFu;
                if (lane < k) {
                    if (LDS_CAND) {
                        const uint32_t v = lc[(size_t)cur * k + lane];
                        q = v == 0xFFFFu ? 0xFFFFFFFFu : v;
                    } else {
                        q = cand[(size_t)cur * k + lane];
                    }
                }
                const bool open = q != 0xFFFFFFFFu && !visited[q];
                const uint64_t m = __builtin_amdgcn_ballot_w64(open);
                if (m == 0) break;
                const uint32_t nx = (uint32_t)__builtin_amdgcn_readlane((int)q, __builtin_ffsll((long long)m) - 1);
                if (lane == 0) {
                    path[len] = nx;
                    visited[nx] = 1;
                }
                ++len;
                cur = nx;
            }
            if (lane == 0) {
                s_len = len;
                s_cur = cur;
            }
        }
        TL_SYNC();
        if (s_len >= n) break;
        // :50-63 fallback: globally nearest unvisited; ties -> lowest position (the reference iterates a HashSet).
        // sqrt is monotone, so the nearest city has the smallest squared distance: pass 1 reduces min sq (u32 bits order
        // like the floats), pass 2 takes the lowest position among the cities whose ROUNDED distance equals the rounded
        // minimum (different squares can round to the same f32 distance, and the reference compares distances).
        const uint32_t cur = s_cur;
        const float2 pc = LDS_XY ? lxy[cur] : xy[cur];
        uint32_t msq = 0xFFFFFFFFu, mpos = 0xFFFFFFFFu;
        if (regs) {
            // this thread's cities (tid + 1024 m) never change: coordinates in registers, squared distances kept for pass 2
            uint32_t sqb[kNnRegs];
            unsigned char vis[kNnRegs];
#pragma unroll
            for (int m = 0; m < kNnRegs; ++m) {  // all flag reads in flight together (slots beyond n read a valid byte, masked below)
                const uint32_t p = tid + (uint32_t)m * kLkNT;
                vis[m] = visited[p < n ? p : 0u];
            }
#pragma unroll
            for (int m = 0; m < kNnRegs; ++m) {
                const uint32_t p = tid + (uint32_t)m * kLkNT;
                const uint32_t b = __builtin_bit_cast(uint32_t, sqdist(pc, rxy[m]));
                sqb[m] = (p < n && !vis[m]) ? b : 0xFFFFFFFFu;
                msq = sqb[m] < msq ? sqb[m] : msq;
            }
            msq = wave_min_u32(msq);
            if (lane == 0 && msq != 0xFFFFFFFFu) atomicMin(&s_minsq, msq);
            TL_SYNC();
            const uint32_t gsq = s_minsq;
            const float dmin = sqrt_rn(__builtin_bit_cast(float, gsq));
            const uint32_t limb = __builtin_bit_cast(uint32_t, __builtin_bit_cast(float, gsq) * 1.000001f);
#pragma unroll
            for (int m = 0; m < kNnRegs; ++m) {
                if (sqb[m] <= limb) {  // closed cities and slots beyond n carry 0xFFFFFFFF (uniform guards here measured slower)
                    const uint32_t p = tid + (uint32_t)m * kLkNT;
                    if (sqrt_rn(__builtin_bit_cast(float, sqb[m])) == dmin) mpos = p < mpos ? p : mpos;
                }
            }
        } else {
            for (uint32_t p = tid; p < n; p += kLkNT) {
                if (visited[p]) continue;
                const uint32_t b = __builtin_bit_cast(uint32_t, sqdist(pc, LDS_XY ? lxy[p] : xy[p]));
                msq = b < msq ? b : msq;
            }
            msq = wave_min_u32(msq);
            if (lane == 0 && msq != 0xFFFFFFFFu) atomicMin(&s_minsq, msq);
            TL_SYNC();
            const uint32_t gsq = s_minsq;
            const float dmin = sqrt_rn(__builtin_bit_cast(float, gsq));
            // squares that can still round to dmin lie within a few ulps of the minimum; everything else is farther
            const float lim = __builtin_bit_cast(float, gsq) * 1.000001f;
            for (uint32_t p = tid; p < n; p += kLkNT) {
                if (visited[p]) continue;
                const float sq = sqdist(pc, LDS_XY ? lxy[p] : xy[p]);
                if (sq <= lim && sqrt_rn(sq) == dmin) mpos = p < mpos ? p : mpos;
            }
        }
        mpos = wave_min_u32(mpos);
        if (lane == 0 && mpos != 0xFFFFFFFFu) atomicMin(&s_minpos, mpos);
        TL_SYNC();
    }
}

}  // namespace

hipError_t launch_knn(const float2 *xy, uint32_t n, uint32_t k, uint32_t *cand, hipStream_t s, int form)
{
    if (form != 1) {  // several lanes per city: 16 while that still fills the chip (n <= 32 K), else 4 (form 4 forces it)
        if (n <= 32768u && form != 4) {
            const uint32_t gq = (n + 15u) / 16u;
            if (k <= 4) hipLaunchKernelGGL((k_knn_quad<4, 16>), dim3(gq), dim3(256), 0, s, xy, n, k, cand);
            else if (k <= 8) hipLaunchKernelGGL((k_knn_quad<8, 16>), dim3(gq), dim3(256), 0, s, xy, n, k, cand);
            else if (k <= 16) hipLaunchKernelGGL((k_knn_quad<16, 16>), dim3(gq), dim3(256), 0, s, xy, n, k, cand);
            else if (k <= 32) hipLaunchKernelGGL((k_knn_quad<32, 16>), dim3(gq), dim3(256), 0, s, xy, n, k, cand);
            else hipLaunchKernelGGL((k_knn_quad<64, 16>), dim3(gq), dim3(256), 0, s, xy, n, k, cand);  // (round 5: lists of up to 64, the k-buffer in registers)
            return hipGetLastError();
        }
        const uint32_t gq = (n + 63u) / 64u;
        if (k <= 4) hipLaunchKernelGGL((k_knn_quad<4, 4>), dim3(gq), dim3(256), 0, s, xy, n, k, cand);
        else if (k <= 8) hipLaunchKernelGGL((k_knn_quad<8, 4>), dim3(gq), dim3(256), 0, s, xy, n, k, cand);
        else if (k <= 16) hipLaunchKernelGGL((k_knn_quad<16, 4>), dim3(gq), dim3(256), 0, s, xy, n, k, cand);
        else if (k <= 32) hipLaunchKernelGGL((k_knn_quad<32, 4>), dim3(gq), dim3(256), 0, s, xy, n, k, cand);
        else hipLaunchKernelGGL((k_knn_quad<64, 4>), dim3(gq), dim3(256), 0, s, xy, n, k, cand);  // (round 5: lists of up to 64, the k-buffer in registers)
        return hipGetLastError();
    }
#ifdef TL_TUNE  // one lane per city: a rejected form, tuning build only
    const uint32_t grid = (n + 255u) / 256u;
    if (k <= 4) hipLaunchKernelGGL(k_knn<4>, dim3(grid), dim3(256), 0, s, xy, n, k, cand);
    else if (k <= 8) hipLaunchKernelGGL(k_knn<8>, dim3(grid), dim3(256), 0, s, xy, n, k, cand);
    else hipLaunchKernelGGL(k_knn<16>, dim3(grid), dim3(256), 0, s, xy, n, k, cand);
    return hipGetLastError();
#else
    return hipErrorInvalidValue;
#endif
}

hipError_t launch_nn_seed(const float2 *xy, uint32_t n, const uint32_t *cand, uint32_t k, uint32_t *path, int lds_bytes, hipStream_t s)
{
    // what fits into the CU's LDS next to the visited flags: the candidate lists (u16), then the coordinates
    const size_t cap = (size_t)lds_bytes > 1024 ? (size_t)lds_bytes - 1024 : 0;  // static shared + slack
    const size_t b_vis = ((size_t)n + 15u) & ~(size_t)15u;
    const size_t b_cand = (((size_t)n * k * 2u) + 15u) & ~(size_t)15u;
    const size_t b_xy = (size_t)n * 8u;
    if (b_vis > cap) return hipErrorInvalidValue;  // n beyond ~160 K cities: callers check
    const bool lds_cand = k > 0 && n < 65535u && b_vis + b_cand <= cap;
    const bool lds_xy = b_vis + (lds_cand ? b_cand : 0) + b_xy <= cap;
    const size_t lds = b_vis + (lds_cand ? b_cand : 0) + (lds_xy ? b_xy : 0);
    auto go = [&](auto kern) -> hipError_t {
        hipError_t e = allow_max_lds(reinterpret_cast<const void *>(kern));
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3(1), dim3(kLkNT), lds, s, xy, n, cand, k, path);
        return hipGetLastError();
    };
    auto pick = [&](auto cand_t, auto xy_t) -> hipError_t {
        constexpr bool LC = decltype(cand_t)::value, LX = decltype(xy_t)::value;
        const uint32_t per = (n + (uint32_t)kLkNT - 1u) / (uint32_t)kLkNT;  // cities per thread
        if (per <= 4u) return go(k_nn_seed<LC, LX, 4>);
        if (per <= 8u) return go(k_nn_seed<LC, LX, 8>);
        if (per <= 12u) return go(k_nn_seed<LC, LX, 12>);
        if (per <= 16u) return go(k_nn_seed<LC, LX, 16>);
        return go(k_nn_seed<LC, LX, 0>);
    };
    if (lds_cand && lds_xy) return pick(std::true_type{}, std::true_type{});
    if (lds_cand) return pick(std::true_type{}, std::false_type{});
    if (lds_xy) return pick(std::false_type{}, std::true_type{});
    return pick(std::false_type{}, std::false_type{});
}

// ---------------------------------------------------------------------------------------------- NN seed, matrix form
// nearest_neighbor::solve over an EXPLICIT / GEO problem (nearest_neighbor.rs:8-76 with problem.distances =
// the packed matrix, distance_matrix.rs:259-297): every step takes the first unvisited city in (distance, position)
// order — what both the k-buffer rule (mod.rs:1848-1855, scan in position order, insert after equals) and the fallback
// (:50-63; ties -> lowest position, the oracle's rule where the reference iterates a HashSet) pick.  One workgroup, the
// visited flags in LDS, one row of the matrix per step: argmin of the packed key (sortable distance bits << 32 | position).
__global__ __launch_bounds__(kLkNT) void k_nn_seed_dm(const float *__restrict__ dm, uint32_t n, uint32_t *__restrict__ path)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char nn_smem[];
    __shared__ unsigned long long s_best[2];
    unsigned char *visited = nn_smem;
    const uint32_t tid = threadIdx.x;
    for (uint32_t p = tid; p < n; p += kLkNT) visited[p] = p == 0u;  // :28-30 starts at cities[0]
    if (tid < 2) s_best[tid] = ~0ull;
    if (tid == 0) path[0] = 0u;
    TL_SYNC();
    uint32_t cur = 0;
    for (uint32_t step = 1; step < n; ++step) {
        unsigned long long best = ~0ull;
        for (uint32_t p = tid; p < n; p += kLkNT) {
            if (visited[p]) continue;
            const uint32_t b = __builtin_bit_cast(uint32_t, dm_lookup(dm, cur, p));
            const uint32_t key = (b & 0x80000000u) ? ~b : (b | 0x80000000u);  // f32 order as u32 order
            const unsigned long long kk = ((unsigned long long)key << 32) | p;
            best = kk < best ? kk : best;
        }
#pragma unroll
        for (int sft = 32; sft >= 1; sft >>= 1) {
            const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)best, sft), hi = (uint32_t)__shfl_xor((int)(uint32_t)(best >> 32), sft);
            const unsigned long long o = ((unsigned long long)hi << 32) | lo;
            best = o < best ? o : best;
        }
        unsigned long long *slot = &s_best[step & 1u];
        if ((tid & 63u) == 0u) atomicMin(slot, best);
        TL_SYNC();
        cur = (uint32_t)(*slot & 0xFFFFFFFFull);
        if (tid == 0) {
            path[step] = cur;
            visited[cur] = 1;
            s_best[(step + 1u) & 1u] = ~0ull;
        }
        TL_SYNC();
    }
}

hipError_t launch_nn_seed_dm(const float *dm, uint32_t n, uint32_t *path, int lds_bytes, hipStream_t s)
{
    const size_t lds = ((size_t)n + 15u) & ~(size_t)15u;
    if (lds + 1024 > (size_t)lds_bytes) return hipErrorInvalidValue;  // callers check
    hipError_t e = allow_max_lds(reinterpret_cast<const void *>(k_nn_seed_dm));
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_nn_seed_dm, dim3(1), dim3(kLkNT), lds, s, dm, n, path);
    return hipGetLastError();
}

}  // namespace tl
