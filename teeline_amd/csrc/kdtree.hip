// kdtree.hip — lin_kernighan::build_candidates (lin_kernighan.rs:12-27) the way the reference does it: a kd-tree over the
// cities (kdtree.rs:19-73) and one k-NN query per city (kdtree.rs:193-212 + NearestResult, mod.rs:1839-1889).
//
// Why not the brute-force scan (k_knn, lk.hip): both give "the k nearest, ascending f32 distance" — but where two candidates
// of a city are at the SAME f32 distance the k-buffer keeps them in the order they were offered (insert after equals,
// mod.rs:1851), i.e. in the TREE'S VISITING ORDER, and a candidate at exactly the k-th distance is refused (d < radius,
// :1848), so the visiting order also decides who stays in the buffer.  On lattice instances (a280: 198 of 280 cities at
// k = 5) a scan in position order returns different lists, and Lin-Kernighan walks its candidates in list order: different
// tours.  So the product walks the same tree in the same order.
//
//   build : median split by coord = depth % 2 over points ordered by (exact coordinate value, position); where no two points
//           straddling a median compare Equal (kdtree.rs:301-317) this is the reference's tree whatever its
//           select_nth_unstable_by (kdtree.rs:63) does, elsewhere the reference's tree is implementation-defined and this rule
//           (the CPU oracle under oracle/ states the same one) is the specification.  On the device, level by level: two
//           index lists presorted by (x, position) and (y, position) (64-bit radix sort), and per level every segment of the
//           list sorted by the level's coordinate yields its pivot (the element at len / 2) while the other list is split
//           stably into left | pivot | right with one prefix sum over "goes left" flags.  A node's index is its pivot's
//           final place in the lists, so children are the midpoints of the two sub-ranges.
//   device: one lane per city runs KDNode::nearest with an explicit stack (node, stage) in LDS: the node itself, the
//           near branch, and — decided only AFTER the near branch has returned, with the radius it left — the far branch
//           iff search_radius() > |split|.  Distances are the reference's correctly rounded f32 (tl::dist).
#include "tl_kernels.h"

#include <hipcub/hipcub.hpp>

#pragma clang fp contract(off)

namespace tl {

namespace {

constexpr int kKdStack = 40;  // tree height <= 33 for n < 2^32 (median split); one u32 per level and lane

// kdtree.rs:301-317: -1 Less, 0 Equal, +1 Greater with tol = max(|a|, |b|) * f32::EPSILON
__host__ __device__ __forceinline__ int cmp_coord(float a, float b)
{
    const float tol = fmaxf(fabsf(a), fabsf(b)) * 1.1920929e-07f;
    if (fabsf(a - b) <= tol) return 0;
    return a < b ? -1 : 1;
}

// ---------------------------------------------------------------------------------------------- device tree build
// Workspace layout (u32 words unless noted), n elements each: S[0] (by x), S[1] (by y), Onew, seg_lo[2], seg_hi[2] (double
// buffered per-position segment bounds, hi == lo: position finished), side (per POINT), flag, scan; u64: keys in / out.
struct KdBuildWs {
    unsigned long long *keys_in, *keys_out;
    uint32_t *S[2], *Onew, *seg_lo[2], *seg_hi[2], *side, *flag, *scan, *pivpos;
};

__global__ __launch_bounds__(256) void k_kd_keys(const float2 *__restrict__ xy, uint32_t n, int coord, unsigned long long *__restrict__ keys)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const float v = (coord ? xy[i].y : xy[i].x) + 0.0f;  // -0.0 -> +0.0: the order is the float order (a < b), ties by position
    const uint32_t b = __builtin_bit_cast(uint32_t, v);
    const uint32_t sortable = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
    keys[i] = ((unsigned long long)sortable << 32) | i;
}

__global__ __launch_bounds__(256) void k_kd_unpack(const unsigned long long *__restrict__ keys, uint32_t n, uint32_t *__restrict__ S, uint32_t *__restrict__ seg_lo,
                                                   uint32_t *__restrict__ seg_hi)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    S[i] = (uint32_t)(keys[i] & 0xFFFFFFFFull);
    if (seg_lo) {  // level 0: one segment [0, n)
        seg_lo[i] = 0u;
        seg_hi[i] = n;
    }
}

// level step 1 — over the positions of the list sorted by this level's coordinate: every segment's element at len / 2 is
// its pivot (kdtree.rs:61-67) and becomes node `mid`; every point learns its side
__global__ __launch_bounds__(256) void k_kd_pivots(const float2 *__restrict__ xy, uint32_t n, uint32_t coord, const uint32_t *__restrict__ Sc,
                                                   const uint32_t *__restrict__ seg_lo, const uint32_t *__restrict__ seg_hi, uint32_t *__restrict__ side,
                                                   KdNode *__restrict__ nodes)
{
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= n) return;
    const uint32_t lo = seg_lo[t], hi = seg_hi[t];
    if (hi <= lo) return;  // finished (a pivot of an earlier level)
    const uint32_t mid = lo + (hi - lo) / 2u;
    const uint32_t p = Sc[t];
    side[p] = t < mid ? 0u : (t == mid ? 1u : 2u);
    if (t == mid) {
        KdNode nd;
        nd.x = xy[p].x;
        nd.y = xy[p].y;
        nd.pos = p;
        nd.coord = coord;
        nd.left = mid > lo ? (int32_t)(lo + (mid - lo) / 2u) : -1;                  // :70 points before the pivot
        nd.right = hi > mid + 1u ? (int32_t)(mid + 1u + (hi - mid - 1u) / 2u) : -1;  // :68 split_off(pivot_idx + 1)
        nodes[mid] = nd;
    }
}

// level step 2 — over the positions of the OTHER list: "goes left" flags for the prefix sum, and where the pivot sits
__global__ __launch_bounds__(256) void k_kd_flags(uint32_t n, const uint32_t *__restrict__ O, const uint32_t *__restrict__ seg_lo,
                                                  const uint32_t *__restrict__ seg_hi, const uint32_t *__restrict__ side, uint32_t *__restrict__ flag,
                                                  uint32_t *__restrict__ pivpos)
{
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= n) return;
    const uint32_t lo = seg_lo[t], hi = seg_hi[t];
    uint32_t f = 0u;
    if (hi > lo) {
        const uint32_t sd = side[O[t]];
        f = sd == 0u ? 1u : 0u;
        if (sd == 1u) pivpos[lo] = t;
    }
    flag[t] = f;
}

// level step 3 — the other list, split stably into [lo, mid) | pivot | (mid, hi), and the next level's segment bounds
__global__ __launch_bounds__(256) void k_kd_split(uint32_t n, const uint32_t *__restrict__ O, const uint32_t *__restrict__ seg_lo, const uint32_t *__restrict__ seg_hi,
                                                  const uint32_t *__restrict__ side, const uint32_t *__restrict__ scan, const uint32_t *__restrict__ pivpos,
                                                  uint32_t *__restrict__ Onew, uint32_t *__restrict__ nlo, uint32_t *__restrict__ nhi)
{
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= n) return;
    const uint32_t lo = seg_lo[t], hi = seg_hi[t];
    const uint32_t p = O[t];
    if (hi <= lo) {  // finished positions keep their point and stay finished
        Onew[t] = p;
        nlo[t] = t;
        nhi[t] = t;
        return;
    }
    const uint32_t mid = lo + (hi - lo) / 2u;
    const uint32_t lrank = scan[t] - scan[lo];  // "goes left" elements of the segment before t
    const uint32_t sd = side[p];
    uint32_t dst;
    if (sd == 0u) dst = lo + lrank;
    else if (sd == 1u) dst = mid;
    else dst = mid + 1u + ((t - lo) - lrank - (pivpos[lo] < t ? 1u : 0u));
    Onew[dst] = p;
    // bounds of the NEXT level, written for position t itself (both lists share the ranges): left part, pivot, right part
    if (t < mid) { nlo[t] = lo; nhi[t] = mid; }
    else if (t == mid) { nlo[t] = t; nhi[t] = t; }
    else { nlo[t] = mid + 1u; nhi[t] = hi; }
}

}  // namespace

size_t kdtree_build_ws_bytes(uint32_t n, size_t *cub_bytes_out)
{
    size_t sort_b = 0, scan_b = 0;
    (void)hipcub::DeviceRadixSort::SortKeys(nullptr, sort_b, (const unsigned long long *)nullptr, (unsigned long long *)nullptr, (int)n, 0, 64, nullptr);
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, scan_b, (const uint32_t *)nullptr, (uint32_t *)nullptr, (int)n, nullptr);
    const size_t cub_b = ((sort_b > scan_b ? sort_b : scan_b) + 255) & ~(size_t)255;
    if (cub_bytes_out) *cub_bytes_out = cub_b;
    const size_t np = ((size_t)n + 63) & ~(size_t)63;
    return cub_b + np * 8 * 2 + np * 4 * 12 + 256;
}

// kdtree::from_cities (kdtree.rs:19-34) on the device.  `ws` has kdtree_build_ws_bytes(n) bytes, `nodes` n entries; the root
// is node n / 2.  Everything is enqueued on `s`; nothing is waited for.
hipError_t kdtree_build_dev(const float2 *xy, uint32_t n, void *ws, KdNode *nodes, hipStream_t s)
{
    size_t cub_b = 0;
    (void)kdtree_build_ws_bytes(n, &cub_b);
    const size_t np = ((size_t)n + 63) & ~(size_t)63;
    unsigned char *w = (unsigned char *)ws;
    void *cub_tmp = w;
    w += cub_b;
    KdBuildWs B;
    B.keys_in = (unsigned long long *)w; w += np * 8;
    B.keys_out = (unsigned long long *)w; w += np * 8;
    uint32_t **slots[] = {&B.S[0], &B.S[1], &B.Onew, &B.seg_lo[0], &B.seg_lo[1], &B.seg_hi[0], &B.seg_hi[1], &B.side, &B.flag, &B.scan, &B.pivpos};
    for (uint32_t **sl : slots) { *sl = (uint32_t *)w; w += np * 4; }
    const dim3 grid((n + 255u) / 256u), blk(256);
    hipError_t e;
    for (int c = 0; c < 2; ++c) {
        hipLaunchKernelGGL(k_kd_keys, grid, blk, 0, s, xy, n, c, B.keys_in);
        size_t tb = cub_b;
        if ((e = hipcub::DeviceRadixSort::SortKeys(cub_tmp, tb, B.keys_in, B.keys_out, (int)n, 0, 64, s)) != hipSuccess) return e;
        hipLaunchKernelGGL(k_kd_unpack, grid, blk, 0, s, B.keys_out, n, B.S[c], c == 0 ? B.seg_lo[0] : (uint32_t *)nullptr, c == 0 ? B.seg_hi[0] : (uint32_t *)nullptr);
    }
    uint32_t levels = 1;  // height of a median-split tree: the smallest h with 2^h > n
    while ((1ull << levels) <= (unsigned long long)n) ++levels;
    int cur = 0;
    for (uint32_t d = 0; d < levels; ++d) {
        const uint32_t c = d & 1u;
        uint32_t *Sc = B.S[c], *O = B.S[c ^ 1u];
        hipLaunchKernelGGL(k_kd_pivots, grid, blk, 0, s, xy, n, c, Sc, B.seg_lo[cur], B.seg_hi[cur], B.side, nodes);
        hipLaunchKernelGGL(k_kd_flags, grid, blk, 0, s, n, O, B.seg_lo[cur], B.seg_hi[cur], B.side, B.flag, B.pivpos);
        size_t tb = cub_b;
        if ((e = hipcub::DeviceScan::ExclusiveSum(cub_tmp, tb, B.flag, B.scan, (int)n, s)) != hipSuccess) return e;
        hipLaunchKernelGGL(k_kd_split, grid, blk, 0, s, n, O, B.seg_lo[cur], B.seg_hi[cur], B.side, B.scan, B.pivpos, B.Onew, B.seg_lo[cur ^ 1], B.seg_hi[cur ^ 1]);
        // the split list replaces the other list (pointer swap: the next level reads it as Sc)
        B.S[c ^ 1u] = B.Onew;
        B.Onew = O;
        cur ^= 1;
    }
    return hipGetLastError();
}

namespace {

template <int KMAX>
__global__ __launch_bounds__(256) void k_knn_kdtree(const KdNode *__restrict__ nodes, const float2 *__restrict__ xy, uint32_t n, uint32_t k,
                                                    uint32_t *__restrict__ cand, uint32_t root)
{
    __shared__ uint32_t stk[kKdStack][256];  // [level][lane]: conflict-free, 40 KB
    const uint32_t tid = threadIdx.x, c = blockIdx.x * 256u + tid;
    if (c >= n) return;  // no barrier below
    const float2 q = xy[c];
    float bd[KMAX];
    uint32_t bp[KMAX];
#pragma unroll
    for (int t = 0; t < KMAX; ++t) {
        bd[t] = __builtin_inff();
        bp[t] = 0xFFFFFFFFu;
    }
    float radius = __builtin_inff();  // search_radius(): INFINITY until the buffer holds k, then the k-th kept distance
    int sp = 0;
    stk[0][tid] = root << 1;  // (root << 1) | stage 0
    while (sp >= 0) {
        const uint32_t e = stk[sp][tid];
        const KdNode nd = nodes[e >> 1];
        const float pc = nd.coord ? nd.y : nd.x, qc = nd.coord ? q.y : q.x;
        const bool greater = cmp_coord(pc, qc) > 0;  // kdtree.rs:196-200: Greater -> (left, right), else (right, left)
        if ((e & 1u) == 0u) {
            // ---- first visit: acc.add(self.point, distance) (:194), then the near branch (:202-204)
            if (nd.pos != c) {  // self excluded by id (mod.rs:1840-1842); k >= 1 here
                const float d = dist(make_float2(nd.x, nd.y), q);
                if (d < radius) {  // mod.rs:1848; insert at partition_point(distance <= d): after equals, then truncate(k)
                    float cd = d;
                    uint32_t cp = nd.pos;
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
                }
            }
            stk[sp][tid] = e | 1u;
            const int32_t closest = greater ? nd.left : nd.right;
            if (closest >= 0 && sp + 1 < kKdStack) stk[++sp][tid] = (uint32_t)closest << 1;
        } else {
            // ---- back from the near branch: the far one iff search_radius() > split_distance (:206-211)
            --sp;
            const int32_t further = greater ? nd.right : nd.left;
            const float split = fabsf(pc - qc);  // kdtree.rs:297-299
            if (radius > split && further >= 0) stk[++sp][tid] = (uint32_t)further << 1;
        }
    }
#pragma unroll
    for (int t = 0; t < KMAX; ++t)
        if ((uint32_t)t < k) cand[(size_t)c * k + t] = bp[t];
}

}  // namespace

hipError_t launch_knn_kdtree(const KdNode *nodes, const float2 *xy, uint32_t n, uint32_t k, uint32_t *cand, hipStream_t s)
{
    const uint32_t grid = (n + 255u) / 256u, root = n / 2u;  // the level-0 segment [0, n) has its pivot at n / 2
    if (k <= 4) hipLaunchKernelGGL(k_knn_kdtree<4>, dim3(grid), dim3(256), 0, s, nodes, xy, n, k, cand, root);
    else if (k <= 8) hipLaunchKernelGGL(k_knn_kdtree<8>, dim3(grid), dim3(256), 0, s, nodes, xy, n, k, cand, root);
    else if (k <= 16) hipLaunchKernelGGL(k_knn_kdtree<16>, dim3(grid), dim3(256), 0, s, nodes, xy, n, k, cand, root);
    else if (k <= 32) hipLaunchKernelGGL(k_knn_kdtree<32>, dim3(grid), dim3(256), 0, s, nodes, xy, n, k, cand, root);
    else hipLaunchKernelGGL(k_knn_kdtree<64>, dim3(grid), dim3(256), 0, s, nodes, xy, n, k, cand, root);  // (round 5: lists of up to 64)
    return hipGetLastError();
}

}  // namespace tl
