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
//   host  : tree build — median split by coord = depth % 2 over points ordered by (exact coordinate value, position); where
//           no two points straddling a median compare Equal (kdtree.rs:301-317) this is the reference's tree whatever
//           its select_nth_unstable_by (kdtree.rs:63) does, elsewhere the reference's tree is implementation-defined and
//           this rule (the CPU oracle under oracle/ states the same one) is the specification.  O(n log n): two index
//           lists presorted by x and by y, split stably at every node.
//   device: one lane per city runs KDNode::nearest with an explicit stack (node, stage) in LDS: the node itself, the
//           near branch, and — decided only AFTER the near branch has returned, with the radius it left — the far branch
//           iff search_radius() > |split|.  Distances are the reference's correctly rounded f32 (tl::dist).
#include "tl_kernels.h"

#include <algorithm>
#include <vector>

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

struct Builder {
    const float *xy;
    std::vector<KdNode> &nodes;
    std::vector<uint32_t> sorted[2];  // the subtree's points in [lo, hi) of both lists: by (x, position) and by (y, position)
    std::vector<uint32_t> tmp;
    std::vector<unsigned char> side;
    bool tie_free = true;

    int32_t build(uint32_t lo, uint32_t hi, uint32_t depth)
    {
        if (lo >= hi) return -1;  // kdtree.rs:37-39
        const int32_t me = (int32_t)nodes.size();
        nodes.push_back(KdNode{});
        const uint32_t len = hi - lo, c = depth & 1u;
        KdNode nd{};
        nd.coord = c;
        nd.left = nd.right = -1;
        if (len == 1) {  // :41-43 leaf
            nd.pos = sorted[0][lo];
        } else {
            std::vector<uint32_t> &S = sorted[c], &O = sorted[c ^ 1u];
            const uint32_t mid = lo + len / 2u;  // :61 pivot_idx = len / 2
            const uint32_t pivot = S[mid];
            const float pv = xy[2 * pivot + c];
            if (cmp_coord(xy[2 * S[mid - 1] + c], pv) == 0) tie_free = false;
            if (mid + 1 < hi && cmp_coord(xy[2 * S[mid + 1] + c], pv) == 0) tie_free = false;
            for (uint32_t t = lo; t < mid; ++t) side[S[t]] = 0;
            side[pivot] = 1;
            for (uint32_t t = mid + 1; t < hi; ++t) side[S[t]] = 2;
            uint32_t wl = lo, wr = mid + 1;  // the other list, split stably into [lo, mid) | pivot | (mid, hi)
            for (uint32_t t = lo; t < hi; ++t) {
                const uint32_t p = O[t];
                if (side[p] == 0) tmp[wl++] = p;
                else if (side[p] == 2) tmp[wr++] = p;
            }
            tmp[mid] = pivot;
            std::copy(tmp.begin() + lo, tmp.begin() + hi, O.begin() + lo);
            nd.pos = pivot;  // :67
            nd.left = build(lo, mid, depth + 1);       // :70 points before the pivot
            nd.right = build(mid + 1, hi, depth + 1);  // :68 split_off(pivot_idx + 1)
        }
        nd.x = xy[2 * nd.pos];
        nd.y = xy[2 * nd.pos + 1];
        nodes[me] = nd;
        return me;
    }
};

}  // namespace

// kdtree::from_cities (kdtree.rs:19-34): node 0 is the root (n >= 1).  Returns tie_free (see the header comment).
bool kdtree_build_host(const float *xy, uint32_t n, std::vector<KdNode> &nodes)
{
    nodes.clear();
    nodes.reserve(n);
    Builder b{xy, nodes, {}, {}, {}};
    for (int c = 0; c < 2; ++c) {
        b.sorted[c].resize(n);
        for (uint32_t i = 0; i < n; ++i) b.sorted[c][i] = i;
        std::sort(b.sorted[c].begin(), b.sorted[c].end(), [&](uint32_t p, uint32_t q) {
            const float a = xy[2 * p + c], d = xy[2 * q + c];
            if (a < d) return true;
            if (d < a) return false;
            return p < q;
        });
    }
    b.tmp.resize(n);
    b.side.resize(n);
    b.build(0, n, 0);
    return b.tie_free;
}

namespace {

template <int KMAX>
__global__ __launch_bounds__(256) void k_knn_kdtree(const KdNode *__restrict__ nodes, const float2 *__restrict__ xy, uint32_t n, uint32_t k,
                                                    uint32_t *__restrict__ cand)
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
    stk[0][tid] = 0u;  // (root << 1) | stage 0
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
    const uint32_t grid = (n + 255u) / 256u;
    if (k <= 4) hipLaunchKernelGGL(k_knn_kdtree<4>, dim3(grid), dim3(256), 0, s, nodes, xy, n, k, cand);
    else if (k <= 8) hipLaunchKernelGGL(k_knn_kdtree<8>, dim3(grid), dim3(256), 0, s, nodes, xy, n, k, cand);
    else hipLaunchKernelGGL(k_knn_kdtree<16>, dim3(grid), dim3(256), 0, s, nodes, xy, n, k, cand);
    return hipGetLastError();
}

}  // namespace tl
