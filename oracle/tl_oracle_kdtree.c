/*
 * tl_oracle_kdtree.c — CPU ORACLE, kd-tree part (test infrastructure only; see tl_oracle.h).
 *
 * Restates the reference's kd-tree (timgluz/teeline src/tsp/kdtree.rs) in plain C so that the k-NN candidate lists
 * of lin_kernighan::build_candidates (lin_kernighan.rs:12-27) can be produced the way the reference produces them:
 *   - from_cities / build_subtree / partition_points   kdtree.rs:19-73   (median split, coord = depth % 2)
 *   - KDPoint::cmp_by_coord                            kdtree.rs:301-317 (relative-epsilon three-way compare)
 *   - KDNode::nearest                                  kdtree.rs:193-212 (node first, near branch, far branch iff
 *                                                                         search_radius() > split distance)
 *   - NearestResult::add / search_radius               mod.rs:1839-1889  (insert iff d < radius, after equal distances)
 *
 * One thing cannot be restated: partition_points uses `select_nth_unstable_by` (kdtree.rs:63), whose arrangement of
 * elements that compare Equal is an implementation detail of the Rust standard library.  The tree — and with it the
 * visiting order that breaks exact distance ties in the k-buffer — is unique exactly when no two points of a subtree
 * that straddle its median compare Equal on the split coordinate.  This file's rule for that case (shared with the GPU
 * product, teeline_amd/csrc/kdtree.hip): the points of a subtree are ordered by (exact coordinate value, position in the
 * city array) and the element at len/2 is the pivot.  `*tie_free` reports whether the rule was ever exercised:
 * tie_free == 1 means the tree built here is the reference's tree, whatever its select implementation.
 */
#include "tl_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    uint32_t pos;        /* index into the caller's city array (= KDPoint.id of build_candidates' cities) */
    int32_t left, right; /* node indices, -1 = None */
    uint32_t depth;
} kdnode;

typedef struct {
    const float *xy;
    kdnode *nodes;
    uint32_t n_nodes;
    int tie_free;
} kdtree;

/* kdtree.rs:301-317: -1 Less, 0 Equal, +1 Greater; tol = max(|a|,|b|) * f32::EPSILON */
static int cmp_coord(float a, float b)
{
    const float tol = fmaxf(fabsf(a), fabsf(b)) * 1.1920929e-07f;
    if (fabsf(a - b) <= tol) return 0;
    return a < b ? -1 : 1;
}

/* (exact coordinate value, position) order: a total order, so the result does not depend on the incoming arrangement */
static int before(const float *xy, uint32_t p, uint32_t q, int coord)
{
    const float a = xy[2 * p + coord], b = xy[2 * q + coord];
    if (a < b) return 1;
    if (b < a) return 0;
    return p < q;
}

static void msort(const float *xy, uint32_t *v, uint32_t *tmp, uint32_t len, int coord)
{
    if (len < 2) return;
    const uint32_t h = len / 2;
    msort(xy, v, tmp, h, coord);
    msort(xy, v + h, tmp, len - h, coord);
    uint32_t a = 0, b = h, w = 0;
    while (a < h && b < len) {
        if (before(xy, v[b], v[a], coord)) tmp[w++] = v[b++];
        else tmp[w++] = v[a++];
    }
    while (a < h) tmp[w++] = v[a++];
    while (b < len) tmp[w++] = v[b++];
    memcpy(v, tmp, (size_t)len * sizeof(uint32_t));
}

/* kdtree.rs:36-73 build_subtree + partition_points.  `pts` is consumed (reordered). */
static int32_t build_subtree(kdtree *t, uint32_t *pts, uint32_t *tmp, uint32_t len, uint32_t depth)
{
    if (len == 0) return -1;                          /* :37-39 */
    const int32_t me = (int32_t)t->n_nodes++;
    kdnode *nd = &t->nodes[me];
    nd->depth = depth;
    nd->left = nd->right = -1;
    if (len == 1) {                                   /* :41-43 leaf */
        nd->pos = pts[0];
        return me;
    }
    const int coord = (int)(depth % 2);               /* :60 */
    const uint32_t pivot = len / 2;                   /* :61 */
    msort(t->xy, pts, tmp, len, coord);               /* :63 select_nth_unstable_by — see the header */
    /* the selection is unique iff the pivot compares unequal to both of its sorted neighbours */
    const float pv = t->xy[2 * pts[pivot] + coord];
    if (cmp_coord(t->xy[2 * pts[pivot - 1] + coord], pv) == 0) t->tie_free = 0;
    if (pivot + 1 < len && cmp_coord(t->xy[2 * pts[pivot + 1] + coord], pv) == 0) t->tie_free = 0;
    nd->pos = pts[pivot];                             /* :67 */
    const int32_t l = build_subtree(t, pts, tmp, pivot, depth + 1);                       /* :70 left = [0, pivot) */
    const int32_t r = build_subtree(t, pts + pivot + 1, tmp, len - pivot - 1, depth + 1); /* :68 right = (pivot, len) */
    t->nodes[me].left = l;
    t->nodes[me].right = r;
    return me;
}

typedef struct {
    float qx, qy;
    uint64_t qid;        /* NearestResult.target.id */
    const uint64_t *ids; /* KDPoint.id per position; NULL = position */
    uint32_t k, cnt;
    float *bd;
    uint32_t *bp;
} knnacc;

/* mod.rs:1839-1860 NearestResult::add, :1882-1889 search_radius */
static float search_radius(const knnacc *a) { return a->cnt < a->k ? INFINITY : a->bd[a->cnt - 1]; }

static void acc_add(knnacc *a, uint32_t pos, float d)
{
    const uint64_t id = a->ids ? a->ids[pos] : (uint64_t)pos;
    if (a->k == 0 || id == a->qid) return;            /* :1840-1842 */
    if (d < search_radius(a)) {                       /* :1848 */
        uint32_t ins = 0;
        while (ins < a->cnt && a->bd[ins] <= d) ++ins; /* :1851 partition_point(|r| r.distance <= new_distance) */
        for (uint32_t s = a->cnt; s > ins; --s) { a->bd[s] = a->bd[s - 1]; a->bp[s] = a->bp[s - 1]; }
        a->bd[ins] = d;
        a->bp[ins] = pos;
        if (a->cnt < a->k) ++a->cnt;                   /* :1854 truncate(n) */
    }
}

/* kdtree.rs:193-212 */
static void node_nearest(const kdtree *t, int32_t ni, knnacc *a)
{
    const kdnode *nd = &t->nodes[ni];
    const float px = t->xy[2 * nd->pos], py = t->xy[2 * nd->pos + 1];
    acc_add(a, nd->pos, tlo_dist(px, py, a->qx, a->qy));                     /* :194 */
    const int coord = (int)(nd->depth % 2);
    const float pc = coord ? py : px, qc = coord ? a->qy : a->qx;
    const int c = cmp_coord(pc, qc);                                          /* :196 self.point.cmp_by_coord(target) */
    const int32_t closest = c > 0 ? nd->left : nd->right;                     /* :198-199 Greater -> (left, right) */
    const int32_t further = c > 0 ? nd->right : nd->left;
    if (closest >= 0) node_nearest(t, closest, a);                            /* :202-204 */
    const float split = fabsf(pc - qc);                                       /* :206, kdtree.rs:297-299 */
    if (search_radius(a) > split && further >= 0) node_nearest(t, further, a); /* :207-211 */
}

static int tree_build(kdtree *t, const float *xy, uint32_t n)
{
    t->xy = xy;
    t->n_nodes = 0;
    t->tie_free = 1;
    t->nodes = (kdnode *)malloc((size_t)(n ? n : 1) * sizeof(kdnode));
    uint32_t *pts = (uint32_t *)malloc((size_t)(n ? n : 1) * sizeof(uint32_t));
    uint32_t *tmp = (uint32_t *)malloc((size_t)(n ? n : 1) * sizeof(uint32_t));
    if (!t->nodes || !pts || !tmp) { free(t->nodes); free(pts); free(tmp); t->nodes = NULL; return TLO_ERR_NOMEM; }
    for (uint32_t i = 0; i < n; ++i) pts[i] = i;      /* :26 points.to_vec(): input order */
    build_subtree(t, pts, tmp, n, 0);
    free(pts);
    free(tmp);
    return TLO_OK;
}

/* lin_kernighan.rs:12-27 build_candidates through the kd-tree: out = n x min(k, n-1) positions. */
int tlo_build_candidates_kdtree(const float *xy, uint32_t n, uint32_t k, uint32_t *out, int *tie_free)
{
    if (!xy || !out || n == 0) return TLO_ERR_BADARG;
    if (k > n - 1) k = n - 1;                          /* :14 */
    kdtree t;
    int rc = tree_build(&t, xy, n);
    if (rc) return rc;
    if (tie_free) *tie_free = t.tie_free;
    if (k == 0) { free(t.nodes); return TLO_OK; }
    knnacc a;
    a.ids = NULL;
    a.k = k;
    a.bd = (float *)malloc((size_t)(k + 1) * sizeof(float));
    a.bp = (uint32_t *)malloc((size_t)(k + 1) * sizeof(uint32_t));
    if (!a.bd || !a.bp) { free(a.bd); free(a.bp); free(t.nodes); return TLO_ERR_NOMEM; }
    for (uint32_t c = 0; c < n; ++c) {                 /* :18-25, cities in input order */
        a.qx = xy[2 * c];
        a.qy = xy[2 * c + 1];
        a.qid = c;
        a.cnt = 0;
        node_nearest(&t, 0, &a);
        for (uint32_t s = 0; s < k; ++s) out[(size_t)c * k + s] = s < a.cnt ? a.bp[s] : 0xFFFFFFFFu;
    }
    free(a.bd);
    free(a.bp);
    free(t.nodes);
    return TLO_OK;
}

/* KDTree::nearest(target, n) (kdtree.rs:116-122) for an arbitrary query point; ids may be NULL (id = position).
 * Returns the number of results; out_pos / out_dist hold them in buffer order. */
int tlo_kdtree_nearest(const float *xy, const uint64_t *ids, uint32_t n, float qx, float qy, uint64_t qid, uint32_t k,
                       uint32_t *out_pos, float *out_dist)
{
    if (n == 0 || k == 0) return 0;                    /* :24 empty tree; mod.rs:1840 n == 0 */
    kdtree t;
    if (tree_build(&t, xy, n)) return TLO_ERR_NOMEM;
    knnacc a;
    a.ids = ids;
    a.k = k;
    a.cnt = 0;
    a.qx = qx;
    a.qy = qy;
    a.qid = qid;
    a.bd = (float *)malloc((size_t)(k + 1) * sizeof(float));
    a.bp = (uint32_t *)malloc((size_t)(k + 1) * sizeof(uint32_t));
    if (!a.bd || !a.bp) { free(a.bd); free(a.bp); free(t.nodes); return TLO_ERR_NOMEM; }
    node_nearest(&t, 0, &a);
    for (uint32_t s = 0; s < a.cnt; ++s) {
        if (out_pos) out_pos[s] = a.bp[s];
        if (out_dist) out_dist[s] = a.bd[s];
    }
    const int cnt = (int)a.cnt;
    free(a.bd);
    free(a.bp);
    free(t.nodes);
    return cnt;
}

static void walk(const kdtree *t, int32_t ni, uint32_t *out, uint32_t *w)
{
    if (ni < 0) return;
    walk(t, t->nodes[ni].left, out, w);
    out[(*w)++] = t->nodes[ni].pos;
    walk(t, t->nodes[ni].right, out, w);
}

/* KDTree::walk (kdtree.rs:98-108): positions in in-order; also the root-to-leaf height. */
int tlo_kdtree_walk(const float *xy, uint32_t n, uint32_t *out_order, int *tie_free)
{
    if (n == 0) return TLO_OK;
    kdtree t;
    if (tree_build(&t, xy, n)) return TLO_ERR_NOMEM;
    uint32_t w = 0;
    walk(&t, 0, out_order, &w);
    if (tie_free) *tie_free = t.tie_free;
    free(t.nodes);
    return TLO_OK;
}
