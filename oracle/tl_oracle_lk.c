/*
 * tl_oracle_lk.c — CPU ORACLE, Lin–Kernighan part (test infrastructure only; see tl_oracle.h).
 *
 * Restates src/tsp/lin_kernighan.rs of the reference (timgluz/teeline) in plain C, working on
 * positions 0..n-1 (the reference indexes its next/prev/used/candidates arrays by city id and
 * never orders by id, so positions are an exact stand-in).
 *
 * Known, documented divergences:
 *  - candidate lists: the reference queries a kd-tree (restated in tl_oracle_kdtree.c); its tree is
 *    implementation-defined only where select_nth_unstable_by (kdtree.rs:63) meets Equal elements around a median,
 *    for which the oracle fixes the (coordinate value, position) order.  tlo_build_candidates below is the
 *    brute-force scan in (distance, position) order: the same lists wherever no two candidates of a city are at
 *    the same f32 distance; it serves the NN seed's rule and as a cross-check.
 *  - kicks: the reference draws from an unseeded thread RNG (lin_kernighan.rs:73); the oracle
 *    takes an explicit seed (splitmix64) so results are reproducible.
 */
#include "tl_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define LK_EPS 1e-6f /* lin_kernighan.rs:252 */

typedef struct {
    const float *xy;
    uint32_t n;
    const uint32_t *cand; /* n * k */
    uint32_t k;
    uint32_t max_depth;
    const uint32_t *next, *prev;
    unsigned char *used;
    uint32_t *chain;
    uint32_t clen;
    uint64_t nodes;
} lkctx;

/* lin_kernighan.rs:104-106: dm.distance_between(a,b) on a matrix rebuilt from the coordinates (:41) */
static inline float dd(const lkctx *c, uint32_t a, uint32_t b)
{
    if (a == b) return 0.0f;
    return tlo_dist(c->xy[2 * a], c->xy[2 * a + 1], c->xy[2 * b], c->xy[2 * b + 1]);
}

/* lin_kernighan.rs:12-27 (brute-force stand-in for kd-tree k-NN; mod.rs:1839-1860 buffer rule) */
int tlo_build_candidates(const float *xy, uint32_t n, uint32_t k, uint32_t *out)
{
    if (!xy || !out || n == 0) return TLO_ERR_BADARG;
    if (k > n - 1) k = n - 1; /* :14 */
    if (k == 0) return TLO_OK;
    float *bd = (float *)malloc((size_t)(k + 1) * sizeof(float));
    uint32_t *bp = (uint32_t *)malloc((size_t)(k + 1) * sizeof(uint32_t));
    if (!bd || !bp) { free(bd); free(bp); return TLO_ERR_NOMEM; }
    for (uint32_t c = 0; c < n; ++c) {
        uint32_t cnt = 0;
        for (uint32_t p = 0; p < n; ++p) {
            if (p == c) continue;
            float d = tlo_dist(xy[2 * p], xy[2 * p + 1], xy[2 * c], xy[2 * c + 1]); /* kdtree.rs:194 */
            float radius = (cnt < k) ? INFINITY : bd[cnt - 1];
            if (d < radius) {
                uint32_t ins = 0;
                while (ins < cnt && bd[ins] <= d) ++ins;
                for (uint32_t t = cnt; t > ins; --t) { bd[t] = bd[t - 1]; bp[t] = bp[t - 1]; }
                bd[ins] = d;
                bp[ins] = p;
                if (cnt < k) ++cnt;
            }
        }
        for (uint32_t t = 0; t < k; ++t) out[(size_t)c * k + t] = bp[t];
    }
    free(bd);
    free(bp);
    return TLO_OK;
}

/* lin_kernighan.rs:175-177 */
static inline int is_tour_edge(const lkctx *c, uint32_t a, uint32_t b)
{
    return c->next[a] == b || c->prev[a] == b;
}

/* lin_kernighan.rs:265-340 */
static int find_lk_chain(lkctx *c, uint32_t t1, uint32_t t_open, float gain, uint32_t depth)
{
    ++c->nodes;
    if (depth >= 1) {
        float close_gain = gain - dd(c, t_open, t1);
        if (close_gain > LK_EPS) return 1;
    }
    if (depth >= c->max_depth) return 0;
    for (uint32_t q = 0; q < c->k; ++q) {
        uint32_t t_next = c->cand[(size_t)t_open * c->k + q];
        float g1 = gain - dd(c, t_open, t_next);
        if (g1 <= LK_EPS) break;
        if (c->used[t_next]) continue;
        if (is_tour_edge(c, t_open, t_next)) continue;
        uint32_t t_break = c->next[t_next];
        if (c->used[t_break]) continue;
        float g2 = g1 + dd(c, t_next, t_break);
        c->chain[c->clen++] = t_next;
        c->used[t_next] = 1;
        c->chain[c->clen++] = t_break;
        c->used[t_break] = 1;
        if (find_lk_chain(c, t1, t_break, g2, depth + 1)) return 1;
        --c->clen;
        c->used[t_break] = 0;
        --c->clen;
        c->used[t_next] = 0;
    }
    return 0;
}

typedef struct { uint32_t v[6]; uint32_t len; } adjl;

static void adj_retain_ne(adjl *a, uint32_t x)
{
    uint32_t w = 0;
    for (uint32_t r = 0; r < a->len; ++r)
        if (a->v[r] != x) a->v[w++] = a->v[r];
    a->len = w;
}

static void adj_push(adjl *a, uint32_t x)
{
    if (a->len < 6) a->v[a->len] = x;
    a->len++;
}

/* lin_kernighan.rs:181-250 */
static int chain_is_valid_tour(const uint32_t *chain, uint32_t clen, const uint32_t *next,
                               const uint32_t *city_ids, uint32_t n, adjl *adj, uint32_t *prev_map)
{
    if (clen < 4) return 0;
    uint32_t k = clen / 2;
    for (uint32_t t = 0; t < n; ++t) prev_map[next[city_ids[t]]] = city_ids[t];
    for (uint32_t t = 0; t < n; ++t) {
        uint32_t c = city_ids[t];
        adj[c].len = 0;
        adj_push(&adj[c], prev_map[c]);
        adj_push(&adj[c], next[c]);
    }
    for (uint32_t i = 0; i < k; ++i) {
        uint32_t a = chain[2 * i], b = chain[2 * i + 1];
        adj_retain_ne(&adj[a], b);
        adj_retain_ne(&adj[b], a);
    }
    for (uint32_t i = 0; i + 1 < k; ++i) {
        uint32_t a = chain[2 * i + 1], b = chain[2 * i + 2];
        adj_push(&adj[a], b);
        adj_push(&adj[b], a);
    }
    {
        uint32_t a = chain[clen - 1], b = chain[0];
        adj_push(&adj[a], b);
        adj_push(&adj[b], a);
    }
    for (uint32_t t = 0; t < n; ++t)
        if (adj[city_ids[t]].len != 2) return 0;
    uint32_t start = chain[0];
    uint32_t prev_c = adj[start].v[0];
    uint32_t current = adj[start].v[1];
    uint32_t count = 1;
    while (current != start) {
        ++count;
        if (count > n) return 0;
        uint32_t next_c = (adj[current].v[0] != prev_c) ? adj[current].v[0] : adj[current].v[1];
        prev_c = current;
        current = next_c;
    }
    return count == n;
}

/* lin_kernighan.rs:397-450 */
static void apply_lk_chain(const uint32_t *chain, uint32_t clen, uint32_t *tour, uint32_t n, adjl *adj)
{
    uint32_t k = clen / 2;
    for (uint32_t t = 0; t < n; ++t) adj[t].len = 0;
    for (uint32_t i = 0; i < n; ++i) {
        uint32_t a = tour[i], b = tour[(i + 1) % n];
        adj_push(&adj[a], b);
        adj_push(&adj[b], a);
    }
    for (uint32_t i = 0; i < k; ++i) {
        uint32_t a = chain[2 * i], b = chain[2 * i + 1];
        adj_retain_ne(&adj[a], b);
        adj_retain_ne(&adj[b], a);
    }
    for (uint32_t i = 0; i + 1 < k; ++i) {
        uint32_t a = chain[2 * i + 1], b = chain[2 * i + 2];
        adj_push(&adj[a], b);
        adj_push(&adj[b], a);
    }
    {
        uint32_t a = chain[clen - 1], b = chain[0];
        adj_push(&adj[a], b);
        adj_push(&adj[b], a);
    }
    uint32_t start = tour[0];
    uint32_t prev_city = UINT32_MAX; /* :433 sentinel */
    uint32_t current = start;
    for (uint32_t rank = 0; rank < n; ++rank) {
        tour[rank] = current;
        uint32_t next_city = current; /* :440 unwrap_or(current) */
        for (uint32_t q = 0; q < adj[current].len && q < 6; ++q)
            if (adj[current].v[q] != prev_city) { next_city = adj[current].v[q]; break; }
        prev_city = current;
        current = next_city;
    }
}

/* lin_kernighan.rs:345-389 find_lk_move: first (t1 in city_ids order, next-then-prev) pair whose first profitable chain
 * is a valid tour.  `used` must be all zero on entry and is all zero again on a miss.  Returns the chain length or 0. */
static uint32_t find_lk_move(const float *xy, uint32_t n, const uint32_t *cand, uint32_t k, uint32_t max_depth,
                             const uint32_t *next, const uint32_t *prev, const uint32_t *city_ids, unsigned char *used,
                             uint32_t *chain, adjl *adj, uint32_t *prev_map, tlo_stats *st)
{
    lkctx c = {xy, n, cand, k, max_depth, next, prev, used, chain, 0, 0};
    for (uint32_t ti = 0; ti < n; ++ti) { /* :357 */
        uint32_t t1 = city_ids[ti];
        uint32_t t2s[2] = {next[t1], prev[t1]};
        for (int o = 0; o < 2; ++o) { /* :359 */
            uint32_t t2 = t2s[o];
            float g0 = dd(&c, t1, t2);
            c.clen = 0;
            chain[c.clen++] = t1;
            chain[c.clen++] = t2;
            used[t1] = 1;
            used[t2] = 1;
            int found = find_lk_chain(&c, t1, t2, g0, 0);
            if (st) st->candidates += 1;
            if (found) {
                if (chain_is_valid_tour(chain, c.clen, next, city_ids, n, adj, prev_map)) return c.clen;
                for (uint32_t q = 0; q < c.clen; ++q) used[chain[q]] = 0; /* :378-380 */
            } else {
                used[t1] = 0;
                used[t2] = 0;
            }
        }
    }
    return 0;
}

/* lin_kernighan.rs:134-145 */
void tlo_flat_to_next_prev(const uint32_t *tour, uint32_t n, uint32_t *next, uint32_t *prev)
{
    for (uint32_t i = 0; i < n; ++i) {
        uint32_t a = tour[i], b = tour[(i + 1) % n];
        next[a] = b;
        prev[b] = a;
    }
}

/* lin_kernighan.rs:454-481 */
int tlo_lk_pass(const float *xy, uint32_t n, uint32_t *tour, const uint32_t *cand, uint32_t k,
                uint32_t max_depth, tlo_stats *st)
{
    if (n < 4) return 0;
    if (k > n - 1) k = n - 1;
    uint32_t *city_ids = (uint32_t *)malloc((size_t)n * sizeof(uint32_t));
    uint32_t *next = (uint32_t *)malloc((size_t)n * sizeof(uint32_t));
    uint32_t *prev = (uint32_t *)malloc((size_t)n * sizeof(uint32_t));
    uint32_t *prev_map = (uint32_t *)malloc((size_t)n * sizeof(uint32_t));
    unsigned char *used = (unsigned char *)calloc(n, 1);
    uint32_t *chain = (uint32_t *)malloc((size_t)(2 * max_depth + 4) * sizeof(uint32_t));
    adjl *adj = (adjl *)malloc((size_t)n * sizeof(adjl));
    int improved = 0;
    if (!city_ids || !next || !prev || !prev_map || !used || !chain || !adj) goto done;
    memcpy(city_ids, tour, (size_t)n * sizeof(uint32_t)); /* :466 fixed for the whole pass */

    for (;;) {
        tlo_flat_to_next_prev(tour, n, next, prev); /* :470 */
        memset(used, 0, n);
        const uint32_t clen = find_lk_move(xy, n, cand, k, max_depth, next, prev, city_ids, used, chain, adj, prev_map, st);
        if (st) st->sweeps += 1;
        if (!clen) break;
        apply_lk_chain(chain, clen, tour, n, adj);
        improved = 1;
        if (st) { st->moves += 1; st->reversed += clen / 2; }
    }
done:
    free(city_ids); free(next); free(prev); free(prev_map); free(used); free(chain); free(adj);
    return improved;
}

/* find_lk_move (:345-389) on a flat tour with city_ids = the tour, as the reference's unit tests call it
 * (lin_kernighan.rs:656-700, 768-812).  out_chain needs 2 * max_depth + 4 entries.  Returns the chain length or 0. */
int tlo_find_lk_move(const float *xy, uint32_t n, const uint32_t *tour, const uint32_t *cand, uint32_t k,
                     uint32_t max_depth, uint32_t *out_chain)
{
    if (n < 2) return 0;
    if (k > n - 1) k = n - 1;
    uint32_t *next = (uint32_t *)malloc((size_t)n * sizeof(uint32_t));
    uint32_t *prev = (uint32_t *)malloc((size_t)n * sizeof(uint32_t));
    uint32_t *prev_map = (uint32_t *)malloc((size_t)n * sizeof(uint32_t));
    unsigned char *used = (unsigned char *)calloc(n, 1);
    adjl *adj = (adjl *)malloc((size_t)n * sizeof(adjl));
    uint32_t clen = 0;
    if (next && prev && prev_map && used && adj) {
        tlo_flat_to_next_prev(tour, n, next, prev);
        clen = find_lk_move(xy, n, cand, k, max_depth, next, prev, tour, used, out_chain, adj, prev_map, NULL);
    }
    free(next); free(prev); free(prev_map); free(used); free(adj);
    return (int)clen;
}

/* apply_lk_chain (:397-450) on a flat tour. */
int tlo_apply_lk_chain(uint32_t *tour, uint32_t n, const uint32_t *chain, uint32_t clen)
{
    adjl *adj = (adjl *)malloc((size_t)n * sizeof(adjl));
    if (!adj) return TLO_ERR_NOMEM;
    apply_lk_chain(chain, clen, tour, n, adj);
    free(adj);
    return TLO_OK;
}

/* lin_kernighan.rs:485-499 */
void tlo_double_bridge(const uint32_t *tour, uint32_t n, uint32_t r1, uint32_t r2, uint32_t r3,
                       uint32_t *out)
{
    if (n < 8) { memcpy(out, tour, (size_t)n * sizeof(uint32_t)); return; }
    uint32_t p1 = 1 + r1, p2 = p1 + 1 + r2, p3 = p2 + 1 + r3;
    uint32_t w = 0;
    for (uint32_t t = 0; t < p1; ++t) out[w++] = tour[t];
    for (uint32_t t = p2; t < p3; ++t) out[w++] = tour[t];
    for (uint32_t t = p1; t < p2; ++t) out[w++] = tour[t];
    for (uint32_t t = p3; t < n; ++t) out[w++] = tour[t];
}

/* lin_kernighan.rs:118-122: (0..n).map(d(tour[i], tour[(i+1)%n])).sum() — sequential from 0 */
static float lk_tour_distance(const float *xy, const uint32_t *tour, uint32_t n)
{
    float s = 0.0f;
    for (uint32_t i = 0; i < n; ++i) {
        uint32_t a = tour[i], b = tour[(i + 1) % n];
        s += (a == b) ? 0.0f : tlo_dist(xy[2 * a], xy[2 * a + 1], xy[2 * b], xy[2 * b + 1]);
    }
    return s;
}

static inline uint64_t sm64(uint64_t *state)
{
    uint64_t z = (*state += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

/* lin_kernighan.rs:35-100; cand_in = precomputed candidate lists (n x min(n_nearest, n-1)) or NULL (brute force) */
/* snaps / snap_dist (optional): what the reference sends as PathUpdate(best_tour, best_dist) (:71, :90), in order — at most cap
 * tours of n positions; *count counts them all */
static int lk_solve_impl(const float *xy, const float *packed, uint32_t n, const uint32_t *init, uint32_t epochs,
                         uint32_t platoo_epochs, uint32_t n_nearest, uint32_t max_depth, uint64_t seed,
                         const uint32_t *cand_in, uint32_t *out, float *out_cost, tlo_stats *st,
                         uint32_t *snaps, float *snap_dist, uint32_t cap, uint32_t *count)
{
    uint32_t nsnap = 0;
    if (count) *count = 0;
    if (!xy || !out || n == 0) return TLO_ERR_BADARG;
    if (st) memset(st, 0, sizeof(*st));
    uint32_t k = n_nearest;
    if (k > n - 1) k = n - 1;
    uint32_t *cand = (uint32_t *)malloc((size_t)n * (k ? k : 1) * sizeof(uint32_t));
    uint32_t *candidate = (uint32_t *)malloc((size_t)n * sizeof(uint32_t));
    if (!cand || !candidate) { free(cand); free(candidate); return TLO_ERR_NOMEM; }
    if (cand_in) memcpy(cand, cand_in, (size_t)n * k * sizeof(uint32_t));
    else tlo_build_candidates_kdtree(xy, n, k, cand, NULL); /* :43 build_candidates = kd-tree k-NN (tl_oracle_kdtree.c) */

    if (init) memcpy(out, init, (size_t)n * sizeof(uint32_t)); /* :45-46 */
    else tlo_nearest_neighbor(packed ? NULL : xy, packed, n, 3, out, NULL); /* :47-55 nearest_neighbor::solve(problem):
                                                                                 problem.distances, default n_nearest = 3 */

    if (n >= 4) { /* :57-59 */
        tlo_lk_pass(xy, n, out, cand, k, max_depth, st); /* :61-68 */
        float best_dist = lk_tour_distance(xy, out, n);   /* :70 */
        if (snaps) { /* :71 send_progress */
            if (nsnap < cap) { memcpy(snaps + (size_t)nsnap * n, out, (size_t)n * sizeof(uint32_t)); snap_dist[nsnap] = best_dist; }
            ++nsnap;
        }
        uint64_t rng = seed;
        uint32_t platoo = 0;
        for (uint32_t e = 0; e < epochs; ++e) { /* :75 */
            uint32_t q = n / 4;
            uint32_t r1 = 0, r2 = 0, r3 = 0;
            if (n >= 8) { r1 = (uint32_t)(sm64(&rng) % q); r2 = (uint32_t)(sm64(&rng) % q); r3 = (uint32_t)(sm64(&rng) % q); }
            tlo_double_bridge(out, n, r1, r2, r3, candidate); /* :76 */
            tlo_lk_pass(xy, n, candidate, cand, k, max_depth, st);
            float dist = lk_tour_distance(xy, candidate, n);
            if (dist < best_dist) { /* :86 */
                memcpy(out, candidate, (size_t)n * sizeof(uint32_t));
                best_dist = dist;
                platoo = 0;
                if (snaps) { /* :90 send_progress */
                    if (nsnap < cap) { memcpy(snaps + (size_t)nsnap * n, out, (size_t)n * sizeof(uint32_t)); snap_dist[nsnap] = best_dist; }
                    ++nsnap;
                }
            } else {
                if (++platoo >= platoo_epochs) break; /* :92-95 */
            }
        }
    }
    free(cand);
    free(candidate);
    /* :99 Solution::new -> total through problem.distances.tour_length (the search above used the Euclidean matrix
     * rebuilt from the coordinates, :41) */
    if (out_cost) *out_cost = tlo_tour_length(packed ? NULL : xy, packed, n, out);
    if (count) *count = nsnap;
    return TLO_OK;
}

int tlo_lin_kernighan_cand(const float *xy, const float *packed, uint32_t n, const uint32_t *init, uint32_t epochs,
                           uint32_t platoo_epochs, uint32_t n_nearest, uint32_t max_depth, uint64_t seed,
                           const uint32_t *cand_in, uint32_t *out, float *out_cost, tlo_stats *st)
{
    return lk_solve_impl(xy, packed, n, init, epochs, platoo_epochs, n_nearest, max_depth, seed, cand_in, out, out_cost, st, NULL, NULL, 0, NULL);
}

/* the same solve with the progress messages of lin_kernighan.rs:71,90 recorded (tours as positions, best_dist) */
int tlo_lin_kernighan_trace(const float *xy, const float *packed, uint32_t n, const uint32_t *init, uint32_t epochs,
                            uint32_t platoo_epochs, uint32_t n_nearest, uint32_t max_depth, uint64_t seed,
                            uint32_t *out, float *out_cost, tlo_stats *st, uint32_t *snaps, float *snap_dist, uint32_t cap, uint32_t *count)
{
    if (!snaps || !snap_dist || !count) return TLO_ERR_BADARG;
    return lk_solve_impl(xy, packed, n, init, epochs, platoo_epochs, n_nearest, max_depth, seed, NULL, out, out_cost, st, snaps, snap_dist, cap, count);
}

int tlo_lin_kernighan(const float *xy, uint32_t n, const uint32_t *init, uint32_t epochs,
                      uint32_t platoo_epochs, uint32_t n_nearest, uint32_t max_depth, uint64_t seed,
                      uint32_t *out, float *out_cost, tlo_stats *st)
{
    return tlo_lin_kernighan_cand(xy, NULL, n, init, epochs, platoo_epochs, n_nearest, max_depth, seed, NULL, out, out_cost, st);
}
