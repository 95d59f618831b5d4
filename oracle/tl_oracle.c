/*
 * tl_oracle.c — CPU ORACLE (test infrastructure only; see tl_oracle.h for the contract).
 *
 * Plain-C restatement of the reference's 2-opt / 3-opt / nearest-neighbour / distance-matrix
 * code paths.  Every function cites the reference file:line it follows
 * (paths relative to the reference repo root, timgluz/teeline).
 *
 * Parity pins: tests/test_oracle_golden.py checks this file against every exact-value test the
 * reference holds for the path and against the reference's committed output numbers.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (oracle/Makefile).  x86-64 SSE2 float math has
 * no excess precision and sqrtss is correctly rounded, which matches Rust's f32 semantics.
 */
#include "tl_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------ */
/* distances                                                                                  */
/* ------------------------------------------------------------------------------------------ */

/* kdtree.rs:291-295  KDPoint::distance */
float tlo_dist(float x1, float y1, float x2, float y2)
{
    float dx = x1 - x2;
    float dy = y1 - y2;
    return sqrtf(dx * dx + dy * dy);
}

/* distance_matrix.rs:59-75  geo_distance (f64 trig, floor, cast to f32) */
static double geo_to_rad(float x)
{
    const double PI = 3.14159265358979323846264338327950288; /* std::f64::consts::PI */
    double deg = (double)truncf(x);
    double min = (double)(x - truncf(x));
    return PI * (deg + 5.0 * min / 3.0) / 180.0;
}

static float geo_distance(float x1, float y1, float x2, float y2)
{
    double lat1 = geo_to_rad(x1), lon1 = geo_to_rad(y1);
    double lat2 = geo_to_rad(x2), lon2 = geo_to_rad(y2);
    double q1 = cos(lon1 - lon2);
    double q2 = cos(lat1 - lat2);
    double q3 = cos(lat1 + lat2);
    const double RRR = 6378.388;
    return (float)floor(RRR * acos(0.5 * ((1.0 + q1) * q2 - (1.0 - q1) * q3)) + 1.0);
}

/* distance_matrix.rs:122-153  DistanceMatrix::build: for i, for j<i push d(cities[i], cities[j]) */
int tlo_dm_build_packed(const float *xy, uint32_t n, float *out)
{
    if (!xy || !out) return TLO_ERR_BADARG;
    if (n < 2) return TLO_ERR_BADARG; /* :124-126 "requires at least 2 points" */
    size_t w = 0;
    for (uint32_t i = 0; i < n; ++i)
        for (uint32_t j = 0; j < i; ++j)
            out[w++] = tlo_dist(xy[2 * i], xy[2 * i + 1], xy[2 * j], xy[2 * j + 1]);
    return TLO_OK;
}

int tlo_dm_build_packed_geo(const float *xy, uint32_t n, float *out)
{
    if (!xy || !out || n < 2) return TLO_ERR_BADARG;
    size_t w = 0;
    for (uint32_t i = 0; i < n; ++i)
        for (uint32_t j = 0; j < i; ++j)
            out[w++] = geo_distance(xy[2 * i], xy[2 * i + 1], xy[2 * j], xy[2 * j + 1]);
    return TLO_OK;
}

/* distance_matrix.rs:177-191  distance_by_pos: 0 if equal; idx = from*(from-1)/2 + to */
float tlo_dm_lookup(const float *packed, uint32_t p, uint32_t q)
{
    if (p == q) return 0.0f;
    uint64_t from = p > q ? p : q;
    uint64_t to = p > q ? q : p;
    return packed[from * (from - 1) / 2 + to];
}

int tlo_dm_expand_full(const float *packed, uint32_t n, float *full)
{
    if (!packed || !full) return TLO_ERR_BADARG;
    for (uint32_t i = 0; i < n; ++i)
        for (uint32_t j = 0; j < n; ++j)
            full[(size_t)i * n + j] = tlo_dm_lookup(packed, i, j);
    return TLO_OK;
}

/* one distance source for every solver below */
typedef struct {
    const float *xy;
    const float *packed;
} dsrc;

static inline float D(const dsrc *s, uint32_t p, uint32_t q)
{
    if (s->packed) return tlo_dm_lookup(s->packed, p, q);
    if (p == q) return 0.0f; /* distance_matrix.rs:198-200 */
    return tlo_dist(s->xy[2 * p], s->xy[2 * p + 1], s->xy[2 * q], s->xy[2 * q + 1]);
}

/* distance_matrix.rs:235-245  tour_length_by_pos */
float tlo_tour_length(const float *xy, const float *packed, uint32_t n, const uint32_t *perm)
{
    if (n < 2) return 0.0f; /* :236-238 */
    dsrc s = {xy, packed};
    float total = D(&s, perm[n - 1], perm[0]);
    for (uint32_t w = 0; w + 1 < n; ++w) total += D(&s, perm[w], perm[w + 1]);
    return total;
}

/* mod.rs:1620-1634 validate_tour */
int tlo_validate_tour(const uint32_t *perm, uint32_t n)
{
    unsigned char *seen = (unsigned char *)calloc(n ? n : 1, 1);
    if (!seen) return 0;
    int ok = 1;
    for (uint32_t i = 0; i < n && ok; ++i) {
        if (perm[i] >= n || seen[perm[i]]) ok = 0;
        else seen[perm[i]] = 1;
    }
    free(seen);
    return ok;
}

/* ------------------------------------------------------------------------------------------ */
/* "ref-faithful" cost model: SipHash-1-3 keyed hash map id -> position                        */
/* (Rust's std HashMap<usize,usize> default hasher; distance_matrix.rs:197-212 does two        */
/*  lookups per distance).  Same values as direct indexing; only used to time a CPU baseline   */
/*  that pays what the reference pays.                                                         */
/* ------------------------------------------------------------------------------------------ */
#define ROTL64(x, b) (((x) << (b)) | ((x) >> (64 - (b))))
#define SIPROUND                                                                                  \
    do {                                                                                          \
        v0 += v1; v1 = ROTL64(v1, 13); v1 ^= v0; v0 = ROTL64(v0, 32);                             \
        v2 += v3; v3 = ROTL64(v3, 16); v3 ^= v2;                                                  \
        v0 += v3; v3 = ROTL64(v3, 21); v3 ^= v0;                                                  \
        v2 += v1; v1 = ROTL64(v1, 17); v1 ^= v2; v2 = ROTL64(v2, 32);                             \
    } while (0)

static inline uint64_t siphash13_u64(uint64_t k0, uint64_t k1, uint64_t m)
{
    uint64_t v0 = k0 ^ 0x736f6d6570736575ULL, v1 = k1 ^ 0x646f72616e646f6dULL;
    uint64_t v2 = k0 ^ 0x6c7967656e657261ULL, v3 = k1 ^ 0x7465646279746573ULL;
    v3 ^= m; SIPROUND; v0 ^= m;
    uint64_t b = (uint64_t)8 << 56;
    v3 ^= b; SIPROUND; v0 ^= b;
    v2 ^= 0xff; SIPROUND; SIPROUND; SIPROUND;
    return v0 ^ v1 ^ v2 ^ v3;
}

typedef struct {
    uint64_t *keys;
    uint32_t *vals;
    uint64_t mask;
} idmap;

static int idmap_init(idmap *m, uint32_t n)
{
    uint64_t cap = 16;
    while (cap < (uint64_t)n * 2) cap <<= 1;
    m->keys = (uint64_t *)malloc(cap * sizeof(uint64_t));
    m->vals = (uint32_t *)malloc(cap * sizeof(uint32_t));
    if (!m->keys || !m->vals) return -1;
    memset(m->keys, 0xff, cap * sizeof(uint64_t));
    m->mask = cap - 1;
    for (uint32_t i = 0; i < n; ++i) {
        uint64_t h = siphash13_u64(0x0706050403020100ULL, 0x0f0e0d0c0b0a0908ULL, i) & m->mask;
        while (m->keys[h] != UINT64_MAX) h = (h + 1) & m->mask;
        m->keys[h] = i;
        m->vals[h] = i;
    }
    return 0;
}

static inline uint32_t idmap_get(const idmap *m, uint64_t id)
{
    uint64_t h = siphash13_u64(0x0706050403020100ULL, 0x0f0e0d0c0b0a0908ULL, id) & m->mask;
    while (m->keys[h] != id) h = (h + 1) & m->mask;
    return m->vals[h];
}

static void idmap_free(idmap *m)
{
    free(m->keys);
    free(m->vals);
}

static inline float D_faithful(const float *packed, const idmap *m, uint32_t id1, uint32_t id2)
{
    if (id1 == id2) return 0.0f;
    return tlo_dm_lookup(packed, idmap_get(m, id1), idmap_get(m, id2));
}

/* ------------------------------------------------------------------------------------------ */
/* 2-opt                                                                                      */
/* ------------------------------------------------------------------------------------------ */

/* two_opt.rs:69-79  swap_2opt: reverse path[from..=to]; no-op when from >= to */
void tlo_swap_2opt(uint32_t *path, uint32_t from, uint32_t to)
{
    if (from >= to) return;
    while (from < to) {
        uint32_t t = path[from];
        path[from] = path[to];
        path[to] = t;
        ++from;
        --to;
    }
}

/* two_opt.rs:7-67 */
int tlo_two_opt(const float *xy, const float *packed, uint32_t n, const uint32_t *init,
                uint32_t *out, float *out_cost, tlo_stats *st, int flavor, uint64_t max_candidates)
{
    if ((!xy && !packed) || !out) return TLO_ERR_BADARG;
    /* :17,29  n_indices = len-1; 0..(n_indices-2) underflows for n <= 2 -> the reference panics */
    if (n < 3) return TLO_ERR_REF_PANICS;
    dsrc s = {xy, packed};
    for (uint32_t i = 0; i < n; ++i) out[i] = init ? init[i] : i; /* :18-20 */

    float *own_packed = NULL;
    idmap map = {0};
    if (flavor == 1) {
        if (!packed) {
            own_packed = (float *)malloc((size_t)n * (n - 1) / 2 * sizeof(float));
            if (!own_packed) return TLO_ERR_NOMEM;
            tlo_dm_build_packed(xy, n, own_packed);
            packed = own_packed;
        }
        if (idmap_init(&map, n)) { free(own_packed); return TLO_ERR_NOMEM; }
    }

    uint64_t sweeps = 0, cands = 0, moves = 0, reversed = 0;
    const uint32_t n_indices = n - 1;
    int improved = 1;
    while (improved) { /* :26-27 */
        improved = 0;
        ++sweeps;
        for (uint32_t i = 0; i + 2 < n_indices; ++i) {       /* :29 */
            for (uint32_t j = i + 2; j < n_indices; ++j) {   /* :34 */
                float cur, neu;
                if (flavor == 1) {
                    cur = D_faithful(packed, &map, out[i], out[i + 1]) +
                          D_faithful(packed, &map, out[j], out[j + 1]);
                    neu = D_faithful(packed, &map, out[i], out[j]) +
                          D_faithful(packed, &map, out[i + 1], out[j + 1]);
                } else {
                    cur = D(&s, out[i], out[i + 1]) + D(&s, out[j], out[j + 1]); /* :35-40 */
                    neu = D(&s, out[i], out[j]) + D(&s, out[i + 1], out[j + 1]); /* :42-47 */
                }
                ++cands;
                if (neu < cur) { /* :49 strict, no epsilon */
                    tlo_swap_2opt(out, i + 1, j); /* :50 */
                    improved = 1;
                    ++moves;
                    reversed += (uint64_t)(j - i);
                }
            }
        }
        if (max_candidates && cands >= max_candidates) break;
    }
    if (flavor == 1) { idmap_free(&map); free(own_packed); packed = s.packed; }
    if (st) { st->sweeps = sweeps; st->candidates = cands; st->moves = moves; st->reversed = reversed; }
    if (out_cost) *out_cost = tlo_tour_length(xy, s.packed, n, out); /* mod.rs:1776-1789 */
    return TLO_OK;
}

/* two_opt.rs:26-61 once more, recording what the progress channel of the reference carries (:30-32, :53-56): for every applied
 * move its (i, j), the new_distance the PathUpdate message is sent with and the sweep it happened in.  log_ij holds 2 words per
 * move; *len counts every move, the logs hold the first `cap`. */
int tlo_two_opt_trace(const float *xy, const float *packed, uint32_t n, const uint32_t *init, uint32_t *out, float *out_cost,
                      tlo_stats *st, uint32_t *log_ij, float *log_dist, uint32_t *log_sweep, uint64_t cap, uint64_t *len)
{
    if ((!xy && !packed) || !out || !len) return TLO_ERR_BADARG;
    if (n < 3) return TLO_ERR_REF_PANICS;
    dsrc s = {xy, packed};
    for (uint32_t i = 0; i < n; ++i) out[i] = init ? init[i] : i; /* :18-20 */
    uint64_t sweeps = 0, cands = 0, moves = 0, reversed = 0;
    const uint32_t n_indices = n - 1;
    int improved = 1;
    while (improved) {
        improved = 0;
        ++sweeps;
        for (uint32_t i = 0; i + 2 < n_indices; ++i) {
            for (uint32_t j = i + 2; j < n_indices; ++j) {
                const float cur = D(&s, out[i], out[i + 1]) + D(&s, out[j], out[j + 1]);
                const float neu = D(&s, out[i], out[j]) + D(&s, out[i + 1], out[j + 1]);
                ++cands;
                if (neu < cur) {
                    tlo_swap_2opt(out, i + 1, j);
                    improved = 1;
                    if (moves < cap) {
                        if (log_ij) { log_ij[2 * moves] = i; log_ij[2 * moves + 1] = j; }
                        if (log_dist) log_dist[moves] = neu; /* :55 PathUpdate(Route::new(&path), new_distance) */
                        if (log_sweep) log_sweep[moves] = (uint32_t)sweeps; /* 1-based pass of the `while improved` loop (:26) */
                    }
                    ++moves;
                    reversed += (uint64_t)(j - i);
                }
            }
        }
    }
    *len = moves;
    if (st) { st->sweeps = sweeps; st->candidates = cands; st->moves = moves; st->reversed = reversed; }
    if (out_cost) *out_cost = tlo_tour_length(xy, s.packed, n, out);
    return TLO_OK;
}

/* BEST-SWEEP 2-opt — this build's own throughput mode (not in the reference). */
int tlo_two_opt_best(const float *xy, const float *packed, uint32_t n, const uint32_t *init,
                     uint32_t *out, float *out_cost, tlo_stats *st, uint64_t max_moves)
{
    if ((!xy && !packed) || !out) return TLO_ERR_BADARG;
    if (n < 3) return TLO_ERR_REF_PANICS;
    dsrc s = {xy, packed};
    for (uint32_t i = 0; i < n; ++i) out[i] = init ? init[i] : i;
    uint64_t sweeps = 0, cands = 0, moves = 0, reversed = 0;
    const uint32_t n_indices = n - 1;
    for (;;) {
        ++sweeps;
        float best = 0.0f;
        uint32_t bi = 0, bj = 0;
        int have = 0;
        for (uint32_t i = 0; i + 2 < n_indices; ++i) {
            for (uint32_t j = i + 2; j < n_indices; ++j) {
                float cur = D(&s, out[i], out[i + 1]) + D(&s, out[j], out[j + 1]);
                float neu = D(&s, out[i], out[j]) + D(&s, out[i + 1], out[j + 1]);
                ++cands;
                if (neu < cur) {
                    float delta = neu - cur;
                    if (!have || delta < best) { best = delta; bi = i; bj = j; have = 1; }
                }
            }
        }
        if (!have) break;
        tlo_swap_2opt(out, bi + 1, bj);
        ++moves;
        reversed += (uint64_t)(bj - bi);
        if (max_moves && moves >= max_moves) break;
    }
    if (st) { st->sweeps = sweeps; st->candidates = cands; st->moves = moves; st->reversed = reversed; }
    if (out_cost) *out_cost = tlo_tour_length(xy, packed, n, out);
    return TLO_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* 3-opt                                                                                      */
/* ------------------------------------------------------------------------------------------ */

/* three_opt.rs:170-180; e[] in TripleEdges field order (:135-153) */
void tlo_reconnection_costs(const float e[12], float out[7])
{
    const float d_ab = e[0], d_c_dt = e[1], d_ac = e[2], d_b_dt = e[3], d_a_dt = e[4], d_ef = e[5],
                d_ce = e[6], d_dt_f = e[7], d_be = e[8], d_cf = e[9], d_bf = e[10], d_ae = e[11];
    out[0] = d_ac + d_b_dt + d_ef;
    out[1] = d_ab + d_ce + d_dt_f;
    out[2] = d_ac + d_be + d_dt_f;
    out[3] = d_a_dt + d_be + d_cf;
    out[4] = d_a_dt + d_ce + d_bf;
    out[5] = d_ae + d_b_dt + d_cf;
    out[6] = d_ae + d_c_dt + d_bf;
}

static void reverse_u32(uint32_t *p, uint32_t lo, uint32_t hi) /* inclusive, lo<=hi+1 */
{
    while (lo < hi) {
        uint32_t t = p[lo];
        p[lo] = p[hi];
        p[hi] = t;
        ++lo;
        --hi;
    }
}

/* three_opt.rs:186-218 */
int tlo_apply_3opt(uint32_t *path, uint32_t n, uint32_t i, uint32_t j, uint32_t k, int kase)
{
    if (!(i < j && j < k && k < n)) return TLO_ERR_BADARG;
    switch (kase) {
    case 1: reverse_u32(path, i + 1, j); return TLO_OK;
    case 2: reverse_u32(path, j + 1, k); return TLO_OK;
    case 3: reverse_u32(path, i + 1, j); reverse_u32(path, j + 1, k); return TLO_OK;
    case 4: case 5: case 6: case 7: {
        uint32_t l1 = j - i, l2 = k - j;
        uint32_t *tmp = (uint32_t *)malloc((size_t)(l1 + l2) * sizeof(uint32_t));
        if (!tmp) return TLO_ERR_NOMEM;
        /* new_mid = seg2' ++ seg1' ; seg1 = path[i+1..=j], seg2 = path[j+1..=k] */
        int rev1 = (kase == 5 || kase == 7), rev2 = (kase == 6 || kase == 7);
        for (uint32_t t = 0; t < l2; ++t) tmp[t] = rev2 ? path[k - t] : path[j + 1 + t];
        for (uint32_t t = 0; t < l1; ++t) tmp[l2 + t] = rev1 ? path[j - t] : path[i + 1 + t];
        memcpy(path + i + 1, tmp, (size_t)(l1 + l2) * sizeof(uint32_t));
        free(tmp);
        return TLO_OK;
    }
    default: return TLO_ERR_REF_PANICS; /* :216 unreachable!() */
    }
}

/* three_opt.rs:58-131 */
static int find_best_move(const dsrc *s, uint32_t n, const uint32_t *path, uint32_t *oi,
                          uint32_t *oj, uint32_t *ok, int *okase, float *osav, uint64_t *evals)
{
    int have = 0;
    float best_savings = 0.0f; /* :61 */
    uint64_t ev = 0;
    for (uint32_t i = 0; i + 2 < n; ++i) { /* :63 0..n-2 */
        uint32_t a = path[i], b = path[i + 1];
        float d_ab = D(s, a, b);
        for (uint32_t j = i + 1; j + 1 < n; ++j) { /* :69 i+1..n-1 */
            uint32_t c = path[j], dt = path[j + 1];
            float d_c_dt = D(s, c, dt), d_ac = D(s, a, c), d_b_dt = D(s, b, dt), d_a_dt = D(s, a, dt);
            for (uint32_t k = j + 1; k < n; ++k) { /* :78 */
                if (i == 0 && k == n - 1) continue; /* :81-83 */
                uint32_t e = path[k], f = path[(k + 1) % n];
                float te[12] = {d_ab, d_c_dt, d_ac, d_b_dt, d_a_dt,
                                D(s, e, f), D(s, c, e), D(s, dt, f), D(s, b, e),
                                D(s, c, f), D(s, b, f), D(s, a, e)};
                float orig = d_ab + d_c_dt + te[5]; /* :96 */
                float costs[7];
                tlo_reconnection_costs(te, costs);
                ++ev;
                /* :113-117 filter(cost < orig).min_by(partial_cmp): FIRST minimum on ties */
                int ci = -1;
                float cmin = 0.0f;
                for (int q = 0; q < 7; ++q) {
                    if (costs[q] < orig && (ci < 0 || costs[q] < cmin)) { ci = q; cmin = costs[q]; }
                }
                if (ci >= 0) {
                    float savings = orig - cmin; /* :120 */
                    if (savings > best_savings) { /* :121 strict */
                        best_savings = savings;
                        *oi = i; *oj = j; *ok = k; *okase = ci + 1;
                        have = 1;
                    }
                }
            }
        }
    }
    if (osav) *osav = best_savings;
    if (evals) *evals += ev;
    return have;
}

int tlo_three_opt_find_best_move(const float *xy, const float *packed, uint32_t n,
                                 const uint32_t *path, uint32_t *oi, uint32_t *oj, uint32_t *ok,
                                 int *okase, float *osav)
{
    dsrc s = {xy, packed};
    if (n < 4) return 0;
    return find_best_move(&s, n, path, oi, oj, ok, okase, osav, NULL);
}

/* three_opt.rs:16-51 */
int tlo_three_opt(const float *xy, const float *packed, uint32_t n, const uint32_t *init,
                  uint32_t *out, float *out_cost, tlo_stats *st, uint64_t max_moves)
{
    if ((!xy && !packed) || !out) return TLO_ERR_BADARG;
    dsrc s = {xy, packed};
    uint64_t passes = 0, evals = 0, moves = 0;
    if (n < 4) { /* :25-28 returns the cities order, ignoring init_tour */
        for (uint32_t i = 0; i < n; ++i) out[i] = i;
    } else {
        for (uint32_t i = 0; i < n; ++i) out[i] = init ? init[i] : i;
        for (;;) {
            uint32_t i, j, k;
            int kase;
            ++passes;
            if (!find_best_move(&s, n, out, &i, &j, &k, &kase, NULL, &evals)) break;
            tlo_apply_3opt(out, n, i, j, k, kase);
            ++moves;
            if (max_moves && moves >= max_moves) break;
        }
    }
    if (st) { st->sweeps = passes; st->candidates = evals; st->moves = moves; st->reversed = 0; }
    if (out_cost) *out_cost = tlo_tour_length(xy, packed, n, out);
    return TLO_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* nearest-neighbour seed                                                                     */
/* ------------------------------------------------------------------------------------------ */

/* nearest_neighbor.rs:8-76; k-buffer = distance_matrix.rs:259-280 + mod.rs:1839-1860 */
int tlo_nearest_neighbor(const float *xy, const float *packed, uint32_t n, uint32_t n_nearest,
                         uint32_t *out, float *out_cost)
{
    if ((!xy && !packed) || !out || n == 0) return TLO_ERR_BADARG;
    dsrc s = {xy, packed};
    unsigned char *visited = (unsigned char *)calloc(n, 1);
    uint32_t *bp = (uint32_t *)malloc(((size_t)n_nearest + 1) * sizeof(uint32_t));
    float *bd = (float *)malloc(((size_t)n_nearest + 1) * sizeof(float));
    if (!visited || !bp || !bd) { free(visited); free(bp); free(bd); return TLO_ERR_NOMEM; }
    uint32_t len = 1;
    out[0] = 0; /* :28 start = cities[0] */
    visited[0] = 1;
    while (len < n) {
        uint32_t cur = out[len - 1];
        /* frontier = distances.nearest(current, n_nearest): positions in ascending order,
         * insert iff d < search_radius (INF until the buffer is full, then the farthest kept),
         * at partition_point(r.distance <= d) i.e. AFTER equal distances, truncate to n_nearest */
        uint32_t cnt = 0;
        for (uint32_t pos = 0; pos < n && n_nearest > 0; ++pos) {
            if (pos == cur) continue;
            float d = D(&s, cur, pos);
            float radius = (cnt < n_nearest) ? INFINITY : bd[cnt - 1];
            if (d < radius) {
                uint32_t ins = 0;
                while (ins < cnt && bd[ins] <= d) ++ins;
                for (uint32_t t = cnt; t > ins; --t) { bd[t] = bd[t - 1]; bp[t] = bp[t - 1]; }
                bd[ins] = d;
                bp[ins] = pos;
                if (cnt < n_nearest) ++cnt; /* truncate(n) */
            }
        }
        uint32_t next = UINT32_MAX;
        for (uint32_t t = 0; t < cnt; ++t)
            if (!visited[bp[t]]) { next = bp[t]; break; } /* :46-49 */
        if (next == UINT32_MAX) {
            /* :50-63 global nearest unvisited.  Reference iterates a HashSet (random order) and
             * min_by keeps the first minimum; ORACLE RULE: lowest position wins ties. */
            float bestd = 0.0f;
            for (uint32_t pos = 0; pos < n; ++pos) {
                if (visited[pos]) continue;
                float d = D(&s, cur, pos);
                if (next == UINT32_MAX || d < bestd) { next = pos; bestd = d; }
            }
        }
        out[len++] = next;
        visited[next] = 1;
    }
    free(visited); free(bp); free(bd);
    if (out_cost) *out_cost = tlo_tour_length(xy, packed, n, out);
    return TLO_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* synthetic inputs (own specification)                                */
/* ------------------------------------------------------------------------------------------ */

void tlo_synth_xy(uint32_t n, uint64_t seed, float *xy)
{
    uint64_t s = seed ? seed : 88172645463325252ULL;
    for (uint32_t i = 0; i < 2 * n; ++i) {
        s ^= s << 13;
        s ^= s >> 7;
        s ^= s << 17;
        xy[i] = (float)(s % 1000000ULL) / 1000.0f;
    }
}

static inline uint64_t splitmix64_next(uint64_t *state)
{
    uint64_t z = (*state += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

void tlo_restart_perm(uint32_t n, uint64_t seed, uint64_t r, uint32_t *perm)
{
    uint64_t st = seed + r;
    for (uint32_t i = 0; i < n; ++i) perm[i] = i;
    for (uint32_t i = n; i-- > 1;) {
        uint32_t j = (uint32_t)(splitmix64_next(&st) % ((uint64_t)i + 1));
        uint32_t t = perm[i];
        perm[i] = perm[j];
        perm[j] = t;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* Or-opt (reference: src/tsp/or_opt.rs) — SURVEY.md §8(f) "next" row 3                        */
/* ------------------------------------------------------------------------------------------ */

/* or_opt.rs:170-184 apply_relocation: drain path[i..i+seg_len], reinsert after original index j */
int tlo_apply_relocation(uint32_t *tour, uint32_t n, uint32_t i, uint32_t seg_len, uint32_t j, int reversed)
{
    if (seg_len == 0 || seg_len > 3 || i + seg_len > n || j >= n) return TLO_ERR_BADARG;
    uint32_t seg[3];
    for (uint32_t t = 0; t < seg_len; ++t) seg[t] = tour[i + t];
    /* drain */
    for (uint32_t t = i; t + seg_len < n; ++t) tour[t] = tour[t + seg_len];
    uint32_t m = n - seg_len;
    uint32_t insert_at = (j >= i + seg_len) ? (j - seg_len + 1) : (j + 1);
    if (insert_at > m) return TLO_ERR_BADARG;
    for (uint32_t t = m; t > insert_at; --t) tour[t - 1 + seg_len] = tour[t - 1];
    for (uint32_t t = 0; t < seg_len; ++t) tour[insert_at + t] = reversed ? seg[seg_len - 1 - t] : seg[t];
    return TLO_OK;
}

/* or_opt.rs:80-164 find_best_move: returns 1 and (delta, i, j, seg_len, reversed) or 0 */
static int or_find_best_move(const dsrc *s, uint32_t n, const uint32_t *path, float *odelta, uint32_t *oi, uint32_t *oj,
                             uint32_t *oseg, int *orev, uint64_t *evals)
{
    float best_delta = -1e-3f; /* :86 */
    int have = 0;
    uint64_t ev = 0;
    for (uint32_t seg_len = 1; seg_len <= 3; ++seg_len) {
        if (n <= seg_len + 1) continue; /* :90-92 */
        for (uint32_t i = 0; i < n; ++i) {
            if (i + seg_len > n) continue; /* :98-100 */
            uint32_t prev = i == 0 ? n - 1 : i - 1;
            uint32_t after_seg = (i + seg_len) % n;
            uint32_t a = path[prev], first_seg = path[i], last_seg = path[i + seg_len - 1], d = path[after_seg];
            float remove_gain = D(s, a, first_seg) + D(s, last_seg, d) - D(s, a, d); /* :114-116 */
            for (uint32_t j = 0; j < n; ++j) {
                if (j == prev || (j >= i && j < i + seg_len)) continue; /* :123-125 */
                uint32_t x = path[j], y = path[(j + 1) % n];
                float edge_xy = D(s, x, y);
                float fwd_delta = -remove_gain + D(s, x, first_seg) + D(s, last_seg, y) - edge_xy; /* :136-139 */
                ++ev;
                if (fwd_delta < best_delta) { best_delta = fwd_delta; *oi = i; *oj = j; *oseg = seg_len; *orev = 0; have = 1; }
                if (seg_len > 1) {
                    float rev_delta = -remove_gain + D(s, x, last_seg) + D(s, first_seg, y) - edge_xy; /* :148-151 */
                    ++ev;
                    if (rev_delta < best_delta) { best_delta = rev_delta; *oi = i; *oj = j; *oseg = seg_len; *orev = 1; have = 1; }
                }
            }
        }
    }
    if (odelta) *odelta = best_delta;
    if (evals) *evals += ev;
    return have;
}

int tlo_or_opt_find_best_move(const float *xy, const float *packed, uint32_t n, const uint32_t *path, float *delta,
                              uint32_t *i, uint32_t *j, uint32_t *seg_len, int *reversed)
{
    dsrc s = {xy, packed};
    if (n < 4) return 0;
    return or_find_best_move(&s, n, path, delta, i, j, seg_len, reversed, NULL);
}

/* or_opt.rs:18-74 solve */
int tlo_or_opt(const float *xy, const float *packed, uint32_t n, const uint32_t *init, uint32_t *out, float *out_cost,
               tlo_stats *st, uint64_t max_moves)
{
    if ((!xy && !packed) || !out) return TLO_ERR_BADARG;
    dsrc s = {xy, packed};
    uint64_t passes = 0, evals = 0, moves = 0;
    if (n < 4) { /* :31-34 */
        for (uint32_t t = 0; t < n; ++t) out[t] = t;
    } else {
        for (uint32_t t = 0; t < n; ++t) out[t] = init ? init[t] : t;
        for (;;) {
            uint32_t i = 0, j = 0, seg = 0;
            int rev = 0;
            ++passes;
            if (!or_find_best_move(&s, n, out, NULL, &i, &j, &seg, &rev, &evals)) break;
            tlo_apply_relocation(out, n, i, seg, j, rev);
            ++moves;
            if (max_moves && moves >= max_moves) break;
        }
    }
    if (st) { st->sweeps = passes; st->candidates = evals; st->moves = moves; st->reversed = 0; }
    if (out_cost) *out_cost = tlo_tour_length(xy, packed, n, out);
    return TLO_OK;
}
