/*
 * tl_oracle.h — CPU ORACLE for the teeline 2-opt / 3-opt / LK hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product (teeline_amd/, include/) never
 * links, imports or falls back to anything in oracle/.
 *
 * It is a plain-C restatement of the reference's algorithms (timgluz/teeline, Rust crate
 * `teeline`), written from the cited files; no reference source is copied.  The reference
 * cannot be compiled here (no cargo/rustc), so the oracle is pinned by
 *   (1) every exact-value unit test the reference holds for this path (tsp5, swap_2opt,
 *       tiny matrices, 3-opt case table / permutations), and
 *   (2) the reference's committed output numbers (docs/benchmarks.md, README.md,
 *       bench/baseline-solvers.tsv) reproduced bit-for-bit — see tests/test_oracle_golden.py.
 *
 * Numerics contract (reference: src/tsp/kdtree.rs:291-295): distances are
 * sqrtf(dx*dx + dy*dy) in f32, no FMA contraction, correctly rounded sqrt.  Build with
 * -ffp-contract=off (see oracle/Makefile).
 *
 * All tours are in POSITIONS (index into the cities array, 0..n-1); the id<->position
 * mapping the reference performs with HashMaps (distance_matrix.rs:197-212) is an identity
 * on values and happens at the boundary.
 */
#ifndef TL_ORACLE_H
#define TL_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TLO_OK 0
#define TLO_ERR_BADARG (-1)
#define TLO_ERR_REF_PANICS (-2) /* input on which the reference itself panics */
#define TLO_ERR_NOMEM (-3)

typedef struct tlo_stats {
    uint64_t sweeps;      /* outer `while improved` iterations (2-opt) / passes (3-opt)      */
    uint64_t candidates;  /* (i,j) pairs / (i,j,k) triples whose delta test was evaluated     */
    uint64_t moves;       /* improving moves applied                                          */
    uint64_t reversed;    /* tour elements moved by segment reversals                         */
} tlo_stats;

/* kdtree.rs:291-295 */
float tlo_dist(float x1, float y1, float x2, float y2);

/* distance_matrix.rs:122-153 — packed strict lower triangle, row-major; out has n(n-1)/2 floats. */
int tlo_dm_build_packed(const float *xy, uint32_t n, float *out);
/* distance_matrix.rs:59-75 GEO variant of the same layout. */
int tlo_dm_build_packed_geo(const float *xy, uint32_t n, float *out);
/* distance_matrix.rs:177-191 */
float tlo_dm_lookup(const float *packed, uint32_t p, uint32_t q);
/* full row-major n x n expansion of the packed triangle (diagonal 0). */
int tlo_dm_expand_full(const float *packed, uint32_t n, float *out_full);

/* distance_matrix.rs:235-245 — total = d(last,first) then += d(w0,w1) in order.
 * Exactly one of xy / packed may be NULL. */
float tlo_tour_length(const float *xy, const float *packed, uint32_t n, const uint32_t *perm);

/* two_opt.rs:7-67 (first-improvement, open path). init may be NULL (identity order).
 * flavor: 0 = on-the-fly / direct index ("best effort"), 1 = "ref-faithful": packed matrix +
 * two SipHash-1-3 id->position lookups per distance (distance_matrix.rs:197-212) — same results,
 * only the cost model differs (used by bench.py's cpu_baseline). */
int tlo_two_opt(const float *xy, const float *packed, uint32_t n, const uint32_t *init,
                uint32_t *out_perm, float *out_cost, tlo_stats *stats, int flavor,
                uint64_t max_candidates /* 0 = run to the local optimum; else stop after the
                                           sweep... never mid-sweep: bounded timing samples stop
                                           at the first sweep boundary past this count */);

/* two_opt.rs:69-79 */
void tlo_swap_2opt(uint32_t *path, uint32_t from, uint32_t to);

/* BEST-SWEEP 2-opt (this build's own specification, NOT the reference's algorithm): per sweep
 * evaluate every (i,j) of the same open-path candidate set, pick the minimum f32 delta
 * (new - cur) < 0 with the lowest linear index on ties, apply, repeat. */
/* two_opt::solve + the moves it applied: (i, j) and the PathUpdate distance of each (two_opt.rs:53-56) */
int tlo_two_opt_trace(const float *xy, const float *packed, uint32_t n, const uint32_t *init, uint32_t *out, float *out_cost,
                      tlo_stats *st, uint32_t *log_ij, float *log_dist, uint32_t *log_sweep, uint64_t cap, uint64_t *len);
int tlo_two_opt_best(const float *xy, const float *packed, uint32_t n, const uint32_t *init,
                     uint32_t *out_perm, float *out_cost, tlo_stats *stats, uint64_t max_moves);

/* three_opt.rs:170-180: 12 distances in TripleEdges order
 * (d_ab,d_c_dt,d_ac,d_b_dt,d_a_dt,d_ef,d_ce,d_dt_f,d_be,d_cf,d_bf,d_ae) -> 7 costs. */
void tlo_reconnection_costs(const float e[12], float out[7]);
/* three_opt.rs:186-218 */
int tlo_apply_3opt(uint32_t *path, uint32_t n, uint32_t i, uint32_t j, uint32_t k, int kase);
/* three_opt.rs:58-131; returns 1 if a move was found. */
int tlo_three_opt_find_best_move(const float *xy, const float *packed, uint32_t n,
                                 const uint32_t *path, uint32_t *oi, uint32_t *oj, uint32_t *ok,
                                 int *okase, float *osavings);
/* three_opt.rs:16-51 */
int tlo_three_opt(const float *xy, const float *packed, uint32_t n, const uint32_t *init,
                  uint32_t *out_perm, float *out_cost, tlo_stats *stats, uint64_t max_moves);

/* nearest_neighbor.rs:8-76 + distance_matrix.rs:259-280 + mod.rs:1839-1889.
 * Fallback tie rule (reference: HashSet iteration order, non-deterministic): lowest position. */
int tlo_nearest_neighbor(const float *xy, const float *packed, uint32_t n, uint32_t n_nearest,
                         uint32_t *out_perm, float *out_cost);

/* mod.rs:1620-1634 validate_tour: 1 if perm is a permutation of 0..n-1. */
int tlo_validate_tour(const uint32_t *perm, uint32_t n);

/* ---- synthetic inputs (this build's own specification) ---- */
/* xorshift64 (s^=s<<13; s^=s>>7; s^=s<<17), seed 88172645463325252 by default; x then y per
 * city; coord = (u % 1000000) / 1000.0f. */
void tlo_synth_xy(uint32_t n, uint64_t seed, float *xy);
/* Fisher–Yates `for i in (1..n).rev(): j = rng(0..=i); swap` from splitmix64(seed + r). */
void tlo_restart_perm(uint32_t n, uint64_t seed, uint64_t r, uint32_t *perm);

/* ---- Or-opt (or_opt.rs) ---- */
/* or_opt.rs:170-184 */
int tlo_apply_relocation(uint32_t *tour, uint32_t n, uint32_t i, uint32_t seg_len, uint32_t j, int reversed);
/* or_opt.rs:80-164; returns 1 if an improving relocation (delta < -1e-3) exists */
int tlo_or_opt_find_best_move(const float *xy, const float *packed, uint32_t n, const uint32_t *path, float *delta,
                              uint32_t *i, uint32_t *j, uint32_t *seg_len, int *reversed);
/* or_opt.rs:18-74 */
int tlo_or_opt(const float *xy, const float *packed, uint32_t n, const uint32_t *init, uint32_t *out_perm,
               float *out_cost, tlo_stats *stats, uint64_t max_moves);

/* ---- Lin–Kernighan (lin_kernighan.rs) ---- */
/* lin_kernighan.rs:12-27 via brute force; k' = min(k, n-1) ids per city, ascending distance,
 * ties -> lowest position (reference: kd-tree traversal order, implementation-defined). */
int tlo_build_candidates(const float *xy, uint32_t n, uint32_t k, uint32_t *out /* n*k' */);
/* lin_kernighan.rs:454-481 lk_pass on a flat tour; returns 1 if improved. */
int tlo_lk_pass(const float *xy, uint32_t n, uint32_t *tour, const uint32_t *cand, uint32_t k,
                uint32_t max_depth, tlo_stats *stats);
/* lin_kernighan.rs:485-499 with explicit draws r1,r2,r3 in [0, n/4). */
void tlo_double_bridge(const uint32_t *tour, uint32_t n, uint32_t r1, uint32_t r2, uint32_t r3,
                       uint32_t *out);
/* lin_kernighan.rs:35-100 with seeded kicks: draws from splitmix64(seed) stream,
 * r = next() % (n/4) (reference: unseeded thread RNG — not reproducible by anyone). */
int tlo_lin_kernighan(const float *xy, uint32_t n, const uint32_t *init, uint32_t epochs,
                      uint32_t platoo_epochs, uint32_t n_nearest, uint32_t max_depth,
                      uint64_t seed, uint32_t *out_perm, float *out_cost, tlo_stats *stats);

/* the same with precomputed candidate lists (n x min(n_nearest, n-1) positions), e.g. tlo_build_candidates_kdtree's, and
 * with problem.distances of a GEO / EXPLICIT problem (packed, may be NULL): it feeds the NN seed (:47-55) and the reported
 * total (:99) only — the search is Euclidean over xy (:41) */
int tlo_lin_kernighan_cand(const float *xy, const float *packed, uint32_t n, const uint32_t *init, uint32_t epochs,
                           uint32_t platoo_epochs, uint32_t n_nearest, uint32_t max_depth, uint64_t seed,
                           const uint32_t *cand, uint32_t *out_perm, float *out_cost, tlo_stats *stats);
/* the same solve (kd-tree candidate lists) with the progress messages of lin_kernighan.rs:71,90 recorded: every best tour the
 * search settles on (positions) and its best_dist, in order; at most cap tours are stored, *count counts them all */
int tlo_lin_kernighan_trace(const float *xy, const float *packed, uint32_t n, const uint32_t *init, uint32_t epochs,
                            uint32_t platoo_epochs, uint32_t n_nearest, uint32_t max_depth, uint64_t seed,
                            uint32_t *out, float *out_cost, tlo_stats *st, uint32_t *snaps, float *snap_dist, uint32_t cap, uint32_t *count);
/* lin_kernighan.rs:134-145 */
void tlo_flat_to_next_prev(const uint32_t *tour, uint32_t n, uint32_t *next, uint32_t *prev);
/* lin_kernighan.rs:345-389 on a flat tour (city_ids = the tour, as the reference's unit tests call it); out_chain needs
 * 2 * max_depth + 4 entries; returns the chain length (0 = None). */
int tlo_find_lk_move(const float *xy, uint32_t n, const uint32_t *tour, const uint32_t *cand, uint32_t k,
                     uint32_t max_depth, uint32_t *out_chain);
/* lin_kernighan.rs:397-450 */
int tlo_apply_lk_chain(uint32_t *tour, uint32_t n, const uint32_t *chain, uint32_t clen);

/* ---- kd-tree (kdtree.rs), tl_oracle_kdtree.c ---- */
/* lin_kernighan.rs:12-27 through the reference's kd-tree (kdtree.rs:19-73 build, :193-212 nearest, mod.rs:1839-1889
 * k-buffer).  *tie_free = 1 iff no median selection met elements comparing Equal (kdtree.rs:301-317) around the pivot,
 * i.e. iff the tree is independent of Rust's select_nth_unstable_by implementation (kdtree.rs:63). */
int tlo_build_candidates_kdtree(const float *xy, uint32_t n, uint32_t k, uint32_t *out, int *tie_free);
/* KDTree::nearest (kdtree.rs:116-122): ids NULL = position; returns the result count (buffer order in out_*). */
int tlo_kdtree_nearest(const float *xy, const uint64_t *ids, uint32_t n, float qx, float qy, uint64_t qid, uint32_t k,
                       uint32_t *out_pos, float *out_dist);
/* KDTree::walk (kdtree.rs:98-108): positions in in-order. */
int tlo_kdtree_walk(const float *xy, uint32_t n, uint32_t *out_order, int *tie_free);

#ifdef __cplusplus
}
#endif
#endif
