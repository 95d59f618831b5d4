// teeline_gpu.hpp — C++ host-side mirror of the reference's solver interface for the 2-opt / 3-opt / LK path,
// over the C ABI of libteeline_gpu.so (include/teeline_gpu.h).  Header-only; no torch, no Python.
//
// The reference is Rust and no Rust toolchain exists in the build image, so this is the compiled-language mirror of
//   src/tsp/kdtree.rs:248-252        KDPoint
//   src/tsp/distance_matrix.rs       DistanceMatrix (packed strict lower triangle)
//   src/tsp/mod.rs:1731-1814         TspProblem, Solution
//   src/tsp/mod.rs:596-613,1249-1267 HeuristicOptions, LKOptions
//   src/tsp/tsplib.rs:101-255        tsplib::read_from_file
//   src/tsp/{two_opt,three_opt,or_opt,lin_kernighan,nearest_neighbor}.rs  solve(problem, opts, progress, init_tour)
//   src/tsp/pipeline.rs:53-80        run_pipeline_stages (warm start + validate_tour)
// Same names, argument meaning and error behaviour: dispatcher-level failures throw std::runtime_error (the
// reference returns Err(String), mod.rs:1661), inputs on which the reference panics throw teeline::ReferencePanic.
#pragma once
#include <teeline_gpu.h>

#include <algorithm>
#include <cctype>
#include <charconv>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <exception>
#include <fstream>
#include <functional>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

namespace teeline {

struct ReferencePanic : std::runtime_error {
    using std::runtime_error::runtime_error;
};

class Context {  // one tl_ctx per thread (teeline-api calls solvers from spawn_blocking, tsp_service.rs:295)
  public:
    explicit Context(int device = 0, uint32_t flags = TL_FLAG_NONE)
    {
        if (tl_create(device, flags, &h_) != TL_OK) throw std::runtime_error(tl_last_error(nullptr));
    }
    ~Context() { tl_destroy(h_); }
    Context(const Context &) = delete;
    Context &operator=(const Context &) = delete;
    tl_ctx *get() const { return h_; }
    void check(int rc) const
    {
        if (rc == TL_OK) return;
        if (rc == TL_ERR_BUSY) throw std::runtime_error("teeline-gpu: the context is in use by another thread (one Context per thread)");
        if (rc == TL_ERR_REF_PANICS) throw ReferencePanic(tl_last_error(h_));
        throw std::runtime_error(std::string("teeline-gpu: ") + tl_last_error(h_));
    }

  private:
    tl_ctx *h_ = nullptr;
};

namespace tsp {

struct KDPoint {  // kdtree.rs:248-252
    size_t id;
    float coords[2];
};

enum class DistanceType { Euc2D, Geo, Explicit };  // mod.rs DistanceType

class DistanceMatrix {  // distance_matrix.rs:82-212
  public:
    DistanceMatrix() = default;
    DistanceMatrix(size_t n, std::vector<float> items, const std::vector<KDPoint> &cities, DistanceType kind)
        : n_(n), items_(std::move(items)), kind_(kind)
    {
        if (items_.size() != n * (n - 1) / 2) throw std::runtime_error("distances length != n*(n-1)/2");  // :100-107
        for (size_t p = 0; p < cities.size(); ++p) city_idx_[cities[p].id] = p;
        if (city_idx_.size() != n) throw std::runtime_error("city_idx size differs from n cities");  // :99
    }
    // DistanceMatrix::build (:122-153) on the GPU
    static DistanceMatrix build(Context &ctx, const std::vector<KDPoint> &cities, DistanceType dt)
    {
        if (cities.size() < 2) throw std::runtime_error("distance matrix requires at least 2 points");
        if (dt == DistanceType::Explicit)
            throw std::runtime_error("cannot build distance matrix from coordinates for EXPLICIT type — use DistanceMatrix::new() with precomputed distances");
        std::vector<float> xy;
        for (auto &c : cities) { xy.push_back(c.coords[0]); xy.push_back(c.coords[1]); }
        const size_t n = cities.size();
        std::vector<float> items(n * (n - 1) / 2);
        ctx.check(tl_dm_build(ctx.get(), xy.data(), (uint32_t)n, dt == DistanceType::Geo ? TL_DIST_GEO : TL_DIST_EUC2D,
                              TL_DM_PACKED_LOWER, items.data(), nullptr));
        return DistanceMatrix(n, std::move(items), cities, dt);
    }
    size_t num_cities() const { return n_; }
    const std::vector<float> &distances() const { return items_; }
    DistanceType kind() const { return kind_; }
    float distance_by_pos(size_t p, size_t q) const  // :177-191
    {
        if (p == q) return 0.0f;
        if (p >= n_ || q >= n_) throw std::out_of_range("position out of range");
        const size_t from = std::max(p, q), to = std::min(p, q);
        return items_.at(from * (from - 1) / 2 + to);
    }
    float distance_between(size_t id1, size_t id2) const  // :197-212
    {
        if (id1 == id2) return 0.0f;
        return distance_by_pos(city_idx_.at(id1), city_idx_.at(id2));
    }
    bool city_id2pos(size_t id, size_t &pos) const
    {
        auto it = city_idx_.find(id);
        if (it == city_idx_.end()) return false;
        pos = it->second;
        return true;
    }

  private:
    size_t n_ = 0;
    std::vector<float> items_;
    DistanceType kind_ = DistanceType::Euc2D;
    std::unordered_map<size_t, size_t> city_idx_;
};

struct TspProblem {  // mod.rs:1731-1741
    std::vector<KDPoint> cities;
    DistanceMatrix distances;  // may be empty for plain EUC_2D: the kernels compute the identical f32 values on the fly
    DistanceType distance_type = DistanceType::Euc2D;

    std::vector<float> xy() const
    {
        std::vector<float> v;
        v.reserve(cities.size() * 2);
        for (auto &c : cities) { v.push_back(c.coords[0]); v.push_back(c.coords[1]); }
        return v;
    }
    const float *explicit_packed() const { return distance_type == DistanceType::Euc2D ? nullptr : distances.distances().data(); }
    std::vector<uint32_t> positions_of(const std::vector<size_t> &tour) const
    {
        std::unordered_map<size_t, uint32_t> idx;
        for (size_t p = 0; p < cities.size(); ++p) idx[cities[p].id] = (uint32_t)p;
        std::vector<uint32_t> out;
        for (size_t id : tour) {
            auto it = idx.find(id);
            if (it == idx.end()) throw ReferencePanic("two_opt: invalid city pair");  // two_opt.rs:37-47 .expect(...)
            out.push_back(it->second);
        }
        // a tour crosses the C ABI as n u32 values: a shorter one would be read past its end by the library
        if (out.size() != cities.size()) throw std::runtime_error("tour length differs from the number of cities");
        return out;
    }
};

struct HeuristicOptions {  // mod.rs:596-613
    size_t epochs = 10000, platoo_epochs = 500, n_nearest = 3;
    bool verbose = false;
};
struct LKOptions {  // mod.rs:1249-1267
    HeuristicOptions heuristic{100, 10, 5, false};
    size_t max_depth = 5;
};

struct Solution {  // mod.rs:1753-1814
    float total = 0.0f;
    std::vector<size_t> route_;
    tl_stats stats{};
    const std::vector<size_t> &route() const { return route_; }
};

enum class ProgressKind { PathUpdate, CityChange, Done };
using ProgressFn = std::function<void(ProgressKind, const std::vector<size_t> &, float)>;  // progress.rs ProgressMessage

inline bool validate_tour(const std::vector<size_t> &tour, const std::vector<KDPoint> &cities)  // mod.rs:1620-1634
{
    if (tour.size() != cities.size()) return false;
    std::vector<size_t> a(tour), b;
    for (auto &c : cities) b.push_back(c.id);
    std::sort(a.begin(), a.end());
    std::sort(b.begin(), b.end());
    return a == b;
}

namespace detail {
inline Solution finish(const TspProblem &p, const std::vector<uint32_t> &pos, float cost, const tl_stats &st, const ProgressFn *tx)
{
    Solution s;
    s.total = cost;
    s.stats = st;
    for (uint32_t q : pos) s.route_.push_back(p.cities[q].id);
    if (tx && *tx) {
        (*tx)(ProgressKind::PathUpdate, s.route_, cost);
        (*tx)(ProgressKind::Done, s.route_, cost);
    }
    return s;
}
}  // namespace detail

namespace two_opt {  // two_opt.rs:7-67
inline Solution solve(Context &ctx, const TspProblem &problem, const HeuristicOptions & /*_opts*/, const ProgressFn *progress_tx,
                      const std::vector<size_t> *init_tour, int mode = TL_MODE_REF_ORDER)
{
    const auto xy = problem.xy();
    const uint32_t n = (uint32_t)problem.cities.size();
    std::vector<uint32_t> init, out(n);
    if (init_tour) init = problem.positions_of(*init_tour);
    float cost = 0.f;
    tl_stats st{};
    // With a progress callback the descent also lists its moves (tl_two_opt_trace) and the reference's exact message sequence —
    // CityChange(path[i]) per outer i of every sweep, PathUpdate(path, new_distance) per move (two_opt.rs:30-32,53-56) — is
    // replayed from them once the kernel is back (coordinates within the LDS-resident descent, or problem.distances of a GEO /
    // EXPLICIT problem); otherwise the final PathUpdate (detail::finish).
    const float *packed = problem.explicit_packed();
    if (progress_tx && *progress_tx && mode == TL_MODE_REF_ORDER && n >= 3 && n <= 65535u &&
        (packed || n <= tl_two_opt_lds_max_n(ctx.get()))) {
        std::vector<size_t> route;
        std::vector<uint32_t> pos(n);
        for (uint32_t q = 0; q < n; ++q) pos[q] = init_tour ? init[q] : q;
        for (uint32_t q : pos) route.push_back(problem.cities[q].id);
        (*progress_tx)(ProgressKind::PathUpdate, route, 0.0f);  // :22-24
        uint32_t cap = std::max<uint32_t>(64u, 16u * n), len = 0;
        std::vector<uint32_t> log;
        for (;;) {
            log.assign(cap, 0u);
            ctx.check(tl_two_opt_trace(ctx.get(), xy.data(), n, packed, init_tour ? init.data() : nullptr, out.data(), &cost, &st, log.data(), cap, &len));
            if (len <= cap) break;
            cap = len;  // the descent is deterministic: once more with room for every move
        }
        auto d = [&](uint32_t p, uint32_t q) -> float {  // problem.distances, or KDPoint::distance (kdtree.rs:291-295): separate roundings, correctly rounded sqrt
            if (packed) return problem.distances.distance_by_pos(p, q);
            const volatile float dx = xy[2 * p] - xy[2 * q], dy = xy[2 * p + 1] - xy[2 * q + 1];
            const volatile float sx = dx * dx, sy = dy * dy;
            const volatile float ss = sx + sy;
            return std::sqrt((float)ss);
        };
        size_t k = 0;
        std::vector<size_t> one(1);
        for (uint64_t sw = 0; sw < st.sweeps; ++sw, k += (k < len && log[k] == TL_TRACE_SWEEP) ? 1 : 0)
            for (uint32_t i = 0; i + 3 < n; ++i) {
                one[0] = problem.cities[pos[i]].id;
                (*progress_tx)(ProgressKind::CityChange, one, 0.0f);
                while (k < len && log[k] != TL_TRACE_SWEEP && (log[k] >> 16) == i) {
                    const uint32_t j = log[k] & 0xFFFFu;
                    const float new_distance = d(pos[i], pos[j]) + d(pos[i + 1], pos[j + 1]);
                    std::reverse(pos.begin() + i + 1, pos.begin() + j + 1);
                    for (uint32_t q = 0; q < n; ++q) route[q] = problem.cities[pos[q]].id;
                    (*progress_tx)(ProgressKind::PathUpdate, route, new_distance);
                    ++k;
                }
            }
        Solution s;
        s.total = cost;
        s.stats = st;
        for (uint32_t q : out) s.route_.push_back(problem.cities[q].id);
        (*progress_tx)(ProgressKind::Done, s.route_, cost);
        return s;
    }
    ctx.check(tl_two_opt(ctx.get(), xy.data(), n, problem.explicit_packed(), init_tour ? init.data() : nullptr, mode, out.data(), &cost, &st));
    return detail::finish(problem, out, cost, st, progress_tx);
}

// A population of tours, each refined by its own descent, all concurrently (tl_two_opt_population);
// element k equals solve(ctx, problem, {}, nullptr, &init_tours[k]).
inline std::vector<Solution> solve_population(Context &ctx, const TspProblem &problem, const std::vector<std::vector<size_t>> &init_tours)
{
    const auto xy = problem.xy();
    const uint32_t n = (uint32_t)problem.cities.size(), count = (uint32_t)init_tours.size();
    std::vector<uint32_t> init((size_t)count * n), out((size_t)count * n);
    for (uint32_t k = 0; k < count; ++k) {
        const auto pos = problem.positions_of(init_tours[k]);
        std::copy(pos.begin(), pos.end(), init.begin() + (size_t)k * n);
    }
    std::vector<float> costs(count);
    tl_stats st{};
    ctx.check(tl_two_opt_population(ctx.get(), xy.data(), n, problem.explicit_packed(), init.data(), count, out.data(), costs.data(), &st));
    std::vector<Solution> res;
    for (uint32_t k = 0; k < count; ++k) {
        std::vector<uint32_t> one(out.begin() + (size_t)k * n, out.begin() + (size_t)(k + 1) * n);
        res.push_back(detail::finish(problem, one, costs[k], st, nullptr));
    }
    return res;
}
}  // namespace two_opt

namespace three_opt {  // three_opt.rs:16-51
// three_opt.rs:186-218 apply_3opt: cases 1-3 reverse segments in place, 4-7 swap path[i+1..=j] and path[j+1..=k] with either reversed
inline void apply_3opt(std::vector<uint32_t> &path, uint32_t i, uint32_t j, uint32_t k, uint32_t kase)
{
    auto b1 = path.begin() + i + 1, e1 = path.begin() + j + 1, e2 = path.begin() + k + 1;
    switch (kase) {
    case 1: std::reverse(b1, e1); break;
    case 2: std::reverse(e1, e2); break;
    case 3: std::reverse(b1, e1); std::reverse(e1, e2); break;
    case 4: case 5: case 6: case 7: {
        std::vector<uint32_t> s1(b1, e1), s2(e1, e2);
        if (kase == 5 || kase == 7) std::reverse(s1.begin(), s1.end());
        if (kase == 6 || kase == 7) std::reverse(s2.begin(), s2.end());
        std::copy(s2.begin(), s2.end(), b1);
        std::copy(s1.begin(), s1.end(), b1 + (std::ptrdiff_t)s2.size());
        break;
    }
    default: throw std::runtime_error("apply_3opt: case must be 1-7");
    }
}

inline Solution solve(Context &ctx, const TspProblem &problem, const HeuristicOptions &, const ProgressFn *progress_tx,
                      const std::vector<size_t> *init_tour)
{
    const auto xy = problem.xy();
    const uint32_t n = (uint32_t)problem.cities.size();
    std::vector<uint32_t> init, out(n);
    if (init_tour) init = problem.positions_of(*init_tour);
    float cost = 0.f;
    tl_stats st{};
    if (!(progress_tx && *progress_tx) || n < 4) {  // (n < 4: the reference returns before its first message, three_opt.rs:25-28)
        ctx.check(tl_three_opt(ctx.get(), xy.data(), n, problem.explicit_packed(), init_tour ? init.data() : nullptr, out.data(), &cost, &st));
        return detail::finish(problem, out, cost, st, nullptr);
    }
    // With a progress callback the solve also lists its moves (tl_three_opt_trace) and the reference's message sequence — the
    // start path, the path after every apply_3opt (each with 0.0), Done (three_opt.rs:34,42,47-49) — is replayed from them.
    uint32_t cap = std::max<uint32_t>(64u, 4u * n), len = 0;
    std::vector<uint32_t> log;
    for (;;) {
        log.assign((size_t)cap * 4, 0u);
        ctx.check(tl_three_opt_trace(ctx.get(), xy.data(), n, problem.explicit_packed(), init_tour ? init.data() : nullptr, out.data(), &cost, &st,
                                     log.data(), cap, &len));
        if (len <= cap) break;
        cap = len;  // deterministic: once more with room for every move
    }
    std::vector<uint32_t> pos(n);
    for (uint32_t q = 0; q < n; ++q) pos[q] = init_tour ? init[q] : q;
    std::vector<size_t> route(n);
    auto send = [&]() {
        for (uint32_t q = 0; q < n; ++q) route[q] = problem.cities[pos[q]].id;
        (*progress_tx)(ProgressKind::PathUpdate, route, 0.0f);
    };
    send();
    for (uint32_t m = 0; m < len; ++m) {
        const uint32_t i = log[4 * m], j = log[4 * m + 1], k = log[4 * m + 2], kase = log[4 * m + 3];
        apply_3opt(pos, i, j, k, kase);
        send();
    }
    Solution s = detail::finish(problem, out, cost, st, nullptr);
    (*progress_tx)(ProgressKind::Done, s.route_, cost);
    return s;
}
}  // namespace three_opt

namespace or_opt {  // or_opt.rs:18-74
// or_opt.rs:172-184 apply_relocation: drain the segment, insert it after the old index j, forward or reversed
inline void apply_relocation(std::vector<uint32_t> &tour, size_t i, size_t seg_len, size_t j, bool reversed)
{
    std::vector<uint32_t> seg(tour.begin() + i, tour.begin() + i + seg_len);
    tour.erase(tour.begin() + i, tour.begin() + i + seg_len);
    const size_t at = j >= i + seg_len ? j - seg_len + 1 : j + 1;
    if (reversed) std::reverse(seg.begin(), seg.end());
    tour.insert(tour.begin() + at, seg.begin(), seg.end());
}

// DistanceMatrix::tour_length (distance_matrix.rs:235-245) on positions, on the host, in the reference's f32 order: the closing
// edge first, then every window — through problem.distances where the problem has a matrix, else KDPoint::distance
inline float tour_length_f32(const TspProblem &problem, const std::vector<float> &xy, const std::vector<uint32_t> &pos)
{
    if (pos.size() < 2) return 0.0f;
    const bool have_dm = problem.distances.num_cities() == pos.size();
    auto d = [&](uint32_t p, uint32_t q) -> float {
        if (have_dm) return problem.distances.distance_by_pos(p, q);
        if (p == q) return 0.0f;
        const volatile float dx = xy[2 * p] - xy[2 * q], dy = xy[2 * p + 1] - xy[2 * q + 1];
        const volatile float sx = dx * dx, sy = dy * dy;
        const volatile float ss = sx + sy;
        return std::sqrt((float)ss);
    };
    volatile float total = d(pos.back(), pos.front());
    for (size_t k = 0; k + 1 < pos.size(); ++k) total = total + d(pos[k], pos[k + 1]);
    return total;
}

inline Solution solve(Context &ctx, const TspProblem &problem, const HeuristicOptions &, const ProgressFn *progress_tx,
                      const std::vector<size_t> *init_tour)
{
    const auto xy = problem.xy();
    const uint32_t n = (uint32_t)problem.cities.size();
    std::vector<uint32_t> init, out(n);
    if (init_tour) init = problem.positions_of(*init_tour);
    float cost = 0.f;
    tl_stats st{};
    if (!(progress_tx && *progress_tx) || n < 4) {  // (n < 4: the reference returns before its first message, or_opt.rs:31-34)
        ctx.check(tl_or_opt(ctx.get(), xy.data(), n, problem.explicit_packed(), init_tour ? init.data() : nullptr, out.data(), &cost, &st));
        return detail::finish(problem, out, cost, st, nullptr);
    }
    // With a progress callback the solve also lists its moves (tl_or_opt_trace) and the reference's message sequence — the start
    // path (0.0), the path and its tour_length after every apply_relocation, Done (or_opt.rs:40-42,62-67,70-72) — is replayed.
    uint32_t cap = std::max<uint32_t>(64u, 4u * n), len = 0;
    std::vector<uint32_t> log;
    for (;;) {
        log.assign((size_t)cap * 4, 0u);
        ctx.check(tl_or_opt_trace(ctx.get(), xy.data(), n, problem.explicit_packed(), init_tour ? init.data() : nullptr, out.data(), &cost, &st,
                                  log.data(), cap, &len));
        if (len <= cap) break;
        cap = len;  // deterministic: once more with room for every move
    }
    std::vector<uint32_t> pos(n);
    for (uint32_t q = 0; q < n; ++q) pos[q] = init_tour ? init[q] : q;
    std::vector<size_t> route(n);
    auto send = [&](float d) {
        for (uint32_t q = 0; q < n; ++q) route[q] = problem.cities[pos[q]].id;
        (*progress_tx)(ProgressKind::PathUpdate, route, d);
    };
    send(0.0f);
    for (uint32_t m = 0; m < len; ++m) {
        apply_relocation(pos, log[4 * m], log[4 * m + 2], log[4 * m + 1], log[4 * m + 3] != 0u);
        send(tour_length_f32(problem, xy, pos));
    }
    Solution s = detail::finish(problem, out, cost, st, nullptr);
    (*progress_tx)(ProgressKind::Done, s.route_, cost);
    return s;
}
}  // namespace or_opt

namespace nearest_neighbor {  // nearest_neighbor.rs:8-76
inline Solution solve(Context &ctx, const TspProblem &problem, const HeuristicOptions &opts, const ProgressFn *progress_tx,
                      const std::vector<size_t> * /*_init_tour*/)
{
    if (opts.n_nearest == 0) throw std::runtime_error("n_nearest must be >= 1");  // mod.rs:677-682
    const auto xy = problem.xy();
    const uint32_t n = (uint32_t)problem.cities.size();
    std::vector<uint32_t> out(n);
    float cost = 0.f;
    // GEO / EXPLICIT: the walk reads problem.distances (distance_matrix.rs:259-297)
    ctx.check(tl_nearest_neighbor(ctx.get(), xy.data(), problem.explicit_packed(), n, (uint32_t)opts.n_nearest, out.data(), &cost));
    Solution s = detail::finish(problem, out, cost, tl_stats{}, nullptr);
    if (progress_tx && *progress_tx && n > 0) {
        // the reference's messages follow from the finished walk (nearest_neighbor.rs:32-34,40-42,67-69,72-74): the start city,
        // then per step CityChange(current city) and the path so far, then Done
        std::vector<size_t> path{s.route_[0]}, one(1);
        (*progress_tx)(ProgressKind::PathUpdate, path, 0.0f);
        for (uint32_t t = 1; t < n; ++t) {
            one[0] = s.route_[t - 1];
            (*progress_tx)(ProgressKind::CityChange, one, 0.0f);
            path.push_back(s.route_[t]);
            (*progress_tx)(ProgressKind::PathUpdate, path, 0.0f);
        }
        (*progress_tx)(ProgressKind::Done, s.route_, cost);
    }
    return s;
}
}  // namespace nearest_neighbor

namespace lin_kernighan {  // lin_kernighan.rs:35-100
inline Solution solve(Context &ctx, const TspProblem &problem, const LKOptions &opts, const ProgressFn *progress_tx,
                      const std::vector<size_t> *init_tour, uint64_t seed = 1)
{
    if (opts.heuristic.n_nearest == 0) throw std::runtime_error("n_nearest must be >= 1");  // mod.rs:677-682
    if (opts.max_depth == 0) throw std::runtime_error("max_depth must be >= 1");            // mod.rs:1270-1276
    const auto xy = problem.xy();
    const uint32_t n = (uint32_t)problem.cities.size();
    std::vector<uint32_t> init, out(n);
    if (init_tour) init = problem.positions_of(*init_tour);
    tl_lk_opts o{(uint32_t)opts.heuristic.epochs, (uint32_t)opts.heuristic.platoo_epochs, (uint32_t)opts.heuristic.n_nearest, (uint32_t)opts.max_depth};
    float cost = 0.f;
    tl_stats st{};
    // problem.distances (GEO / EXPLICIT) feeds the NN seed and the reported total only; the search is Euclidean (lin_kernighan.rs:41,47-55,99)
    if (!(progress_tx && *progress_tx) || n < 4) {  // (n < 4: the reference returns before its first message, lin_kernighan.rs:57-59)
        ctx.check(tl_lk(ctx.get(), xy.data(), n, problem.explicit_packed(), init_tour ? init.data() : nullptr, &o, seed, out.data(), &cost, &st));
        return detail::finish(problem, out, cost, st, nullptr);
    }
    // With a progress callback: the best tours the search settles on, in order, handed on WHILE the search runs (tl_lk_live) as the
    // reference sends them — PathUpdate(best_tour, best_dist) after the first lk_pass and per improving epoch, no Done
    // (lin_kernighan.rs:71,90).
    struct Live {
        const TspProblem *problem;
        const ProgressFn *tx;
        std::vector<size_t> route;
        std::exception_ptr err;
    } live{&problem, progress_tx, std::vector<size_t>(n), nullptr};
    auto on_best = [](void *user, const uint32_t *pos, uint32_t nn, float best_dist) {
        Live *L = static_cast<Live *>(user);
        if (L->err) return;
        try {  // an exception must not unwind through the C frames of the library
            for (uint32_t q = 0; q < nn; ++q) L->route[q] = L->problem->cities[pos[q]].id;
            (*L->tx)(ProgressKind::PathUpdate, L->route, best_dist);
        } catch (...) {
            L->err = std::current_exception();
        }
    };
    ctx.check(tl_lk_live(ctx.get(), xy.data(), n, problem.explicit_packed(), init_tour ? init.data() : nullptr, &o, seed, out.data(), &cost, &st,
                         +on_best, &live));
    if (live.err) std::rethrow_exception(live.err);
    return detail::finish(problem, out, cost, st, nullptr);
}
}  // namespace lin_kernighan

namespace tsplib {  // tsplib.rs:142-255

struct TspLibData {
    std::string name, comment;
    std::vector<KDPoint> cities;
    size_t dimension = 0;
    std::vector<float> raw_distances;
    bool has_raw = false;
    DistanceType distance_type = DistanceType::Euc2D;

    TspProblem problem(Context &ctx) const  // TspLibData::distance_matrix (:84-98) + TspProblem::new
    {
        TspProblem p;
        p.cities = cities;
        if (has_raw) {
            p.distances = DistanceMatrix(cities.size(), raw_distances, cities, DistanceType::Explicit);
            p.distance_type = DistanceType::Explicit;
        } else if (distance_type == DistanceType::Geo) {
            p.distances = DistanceMatrix::build(ctx, cities, DistanceType::Geo);
            p.distance_type = DistanceType::Geo;
        }
        return p;
    }
};

inline bool parse_f32(const std::string &t, float &v)
{
    char *end = nullptr;
    v = std::strtof(t.c_str(), &end);  // f32::from_str: parse straight to f32, not strtod + cast
    return end && *end == '\0' && end != t.c_str();
}

inline TspLibData read_from_string(const std::string &text)
{
    std::map<std::string, std::string> meta;
    TspLibData d;
    std::vector<float> weights;
    enum { Start, In, Out, End } state = Start;
    std::string section, raw;
    std::istringstream in(text);
    size_t line_no = 1;
    auto is_section = [](const std::string &l) {
        if (l.empty() || !(std::isupper((unsigned char)l[0]) || l[0] == '_')) return false;
        for (char ch : l)
            if (!(std::isalnum((unsigned char)ch) || ch == '_')) return false;
        return true;
    };
    while (std::getline(in, raw)) {
        size_t b = raw.find_first_not_of(" \t\r\n"), e = raw.find_last_not_of(" \t\r\n");
        std::string line = b == std::string::npos ? "" : raw.substr(b, e - b + 1);
        for (auto &ch : line) ch = (char)std::toupper((unsigned char)ch);
        ++line_no;
        if (state == End) break;
        if (is_section(line)) {
            if (line == "EOF") state = End;
            else { state = In; section = line; }
            continue;
        }
        std::istringstream ls(line);
        std::vector<std::string> tok;
        for (std::string t; ls >> t;) tok.push_back(t);
        if (state == Start) {
            size_t c = line.find(':');
            if (c == std::string::npos) throw std::runtime_error("Failed to extract meta data on line." + std::to_string(line_no));
            auto trim = [](std::string s) {
                size_t b2 = s.find_first_not_of(" \t"), e2 = s.find_last_not_of(" \t");
                return b2 == std::string::npos ? std::string() : s.substr(b2, e2 - b2 + 1);
            };
            meta[trim(line.substr(0, c))] = trim(line.substr(c + 1));
        } else if (state == In && (section == "NODE_COORD_SECTION" || section == "DISPLAY_DATA_SECTION")) {
            float f;
            if (tok.empty() || !parse_f32(tok[0], f)) throw std::runtime_error("Failed to extract coordinates on line." + std::to_string(line_no));
            if (tok.size() < 3) throw std::runtime_error("KDPoint requires at least 2 coordinates");
            KDPoint p{};
            p.id = (size_t)std::stoull(tok[0]);
            if (!parse_f32(tok[1], p.coords[0]) || !parse_f32(tok[2], p.coords[1]))
                throw std::runtime_error("Error on line." + std::to_string(line_no) + " - invalid number");
            d.cities.push_back(p);
        } else if (state == In && section == "EDGE_WEIGHT_SECTION") {
            for (auto &t : tok) {
                float f;
                if (parse_f32(t, f)) weights.push_back(f);
            }
        }
    }
    if (meta.count("TYPE") && meta["TYPE"] == "ATSP") throw std::runtime_error("ATSP (asymmetric TSP) is not supported");
    const std::string ewt = meta.count("EDGE_WEIGHT_TYPE") ? meta["EDGE_WEIGHT_TYPE"] : "";
    d.distance_type = ewt == "GEO" ? DistanceType::Geo : (ewt == "EXPLICIT" ? DistanceType::Explicit : DistanceType::Euc2D);  // unknown -> EUC_2D (:199-202)
    d.dimension = meta.count("DIMENSION") ? (size_t)std::strtoull(meta["DIMENSION"].c_str(), nullptr, 10) : 0;
    if (!weights.empty()) {
        const size_t n = d.dimension;
        const std::string fmt = meta.count("EDGE_WEIGHT_FORMAT") ? meta["EDGE_WEIGHT_FORMAT"] : "";
        std::vector<float> full(n * n, 0.0f);
        if (fmt == "FULL_MATRIX") {
            if (weights.size() != n * n) throw std::runtime_error("FULL_MATRIX: expected " + std::to_string(n * n) + " tokens, got " + std::to_string(weights.size()));
            full = weights;
        } else if (fmt == "UPPER_ROW") {
            if (weights.size() != n * (n - 1) / 2) throw std::runtime_error("UPPER_ROW: wrong token count");
            size_t k = 0;
            for (size_t i = 0; i + 1 < n; ++i)
                for (size_t j = i + 1; j < n; ++j) full[i * n + j] = full[j * n + i] = weights[k++];
        } else if (fmt == "LOWER_DIAG_ROW") {
            if (weights.size() != n * (n + 1) / 2) throw std::runtime_error("LOWER_DIAG_ROW: wrong token count");
            size_t k = 0;
            for (size_t i = 0; i < n; ++i)
                for (size_t j = 0; j <= i; ++j) full[i * n + j] = weights[k++];
        } else {
            throw std::runtime_error("Unsupported EDGE_WEIGHT_FORMAT: " + fmt);
        }
        for (size_t i = 1; i < n; ++i)
            for (size_t j = 0; j < i; ++j) d.raw_distances.push_back(full[i * n + j]);
        d.has_raw = true;
    }
    if (d.cities.empty() && d.has_raw) {  // grid placeholder coordinates (:246-258)
        const size_t cols = (size_t)std::ceil(std::sqrt((double)d.dimension));
        for (size_t i = 0; i < d.dimension; ++i) d.cities.push_back(KDPoint{i + 1, {(float)(i % cols), (float)(i / cols)}});
    }
    if (d.cities.empty()) throw std::runtime_error("Found no valid city coordinates");
    auto lower = [](std::string s) { for (auto &ch : s) ch = (char)std::tolower((unsigned char)ch); return s; };
    d.name = lower(meta.count("NAME") ? meta["NAME"] : "unspecified");
    d.comment = lower(meta.count("COMMENT") ? meta["COMMENT"] : "unspecified");
    return d;
}

inline TspLibData read_from_file(const std::string &path)
{
    std::ifstream f(path);
    if (!f) throw std::runtime_error("tsplib: failed to read file");
    std::stringstream ss;
    ss << f.rdbuf();
    return read_from_string(ss.str());
}

}  // namespace tsplib

namespace opt_tour {  // opt_tour.rs:12-107
struct OptTour {
    std::string name, comment;
    size_t dimension = 0;
    std::vector<size_t> route;
};

inline OptTour read_from_string(const std::string &text)
{
    std::map<std::string, std::string> meta;
    OptTour t;
    enum { Header, TourSection, End } state = Header;
    std::istringstream in(text);
    std::string raw;
    while (std::getline(in, raw)) {
        size_t b = raw.find_first_not_of(" \t\r\n"), e = raw.find_last_not_of(" \t\r\n");
        std::string line = b == std::string::npos ? "" : raw.substr(b, e - b + 1);
        for (auto &ch : line) ch = (char)std::toupper((unsigned char)ch);
        if (line == "EOF" || state == End) break;
        if (state == Header) {
            if (line == "TOUR_SECTION") {
                state = TourSection;
                continue;
            }
            // ^(\w+)\s*:\s*(.+)$
            size_t k = 0;
            while (k < line.size() && (std::isalnum((unsigned char)line[k]) || line[k] == '_')) ++k;
            size_t c = k;
            while (c < line.size() && std::isspace((unsigned char)line[c])) ++c;
            if (k == 0 || c >= line.size() || line[c] != ':') continue;
            size_t v = c + 1;
            while (v < line.size() && std::isspace((unsigned char)line[v])) ++v;
            if (v >= line.size()) continue;
            meta[line.substr(0, k)] = line.substr(v);
        } else {
            std::istringstream ls(line);
            for (std::string tok; ls >> tok;) {
                char *end = nullptr;
                const long long id = std::strtoll(tok.c_str(), &end, 10);
                if (!end || *end != '\0' || end == tok.c_str()) continue;  // isize::from_str failed: ignored
                if (id == -1) {
                    state = End;
                    break;
                }
                if (id > 0) t.route.push_back((size_t)id);
            }
        }
    }
    const std::string ty = meta.count("TYPE") ? meta["TYPE"] : "";
    if (ty != "TOUR") throw std::runtime_error("opt_tour: expected TYPE : TOUR, found TYPE : " + ty);
    t.dimension = 0;
    if (meta.count("DIMENSION")) {
        char *end = nullptr;
        const unsigned long long d = std::strtoull(meta["DIMENSION"].c_str(), &end, 10);
        if (end && *end == '\0' && end != meta["DIMENSION"].c_str()) t.dimension = (size_t)d;
    }
    if (t.route.size() != t.dimension)
        throw std::runtime_error("opt_tour: dimension mismatch — DIMENSION=" + std::to_string(t.dimension) + " but parsed " +
                                 std::to_string(t.route.size()) + " cities");
    t.name = meta.count("NAME") ? meta["NAME"] : "unknown";
    t.comment = meta.count("COMMENT") ? meta["COMMENT"] : "";
    return t;
}

inline OptTour read_from_file(const std::string &path)
{
    std::ifstream f(path);
    if (!f) throw std::runtime_error("opt_tour: cannot open file: " + path);
    std::stringstream ss;
    ss << f.rdbuf();
    return read_from_string(ss.str());
}
}  // namespace opt_tour

namespace random_shuffle {  // random_shuffle.rs:12-27, seeded (the reference draws from an unseeded thread RNG, :20)
// Fisher-Yates `for i in (1..n).rev(): j = rng % (i+1); swap` driven by splitmix64 seeded with `seed` — the same stream the
// device draws restart 0 of `seed` from (two_opt_ref.hip, oracle tlo_restart_perm), so a `shuffle` stage and a multi-start
// restart agree.
inline Solution solve(Context &ctx, const TspProblem &problem, uint64_t seed)
{
    const uint32_t n = (uint32_t)problem.cities.size();
    std::vector<uint32_t> perm(n);
    for (uint32_t i = 0; i < n; ++i) perm[i] = i;
    uint64_t st = seed;
    for (uint32_t i = n; i-- > 1;) {
        uint64_t z = (st += 0x9E3779B97F4A7C15ULL);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
        z ^= z >> 31;
        std::swap(perm[i], perm[(uint32_t)(z % ((uint64_t)i + 1))]);
    }
    float cost = 0.f;
    if (n >= 2) {
        const auto xy = problem.xy();
        ctx.check(tl_tour_length(ctx.get(), problem.explicit_packed() ? nullptr : xy.data(), problem.explicit_packed(), n, perm.data(), &cost));
    }
    return detail::finish(problem, perm, cost, tl_stats{}, nullptr);
}
}  // namespace random_shuffle

// Solvers (mod.rs:47-72) this build accelerates, by the reference's names and aliases (FromStr, mod.rs:559-590)
enum class Solvers { NearestNeighbor, TwoOpt, ThreeOpt, OrOpt, LinKernighan, RandomShuffle };

inline bool solver_from_str(std::string s, Solvers &out, std::string &why)
{
    for (auto &ch : s) ch = (char)std::tolower((unsigned char)ch);
    if (s == "nn" || s == "nearest_neighbor") out = Solvers::NearestNeighbor;
    else if (s == "2opt" || s == "two_opt") out = Solvers::TwoOpt;
    else if (s == "3opt" || s == "three_opt") out = Solvers::ThreeOpt;
    else if (s == "or_opt" || s == "or-opt" || s == "oropt") out = Solvers::OrOpt;
    else if (s == "lk" || s == "lin_kernighan") out = Solvers::LinKernighan;
    else if (s == "shuffle" || s == "random_shuffle") out = Solvers::RandomShuffle;
    else {
        static const char *cpu_only[] = {"aco", "ant_colony", "bhk", "bellman_karp", "branch_bound", "christofides", "chr", "sav", "savings",
                                         "cs", "cuckoo_search", "fpa", "flower_pollination", "fourier", "ga", "genetic_algorithm", "gsa",
                                         "gravitational_search", "gec", "greedy_edge", "pso", "particle_swarm", "sa", "simulated_annealing",
                                         "som", "kohonen", "kohonen_som", "stochastic_hill", "tabu", "tabu_search"};
        for (const char *c : cpu_only)
            if (s == c) {
                why = "solver `" + s + "` is not accelerated by this build (nn, 2opt, 3opt, or_opt, lk, shuffle are)";
                return false;
            }
        why = "unknown solver";  // FromStr's Err (mod.rs:588)
        return false;
    }
    return true;
}

inline const char *solver_name(Solvers s)
{
    switch (s) {
        case Solvers::NearestNeighbor: return "nn";
        case Solvers::TwoOpt: return "2opt";
        case Solvers::ThreeOpt: return "3opt";
        case Solvers::OrOpt: return "or_opt";
        case Solvers::LinKernighan: return "lk";
        default: return "shuffle";
    }
}

namespace pipeline {  // pipeline.rs
struct StageOptions {
    HeuristicOptions heuristic;
    LKOptions lk;
    int two_opt_mode = TL_MODE_REF_ORDER;
    uint64_t seed = 1;  // LK kicks, shuffle
    const ProgressFn *progress = nullptr;  // handed to every stage's solve() (the reference's CLI passes None; teeline-qt a sender)
};
struct StageOutcome {  // :11-14
    Solvers solver;
    Solution solution;
    uint64_t duration_ms = 0;
    double duration_us = 0.0;  // the same interval, finer (the CLI's --timing)
};

// stage_warnings (:92-132), for the solvers this build knows
inline std::vector<std::string> stage_warnings(const std::vector<Solvers> &solvers)
{
    std::vector<std::string> w;
    for (size_t i = 1; i < solvers.size(); ++i)
        if (solvers[i] == Solvers::NearestNeighbor)
            w.push_back("nn at stage " + std::to_string(i) + " discards the warm-start seed from the previous stage");
    return w;
}

// run_pipeline_stages (:53-80): stage k+1 is warm-started with stage k's tour; an invalid seed is dropped with a warning,
// an invalid stage result is a hard error
inline std::vector<StageOutcome> run_pipeline_stages(Context &ctx, const TspProblem &problem, const std::vector<Solvers> &stages,
                                                     const StageOptions &o)
{
    if (stages.empty()) throw std::runtime_error("pipeline has no stages");
    std::vector<StageOutcome> out;
    std::vector<size_t> seed;
    bool have_seed = false;
    for (Solvers s : stages) {
        if (have_seed && !validate_tour(seed, problem.cities)) {
            std::fprintf(stderr, "warning: pipeline: invalid seed; using default seeding\n");
            have_seed = false;
        }
        const std::vector<size_t> *init = have_seed ? &seed : nullptr;
        const auto t0 = std::chrono::steady_clock::now();
        Solution sol;
        switch (s) {
            case Solvers::NearestNeighbor: sol = nearest_neighbor::solve(ctx, problem, o.heuristic, o.progress, init); break;
            case Solvers::TwoOpt: sol = two_opt::solve(ctx, problem, o.heuristic, o.progress, init, o.two_opt_mode); break;
            case Solvers::ThreeOpt: sol = three_opt::solve(ctx, problem, o.heuristic, o.progress, init); break;
            case Solvers::OrOpt: sol = or_opt::solve(ctx, problem, o.heuristic, o.progress, init); break;
            case Solvers::LinKernighan: sol = lin_kernighan::solve(ctx, problem, o.lk, o.progress, init, o.seed); break;
            case Solvers::RandomShuffle: sol = random_shuffle::solve(ctx, problem, o.seed); break;
        }
        const auto t1 = std::chrono::steady_clock::now();
        const uint64_t ms = (uint64_t)std::chrono::duration_cast<std::chrono::milliseconds>(t1 - t0).count();
        if (!validate_tour(sol.route(), problem.cities))
            throw std::runtime_error(std::string("stage ") + solver_name(s) + " invalid tour");  // :70-71
        seed = sol.route();
        have_seed = true;
        out.push_back(StageOutcome{s, std::move(sol), ms, std::chrono::duration<double, std::micro>(t1 - t0).count()});
    }
    return out;
}
}  // namespace pipeline

namespace cli {  // teeline-cli/src/main.rs output helpers
// print_solution (main.rs:645-652): "{:.5} {flag}\n", then every id followed by ONE space, then "\n"
inline std::string format_solution(const Solution &tour, bool is_optimized)
{
    char head[64];
    std::snprintf(head, sizeof(head), "%.5f %d\n", tour.total, is_optimized ? 1 : 0);
    std::string s = head;
    for (size_t id : tour.route()) {
        s += std::to_string(id);
        s += ' ';
    }
    s += '\n';
    return s;
}

struct OptimalComparison {  // main.rs:654-658
    float optimal_cost = 0.f, gap_pct = 0.f;
    std::string opt_name;
};

// compute_optimal_comparison (main.rs:660-684): the optimal tour's cost through the same tour_length; false on a dimension mismatch
inline bool compute_optimal_comparison(Context &ctx, float solver_cost, const TspProblem &problem, const opt_tour::OptTour &ot,
                                       OptimalComparison &cmp)
{
    if (ot.dimension != problem.cities.size()) {
        std::fprintf(stderr, "--optimal-tour: dimension mismatch (%zu vs %zu); skipping comparison\n", ot.dimension, problem.cities.size());
        return false;
    }
    // DistanceMatrix::tour_length (distance_matrix.rs:221-233): 0.0 for fewer than 2 cities or an unknown id
    float optimal = 0.f;
    std::unordered_map<size_t, uint32_t> idx;
    for (size_t p = 0; p < problem.cities.size(); ++p) idx[problem.cities[p].id] = (uint32_t)p;
    std::vector<uint32_t> pos;
    bool known = ot.route.size() >= 2;
    for (size_t id : ot.route) {
        auto it = idx.find(id);
        if (it == idx.end()) {
            known = false;
            break;
        }
        pos.push_back(it->second);
    }
    if (known) {
        const auto xy = problem.xy();
        ctx.check(tl_tour_length(ctx.get(), problem.explicit_packed() ? nullptr : xy.data(), problem.explicit_packed(),
                                 (uint32_t)pos.size(), pos.data(), &optimal));
    }
    cmp.optimal_cost = optimal;
    cmp.gap_pct = optimal > 0.0f ? (solver_cost - optimal) / optimal * 100.0f : 0.0f;  // f32 arithmetic, as the reference
    cmp.opt_name = ot.name;
    return true;
}

// print_optimal_comparison (main.rs:686-698), to stderr
inline std::string format_optimal_comparison(float solver_cost, const OptimalComparison &cmp)
{
    char buf[256];
    std::string s = "--- Comparison ---\n";
    std::snprintf(buf, sizeof(buf), "Optimal  : %.5f  (from %s)\n", cmp.optimal_cost, cmp.opt_name.c_str());
    s += buf;
    std::snprintf(buf, sizeof(buf), "Solver   : %.5f\n", solver_cost);
    s += buf;
    if (std::fabs(cmp.gap_pct) < 0.001f) s += "Gap      : 0.00 % (matches optimal)\n";
    else {
        std::snprintf(buf, sizeof(buf), "Gap      : %+.2f %%\n", cmp.gap_pct);
        s += buf;
    }
    return s;
}

// serde_json's rendering of an f32 inside json!(): widened to f64, shortest round-trip decimal, always with a fraction or
// an exponent ("60.0", never "60")
inline std::string json_f32(float v)
{
    char buf[64];
    auto r = std::to_chars(buf, buf + sizeof(buf), (double)v);
    std::string s(buf, r.ptr);
    if (s.find_first_of(".eE") == std::string::npos) s += ".0";
    return s;
}

// print_solution_json (main.rs:700-712): serde_json::Value prints object keys sorted (BTreeMap: no preserve_order), compact
inline std::string format_solution_json(const Solution &tour, bool is_optimized, const OptimalComparison *opt)
{
    std::string s = "{\"cost\":" + json_f32(tour.total);
    if (opt) s += ",\"gap_pct\":" + json_f32(opt->gap_pct) + ",\"optimal_cost\":" + json_f32(opt->optimal_cost);
    s += std::string(",\"optimized\":") + (is_optimized ? "true" : "false") + ",\"route\":[";
    for (size_t k = 0; k < tour.route().size(); ++k) {
        if (k) s += ',';
        s += std::to_string(tour.route()[k]);
    }
    s += "]}\n";
    return s;
}
}  // namespace cli
}  // namespace tsp
}  // namespace teeline
