// teeline-gpu — minimal CLI façade over the GPU path, mirroring `teeline solve <solver> -i file` /
// `teeline pipeline --steps=nn,2opt -i file` of teeline-cli (teeline-cli/src/main.rs:371-530) for the solvers
// this build accelerates.  Output format = print_solution (main.rs:645-652): "{:.5} {0|1}\n<ids…>\n".
//
//   teeline-gpu solve 2opt|3opt|oropt|lk|nn [-i file] [--no-seed] [--best-sweep] [--seed S] [--epochs E] [--platoo-epochs P]
//                                     [--n-nearest K] [--max-depth D] [--device N] [--stats]
// Like the reference, 2opt/3opt/lk are auto-seeded with a nearest-neighbour stage (mod.rs:129-139) unless --no-seed.
#include "teeline_gpu.hpp"

#include <cstdio>
#include <cstring>
#include <iostream>

using namespace teeline;
using namespace teeline::tsp;

int main(int argc, char **argv)
{
    try {
        if (argc < 3 || (std::string(argv[1]) != "solve")) {
            std::fprintf(stderr, "usage: teeline-gpu solve 2opt|3opt|oropt|lk|nn -i file.tsp [--no-seed] [--best-sweep] [--seed S] [--stats]\n");
            return 2;
        }
        const std::string solver = argv[2];
        std::string file;
        bool no_seed = false, best = false, stats = false;
        int device = 0;
        uint64_t seed = 1;
        HeuristicOptions h;
        LKOptions lk;
        lk.heuristic = h;  // LKOptions::from_cli: heuristic = HeuristicOptions::from_cli (mod.rs:1321-1325)
        for (int a = 3; a < argc; ++a) {
            const std::string s = argv[a];
            auto val = [&]() -> std::string {
                if (a + 1 >= argc) throw std::runtime_error(s + ": missing value");
                return argv[++a];
            };
            if (s == "-i" || s == "--input") file = val();
            else if (s == "--no-seed") no_seed = true;
            else if (s == "--best-sweep") best = true;
            else if (s == "--stats") stats = true;
            else if (s == "--seed") seed = std::stoull(val());
            else if (s == "--device") device = std::stoi(val());
            else if (s == "--epochs") h.epochs = lk.heuristic.epochs = std::stoull(val());
            else if (s == "--platoo-epochs") h.platoo_epochs = lk.heuristic.platoo_epochs = std::stoull(val());
            else if (s == "--n-nearest") h.n_nearest = lk.heuristic.n_nearest = std::stoull(val());
            else if (s == "--max-depth") lk.max_depth = std::stoull(val());
            else throw std::runtime_error("unknown argument " + s);
        }
        if (h.n_nearest == 0) throw std::runtime_error("n_nearest must be >= 1");
        std::string text;
        tsplib::TspLibData data = file.empty() ? tsplib::read_from_string(std::string(std::istreambuf_iterator<char>(std::cin), {}))
                                              : tsplib::read_from_file(file);
        Context ctx(device);
        TspProblem problem = data.problem(ctx);
        // run_pipeline_stages (pipeline.rs:53-80): stage k+1 is warm-started with stage k's tour after validate_tour
        std::vector<size_t> seed_route;
        const std::vector<size_t> *init = nullptr;
        if (solver != "nn" && !no_seed && problem.distance_type == DistanceType::Euc2D) {
            Solution nn = nearest_neighbor::solve(ctx, problem, h, nullptr, nullptr);
            if (!validate_tour(nn.route(), problem.cities)) throw std::runtime_error("pipeline: seed stage produced an invalid tour");
            seed_route = nn.route();
            init = &seed_route;
        }
        Solution sol;
        if (solver == "2opt") sol = two_opt::solve(ctx, problem, h, nullptr, init, best ? TL_MODE_BEST_SWEEP : TL_MODE_REF_ORDER);
        else if (solver == "3opt") sol = three_opt::solve(ctx, problem, h, nullptr, init);
        else if (solver == "oropt" || solver == "or_opt") sol = or_opt::solve(ctx, problem, h, nullptr, init);
        else if (solver == "lk") sol = lin_kernighan::solve(ctx, problem, lk, nullptr, init, seed);
        else if (solver == "nn") sol = nearest_neighbor::solve(ctx, problem, h, nullptr, nullptr);
        else throw std::runtime_error("unknown solver `" + solver + "` (this build accelerates 2opt, 3opt, oropt, lk, nn)");
        if (!validate_tour(sol.route(), problem.cities)) throw std::runtime_error("pipeline: solver produced an invalid tour");  // pipeline.rs:70-71
        std::printf("%.5f %d\n", sol.total, 0);
        for (size_t k = 0; k < sol.route().size(); ++k) std::printf(k ? " %zu" : "%zu", sol.route()[k]);
        std::printf("\n");
        if (stats)
            std::fprintf(stderr, "sweeps=%llu candidates=%llu moves=%llu kernel_ms=%.3f total_ms=%.3f\n", (unsigned long long)sol.stats.sweeps,
                         (unsigned long long)sol.stats.candidates, (unsigned long long)sol.stats.moves, sol.stats.kernel_ms, sol.stats.total_ms);
        return 0;
    } catch (const ReferencePanic &e) {
        std::fprintf(stderr, "thread 'main' panicked: %s\n", e.what());
        return 101;  // Rust's panic exit code
    } catch (const std::exception &e) {
        std::fprintf(stderr, "Error: %s\n", e.what());
        return 1;
    }
}
