"""pipeline::run_pipeline_stages — mirror of src/tsp/pipeline.rs:53-80 for the GPU-accelerated solvers.

Stage k+1 is warm-started with stage k's tour; a seed that fails validate_tour is dropped with a warning (the stage then
uses its default seeding, :60-65); an invalid stage RESULT is a hard error (:70-71).  Each outcome carries the stage name,
its Solution and the wall time in ms (StageOutcome, :11-14).
"""
import time
import warnings

# Solvers::from_str (mod.rs:559-590), restricted to what this build accelerates
SOLVER_NAMES = {"nn": "nearest_neighbor", "nearest_neighbor": "nearest_neighbor", "2opt": "two_opt", "two_opt": "two_opt",
                "3opt": "three_opt", "three_opt": "three_opt", "oropt": "or_opt", "or_opt": "or_opt", "or-opt": "or_opt",
                "lk": "lin_kernighan", "lin_kernighan": "lin_kernighan", "shuffle": "random_shuffle",
                "random_shuffle": "random_shuffle"}
PRESETS = {"fast": ["nn", "2opt"]}  # resolve_preset (main.rs:354-369); classic / thorough end in SA (not accelerated)
AUTO_EXPAND_WITH_NN = {"two_opt", "three_opt", "or_opt", "lin_kernighan"}  # mod.rs:129-139


def steps_for_solve(solver, no_seed=False):
    """`teeline solve <solver>` -> pipeline steps (main.rs:371-398)."""
    s = solver.lower()
    if s in PRESETS:
        return list(PRESETS[s])
    if s not in SOLVER_NAMES:
        raise ValueError(f"unknown solver `{solver}` (this build accelerates {sorted(set(SOLVER_NAMES))})")
    if not no_seed and SOLVER_NAMES[s] in AUTO_EXPAND_WITH_NN:
        return ["nn", s]
    return [s]


def random_shuffle(problem, seed=1, *, ctx=None):
    """random_shuffle::solve (random_shuffle.rs:12-27), seeded: the Fisher-Yates stream of restart 0 of `seed` (synth.restart_perm,
    the device's multi-start generator); the reference draws from an unseeded thread RNG."""
    from . import Solution, synth
    import ctypes as C
    import numpy as np
    from . import default_context
    ctx = ctx or default_context()
    n = len(problem)
    perm = synth.restart_perm(n, seed, 0)
    cost = C.c_float(0.0)
    if n >= 2:
        packed = problem.explicit_packed()
        ctx.check(ctx.lib.tl_tour_length(ctx.handle, None if packed is not None else problem.xy.ctypes.data_as(C.c_void_p),
                                         None if packed is None else packed.ctypes.data_as(C.c_void_p), n,
                                         perm.ctypes.data_as(C.c_void_p), C.byref(cost)))
    return Solution(cost.value, problem.ids[perm], problem, {})


def format_solution(sol, is_optimized=False):
    """print_solution (main.rs:645-652): every id is followed by one space."""
    return f"{float(sol.total):.5f} {1 if is_optimized else 0}\n" + "".join(f"{v} " for v in sol.route()) + "\n"


def _json_f32(v):
    import numpy as np
    return repr(float(np.float32(v)))  # f32 widened to f64, shortest round-trip decimal, "60.0" for integers (serde_json)


def format_solution_json(sol, is_optimized=False, comparison=None):
    """print_solution_json (main.rs:700-712): keys sorted (serde_json's BTreeMap), compact."""
    s = '{"cost":' + _json_f32(sol.total)
    if comparison is not None:
        s += ',"gap_pct":' + _json_f32(comparison[1]) + ',"optimal_cost":' + _json_f32(comparison[0])
    s += ',"optimized":' + ("true" if is_optimized else "false") + ',"route":[' + ",".join(str(v) for v in sol.route()) + "]}\n"
    return s


class StageOutcome:
    def __init__(self, name, solution, duration_ms):
        self.name, self.solution, self.duration_ms = name, solution, duration_ms


def run_pipeline_stages(problem, steps, opts=None, *, ctx=None, lk_seed=1):
    """steps: iterable of solver names ("nn", "2opt", "3opt", "or_opt", "lk", "shuffle"); opts: {name: options} (optional)."""
    from . import (HeuristicOptions, LKOptions, lin_kernighan, nearest_neighbor, or_opt, three_opt, two_opt,
                   validate_tour)
    mods = {"nearest_neighbor": nearest_neighbor, "two_opt": two_opt, "three_opt": three_opt, "or_opt": or_opt,
            "lin_kernighan": lin_kernighan}
    opts = opts or {}
    outcomes, seed = [], None
    for step in steps:
        if step not in SOLVER_NAMES:
            raise ValueError(f"unknown solver `{step}` (this build accelerates {sorted(set(SOLVER_NAMES))})")
        name = SOLVER_NAMES[step]
        init = seed
        if init is not None and not validate_tour(init, problem):
            warnings.warn(f"pipeline: seed for stage `{step}` is not a valid tour; falling back to default seeding")
            init = None
        t0 = time.perf_counter()
        if name == "random_shuffle":
            sol = random_shuffle(problem, lk_seed, ctx=ctx)
        elif name == "lin_kernighan":
            sol = lin_kernighan.solve(problem, opts.get(step) or LKOptions(), None, init, ctx=ctx, seed=lk_seed)
        else:
            sol = mods[name].solve(problem, opts.get(step) or HeuristicOptions(), None, init, ctx=ctx)
        ms = (time.perf_counter() - t0) * 1e3
        if not validate_tour(sol.route(), problem):
            raise RuntimeError(f"pipeline: stage `{step}` produced an invalid tour")
        outcomes.append(StageOutcome(step, sol, ms))
        seed = sol.route()
    return outcomes
