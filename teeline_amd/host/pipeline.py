"""pipeline::run_pipeline_stages — mirror of src/tsp/pipeline.rs:53-80 for the GPU-accelerated solvers.

Stage k+1 is warm-started with stage k's tour; a seed that fails validate_tour is dropped with a warning (the stage then
uses its default seeding, :60-65); an invalid stage RESULT is a hard error (:70-71).  Each outcome carries the stage name,
its Solution and the wall time in ms (StageOutcome, :11-14).
"""
import time
import warnings

SOLVER_NAMES = {"nn": "nearest_neighbor", "2opt": "two_opt", "3opt": "three_opt", "oropt": "or_opt", "or_opt": "or_opt",
                "lk": "lin_kernighan"}


class StageOutcome:
    def __init__(self, name, solution, duration_ms):
        self.name, self.solution, self.duration_ms = name, solution, duration_ms


def run_pipeline_stages(problem, steps, opts=None, *, ctx=None, lk_seed=1):
    """steps: iterable of solver names ("nn", "2opt", "3opt", "oropt", "lk"); opts: {name: options} (optional)."""
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
        if name == "lin_kernighan":
            sol = lin_kernighan.solve(problem, opts.get(step) or LKOptions(), None, init, ctx=ctx, seed=lk_seed)
        else:
            sol = mods[name].solve(problem, opts.get(step) or HeuristicOptions(), None, init, ctx=ctx)
        ms = (time.perf_counter() - t0) * 1e3
        if not validate_tour(sol.route(), problem):
            raise RuntimeError(f"pipeline: stage `{step}` produced an invalid tour")
        outcomes.append(StageOutcome(step, sol, ms))
        seed = sol.route()
    return outcomes
