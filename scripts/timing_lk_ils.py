"""LK on the instances the reference publishes wall times for (bench/baseline-solvers.tsv:17-31: berlin52 0.10-0.22 s, a280 1.1-3.1 s,
att532 14-37 s) with the CLI's defaults (mod.rs:596-613,1321-1325: epochs 10000, platoo 500, n_nearest 3, depth 5), and synthetic
sizes with the library defaults: the LDS-resident single-workgroup ILS (k_lk_ils, default up to n = 2000; TL_FLAG_LK_ILS_LDS beyond)
against the chip-wide scans (TL_FLAG_LK_CHIP_WIDE) — same tours, same counters (asserted); with `oracle` as argv[1] also against
the oracle on one core (small cases).  GPU box:  python scripts/timing_lk_ils.py [oracle]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, teeline_amd as TA
import _tsplib as T

with_oracle = "oracle" in sys.argv
if with_oracle:
    import _oracle as O


def run(xy, opts, seed, flags):
    print(f"  ... n={len(xy)} flags={flags:#x}", file=sys.stderr, flush=True)
    with TA.Context(0, flags) as ctx:
        p = TA.TspProblem(np.arange(len(xy)), xy)
        best = (1e9, 1e9)
        for _ in range(2):
            t = time.perf_counter()
            sol = TA.lin_kernighan.solve(p, opts, None, None, ctx=ctx, seed=seed)
            w = (time.perf_counter() - t) * 1e3
            best = min(best, (sol.stats["kernel_ms"], w))
    return best, sol


cases = []
for name in ("berlin52", "a280", "att532"):
    xy = T.parse_tsplib(os.path.join(ROOT, "tests", "golden", "tsplib", f"{name}.tsp"))["xy"]
    cases.append((name, xy, dict(epochs=10000, platoo_epochs=500, n_nearest=3), 5))
for n in ((100, 400, 1000, 2000, 3000) if "short" not in sys.argv else ()):
    cases.append((f"synth{n}", TA.synth.synth_xy(n), dict(epochs=100, platoo_epochs=10, n_nearest=5), 5))
for name, xy, h, depth in cases:
    opts = TA.LKOptions(TA.HeuristicOptions(**h), depth)
    (k_ils, w_ils), s_ils = run(xy, opts, 1, TA.TL_FLAG_LK_ILS_LDS)
    (k_seq, w_seq), s_seq = run(xy, opts, 1, TA.TL_FLAG_LK_ILS_LDS | TA.TL_FLAG_LK_NO_SPECULATION)
    assert list(s_seq.route()) == list(s_ils.route()) and all(s_ils.stats[q] == s_seq.stats[q] for q in ("sweeps", "candidates", "moves", "reversed")), (name, s_ils.stats, s_seq.stats)
    (k_chip, w_chip), s_chip = run(xy, opts, 1, TA.TL_FLAG_LK_CHIP_WIDE)
    same = list(s_ils.route()) == list(s_chip.route()) and np.float32(s_ils.total).tobytes() == np.float32(s_chip.total).tobytes() and \
        all(s_ils.stats[q] == s_chip.stats[q] for q in ("sweeps", "candidates", "moves", "reversed"))
    rounds = s_ils.stats["sweeps"]
    line = (f"{name:10s} n={len(xy):5d} rounds {rounds:7d} moves {s_ils.stats['moves']:6d} cost {float(s_ils.total):.5f} | ils kernel {k_ils:9.2f} ms wall {w_ils:9.2f} "
            f"({k_ils * 1e3 / max(rounds, 1):6.2f} us/round) | sequential epochs {k_seq:9.2f} ms | chip-wide kernel {k_chip:9.2f} ms wall {w_chip:9.2f} ({k_chip * 1e3 / max(rounds, 1):6.2f} us/round) | same {same}")
    if with_oracle and len(xy) <= 600:
        t = time.perf_counter()
        rc, oroute, ocost, ost, _ = O.lin_kernighan_trace(xy, epochs=h["epochs"], platoo_epochs=h["platoo_epochs"], n_nearest=h["n_nearest"], max_depth=depth, seed=1, cap=2048)
        wo = (time.perf_counter() - t) * 1e3
        eq = list(s_ils.route()) == oroute.tolist() and np.float32(s_ils.total).tobytes() == np.float32(ocost).tobytes() and s_ils.stats["sweeps"] == ost["sweeps"]
        line += f" | oracle 1 core {wo:9.1f} ms, equal {eq}"
    print(line, flush=True)
    assert same, name
