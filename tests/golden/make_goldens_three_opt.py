#!/usr/bin/env python3
"""Regenerates tests/golden/goldens_three_opt.json: FULL 3-opt descents (three_opt.rs:16-51 — repeated find_best_move + apply_3opt
until no triple improves) from the CPU oracle on instances where apply_3opt's segment-swap cases 4-7 (three_opt.rs:186-218) move
long segments many times: a280 from the NN seed (the reference's `thorough` preset order nn -> 3opt) and a synthetic n = 300
instance from a seeded random permutation.  Stored: f32 cost bits, cost to 5 decimals, moves / passes / triples, a CRC of the route.

Usage: python tests/golden/make_goldens_three_opt.py     (about a minute of oracle time)
"""
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import numpy as np  # noqa: E402

import _oracle as O  # noqa: E402
import _tsplib as T  # noqa: E402


def crc(p):
    p = np.asarray(p, dtype=np.uint32)
    return int(np.bitwise_xor.reduce(p * np.arange(1, len(p) + 1, dtype=np.uint32)))


def entry(rc, p, c, st):
    assert rc == 0 and O.validate_tour(p)
    return {"cost_bits": int(np.float32(c).view(np.uint32)), "cost": f"{float(c):.5f}", "stats": st, "route_crc": crc(p),
            "route_head": np.asarray(p[:12]).tolist()}


def main():
    out = {}
    d = T.parse_tsplib(os.path.join(HERE, "tsplib", "a280.tsp"))
    xy, n = d["xy"], d["n"]
    rc, nn, cnn = O.nearest_neighbor(xy, None, n, 3)
    t0 = time.time()
    out["a280_nn_three_opt"] = {"n": n, "nn_cost": f"{float(cnn):.5f}", **entry(*O.three_opt(xy, None, n, init=nn))}
    # the same descent over the packed matrix (DistanceMatrix lookups instead of coordinates): identical by construction of the
    # reference (distances are the same f32 values either way)
    packed = O.dm_build_packed(xy)
    out["a280_nn_three_opt_matrix"] = {"n": n, **entry(*O.three_opt(None, packed, n, init=nn))}
    n2 = 300
    xy2 = O.synth_xy(n2)
    rp = O.restart_perm(n2, 4, 0)
    out["synth300_perm_seed4_three_opt"] = {"n": n2, "xy_head": [f"{float(v):.5f}" for v in xy2[:3].ravel()], "perm_head": rp[:8].tolist(),
                                            **entry(*O.three_opt(xy2, None, n2, init=rp))}
    out["synth300_perm_seed4_three_opt_matrix"] = {"n": n2, **entry(*O.three_opt(None, O.dm_build_packed(xy2), n2, init=rp))}
    print(f"oracle time {time.time() - t0:.1f} s", file=sys.stderr)
    with open(os.path.join(HERE, "goldens_three_opt.json"), "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
        fh.write("\n")


if __name__ == "__main__":
    main()
