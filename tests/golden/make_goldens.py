#!/usr/bin/env python3
"""Regenerates tests/golden/goldens.json from the CPU oracle (oracle/libtl_oracle.so).

The reference (Rust) cannot be built in this environment (no cargo/rustc), so the vectors are
produced by the oracle and PINNED to the reference by the published numbers listed in
tests/test_oracle_golden.py (REFERENCE_PUBLISHED): every cost below that the reference also
publishes must match to the printed precision, and the tiny cases must match the reference's own
unit-test expectations exactly.  Inputs are the reference's fixture data files copied verbatim to
tests/golden/tsplib/ (data, not source).

Usage: python tests/golden/make_goldens.py   (writes goldens.json next to this file)
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import numpy as np  # noqa: E402

import _oracle as O  # noqa: E402
import _tsplib as T  # noqa: E402


def f5(x):
    return f"{float(x):.5f}"


def main():
    out = {}
    for name in ("berlin52", "a280", "att532", "att48"):
        d = T.parse_tsplib(os.path.join(HERE, "tsplib", f"{name}.tsp"))
        xy, n, ids = d["xy"], d["n"], d["ids"]
        entry = {"n": n}
        rc, nn, cnn = O.nearest_neighbor(xy, None, n, 3)
        entry["nn"] = {"cost": f5(cnn), "route_pos": nn.tolist()}
        rc, p, c, st = O.two_opt(xy, None, n, init=nn)
        entry["nn_two_opt"] = {"cost": f5(c), "stats": st, "route_ids": ids[p].tolist()}
        rc, p, c, st = O.two_opt(xy, None, n)
        entry["identity_two_opt"] = {"cost": f5(c), "stats": st, "route_ids": ids[p].tolist()}
        if n <= 300:
            rc, p, c, st = O.or_opt(xy, None, n, init=nn)
            entry["nn_or_opt"] = {"cost": f5(c), "stats": st, "route_ids": ids[p].tolist()}
            rc, p, c, st = O.or_opt(xy, None, n)
            entry["identity_or_opt"] = {"cost": f5(c), "stats": st, "route_ids": ids[p].tolist()}
        if n <= 60:
            rc, p, c, st = O.three_opt(xy, None, n, init=nn)
            entry["nn_three_opt"] = {"cost": f5(c), "stats": st, "route_ids": ids[p].tolist()}
            rc, p, c, st = O.three_opt(xy, None, n)
            entry["identity_three_opt"] = {"cost": f5(c), "stats": st, "route_ids": ids[p].tolist()}
        out[name] = entry
    opt = T.parse_opt_tour(os.path.join(HERE, "tsplib", "berlin52.opt.tour"))
    b = T.parse_tsplib(os.path.join(HERE, "tsplib", "berlin52.tsp"))
    out["berlin52"]["opt_tour_cost"] = f5(O.tour_length(b["xy"], None, np.asarray(opt) - 1))
    for name in ("gr17", "ring6_explicit", "bays29"):
        d = T.parse_tsplib(os.path.join(HERE, "tsplib", f"{name}.tsp"))
        n, packed = d["n"], d["packed"]
        rc, p, c, st = O.two_opt(None, packed, n)
        e = {"n": n, "identity_two_opt": {"cost": f5(c), "stats": st, "route_pos": p.tolist()}}
        rc, p, c, st = O.three_opt(None, packed, n)
        e["identity_three_opt"] = {"cost": f5(c), "stats": st, "route_pos": p.tolist()}
        rc, p, c, st = O.or_opt(None, packed, n)
        e["identity_or_opt"] = {"cost": f5(c), "stats": st, "route_pos": p.tolist()}
        out[name] = e
    d = T.parse_tsplib(os.path.join(HERE, "tsplib", "burma14.tsp"))
    packed = O.dm_build_packed(d["xy"], geo=True)
    rc, p, c, st = O.two_opt(None, packed, d["n"])
    out["burma14_geo"] = {"n": d["n"], "packed_head": [f5(v) for v in packed[:10]],
                          "identity_two_opt": {"cost": f5(c), "stats": st, "route_pos": p.tolist()}}
    # synthetic instances (SURVEY.md §8(d) C2/C3 generator: xorshift64, default seed)
    for n in (1002, 10000):
        xy = O.synth_xy(n)
        rc, nn, cnn = O.nearest_neighbor(xy, None, n, 3)
        rc, p, c, st = O.two_opt(xy, None, n, init=nn)
        e = {"xy_head": [f5(v) for v in xy[:3].ravel()], "nn_cost": f5(cnn),
             "nn_two_opt": {"cost": f5(c), "stats": st, "route_crc": int(np.bitwise_xor.reduce(p * np.arange(1, n + 1, dtype=np.uint32)))}}
        rp = O.restart_perm(n, 12345, 0)
        rc, p, c, st = O.two_opt(xy, None, n, init=rp)
        e["restart0_seed12345_two_opt"] = {"cost": f5(c), "stats": st, "perm_head": rp[:8].tolist(),
                                          "route_crc": int(np.bitwise_xor.reduce(p * np.arange(1, n + 1, dtype=np.uint32)))}
        out[f"synthetic{n}"] = e
    with open(os.path.join(HERE, "goldens.json"), "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    print("wrote goldens.json")


if __name__ == "__main__":
    main()
