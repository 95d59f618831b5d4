#!/usr/bin/env python3
"""Regenerates tests/golden/goldens_large.json: full-size vectors of BASELINE configs[3] and [4] from the CPU oracle.

  configs[3]  synthetic EUC_2D n = 10 000, seeded random restarts 0..7 (seed 12345) of the REF_ORDER 2-opt descent:
              cost, CRC-32 of the final tour (u32 little-endian positions), sweeps / candidates / moves / reversed.
  configs[4]  synthetic n = 13 509 (usa13509 is not in the reference tree, SURVEY.md finding 5): k-NN lists k = 5
              (brute force and through the restated kd-tree: the same), NN seed, nn -> 2-opt, and Lin-Kernighan with
              n_nearest = 5, max_depth = 5, epochs = 2, kick seed 7 from the NN seed.

The oracle needs minutes for these (the LK run ~1 min on one core), so the CPU suite only checks the file's shape and the
cheap entries; the `-m gpu` tests compare the HIP path with the committed values.

Usage: python tests/golden/make_goldens_large.py
"""
import json
import os
import sys
import time
import zlib

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import numpy as np  # noqa: E402

import _oracle as O  # noqa: E402


def f5(x):
    return f"{float(x):.5f}"


def crc(a):
    return int(zlib.crc32(np.ascontiguousarray(a, dtype="<u4").tobytes()))


def main():
    out = {}
    t0 = time.time()
    n = 10000
    xy = O.synth_xy(n)
    batch = {}
    for r in range(8):
        rp = O.restart_perm(n, 12345, r)
        rc, p, c, st = O.two_opt(xy, None, n, init=rp)
        assert rc == 0
        batch[str(r)] = {"cost": f5(c), "route_crc32": crc(p), "stats": st, "init_crc32": crc(rp)}
        print(f"n={n} restart {r}: {f5(c)} {st} ({time.time() - t0:.0f} s)", flush=True)
    out["synthetic10000_seed12345"] = {"n": n, "restarts": batch}

    n = 13509
    xy = O.synth_xy(n)
    e = {"n": n, "xy_crc32": int(zlib.crc32(xy.tobytes()))}
    bf = O.build_candidates(xy, 5)
    kd, tie_free = O.build_candidates_kdtree(xy, 5)
    assert np.array_equal(bf, kd)
    e["knn_k5"] = {"crc32": crc(bf), "kdtree_equal": True, "kdtree_tie_free": tie_free, "head": bf[:3].tolist()}
    rc, nn, cnn = O.nearest_neighbor(xy, None, n, 3)
    e["nn"] = {"cost": f5(cnn), "route_crc32": crc(nn)}
    rc, p, c, st = O.two_opt(xy, None, n, init=nn)
    e["nn_two_opt"] = {"cost": f5(c), "route_crc32": crc(p), "stats": st}
    print(f"n={n} nn->2opt: {f5(c)} {st} ({time.time() - t0:.0f} s)", flush=True)
    rc, p, c, st = O.lin_kernighan(xy, init=nn, epochs=2, platoo_epochs=10, n_nearest=5, max_depth=5, seed=7, cand=kd)
    assert rc == 0
    e["lk_epochs2_seed7_from_nn"] = {"cost": f5(c), "route_crc32": crc(p), "stats": st,
                                     "opts": {"epochs": 2, "platoo_epochs": 10, "n_nearest": 5, "max_depth": 5, "seed": 7}}
    print(f"n={n} LK: {f5(c)} {st} ({time.time() - t0:.0f} s)", flush=True)
    out["synthetic13509"] = e
    with open(os.path.join(HERE, "goldens_large.json"), "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    print("wrote goldens_large.json")


if __name__ == "__main__":
    main()
