#!/usr/bin/env python3
"""Regenerates tests/golden/goldens_limits.json: oracle results of the size-limit cases whose oracle run is too slow for the GPU
suite's time budget (keep `pytest -m gpu` under 300 s).

  lattice257_two_opt   the 257 x 257 lattice (n = 66 049, spacing 3: every distance exact in f32) walked as a snake with five
                       segments reversed, REF_ORDER 2-opt (tests/test_gpu_limits.py::test_two_opt_beyond_65535_cities): cost bits,
                       CRC-32 of the final tour (u32 little-endian positions) and of the initial tour, sweeps / candidates / moves / reversed.

The test builds the same instance and initial tour (checked against init_crc32) and compares the HIP result with these values.
Usage: python tests/golden/make_goldens_limits.py      (~30 s of one core)
"""
import json
import os
import sys
import zlib

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import numpy as np  # noqa: E402

import _oracle as O  # noqa: E402


def crc(a):
    return int(zlib.crc32(np.ascontiguousarray(a, dtype="<u4").tobytes()))


def lattice257():
    m = 257
    gx, gy = np.meshgrid(np.arange(m, dtype=np.float32), np.arange(m, dtype=np.float32))
    xy = np.ascontiguousarray(np.stack([gx.ravel() * 3.0, gy.ravel() * 3.0], 1), dtype=np.float32)
    snake = np.concatenate([(r * m + (np.arange(m) if r % 2 == 0 else np.arange(m)[::-1])) for r in range(m)]).astype(np.uint32)
    init = snake.copy()
    for a, b in ((40, 90), (300, 1500), (65540, 65600), (65700, 65990), (66000, 66040)):
        init[a:b + 1] = init[a:b + 1][::-1].copy()
    return xy, init


def main():
    xy, init = lattice257()
    n = len(xy)
    rc, route, cost, st = O.two_opt(xy, None, n, init=init)
    assert rc == 0 and st["moves"] >= 5
    out = {"lattice257_two_opt": {"n": n, "init_crc32": crc(init), "route_crc32": crc(route), "cost_bits": int(np.float32(cost).view(np.uint32)),
                                  "cost": f"{float(cost):.5f}", "stats": st}}
    with open(os.path.join(HERE, "goldens_limits.json"), "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    print(out)


if __name__ == "__main__":
    main()
