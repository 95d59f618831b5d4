"""Synthetic inputs of the benchmark configurations (this build's own specification).

C3/C4 points: xorshift64 (s ^= s<<13; s ^= s>>7; s ^= s<<17), seed 88172645463325252, x then y per city,
coord = (u % 1_000_000) / 1000.0f  -> uniform on [0, 1000)^2 with three decimals.
Restart r of seed S starts from the Fisher–Yates permutation `for i in (1..n).rev(): j = rng % (i+1); swap`
driven by splitmix64 seeded with S + r (generated on the device by the descent kernel itself).
"""
import numpy as np

DEFAULT_XY_SEED = 88172645463325252
_M = (1 << 64) - 1


def synth_xy(n, seed=DEFAULT_XY_SEED):
    s = seed or DEFAULT_XY_SEED
    out = np.empty(2 * n, dtype=np.float32)
    for i in range(2 * n):
        s ^= (s << 13) & _M
        s ^= s >> 7
        s ^= (s << 17) & _M
        out[i] = np.float32(s % 1_000_000) / np.float32(1000.0)
    return out.reshape(n, 2)


def restart_perm(n, seed, r):
    st = (seed + r) & _M
    perm = np.arange(n, dtype=np.uint32)
    for i in range(n - 1, 0, -1):
        st = (st + 0x9E3779B97F4A7C15) & _M
        z = st
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M
        z ^= z >> 31
        j = z % (i + 1)
        perm[i], perm[j] = perm[j], perm[i]
    return perm
