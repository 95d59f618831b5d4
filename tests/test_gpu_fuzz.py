"""GPU fuzz / adversarial-numerics parity (-m gpu): REF_ORDER 2-opt must stay bit-identical to the oracle when the
approximate stages of the decision cascade (hardware sqrt, margins, bounding boxes) are stressed: huge and tiny
coordinate scales (overflow to inf, denormal squares), collinear and clustered points, exact duplicates, near-ties."""
import numpy as np
import pytest

import _oracle as O

pytestmark = pytest.mark.gpu


def run_both(ctx, xy, init=None):
    import teeline_amd as TA
    n = len(xy)
    sol = TA.two_opt.solve(TA.TspProblem(np.arange(n), xy), None, None, None if init is None else [int(v) for v in init], ctx=ctx)
    rc, route, cost, st = O.two_opt(xy, None, n, init=init)
    assert rc == 0
    assert list(sol.route()) == route.tolist(), "tour differs from the oracle"
    a, b = np.float32(sol.total), np.float32(cost)
    assert a.tobytes() == b.tobytes() or (np.isnan(a) and np.isnan(b))
    assert (sol.stats["sweeps"], sol.stats["moves"]) == (st["sweeps"], st["moves"])


@pytest.mark.parametrize("scale", [1e-25, 1e-19, 1e-12, 1e-3, 1.0, 1e6, 1e15, 1e18])
def test_coordinate_scales(ctx, scale):
    # squares underflow to denormals / zero at the small end and approach f32 max at the large end
    rng = np.random.default_rng(int(abs(np.log10(scale))) + 1)
    xy = (rng.random((400, 2)) * 1000.0 * scale).astype(np.float32)
    run_both(ctx, xy)
    run_both(ctx, xy, O.restart_perm(400, 5, 1))


def test_overflowing_distances(ctx):
    # dx*dx overflows to +inf: every comparison degenerates exactly like the reference's f32 arithmetic does
    rng = np.random.default_rng(9)
    xy = (rng.random((200, 2)) * 3.0e19).astype(np.float32)
    run_both(ctx, xy)


@pytest.mark.parametrize("kind", ["collinear", "clusters", "duplicates", "two_values", "circle", "near_ties"])
def test_degenerate_geometry(ctx, kind):
    rng = np.random.default_rng(abs(hash(kind)) % 1000)
    n = 700
    if kind == "collinear":
        t = rng.random(n).astype(np.float32) * 1000
        xy = np.stack([t, np.float32(0.5) * t + np.float32(3)], 1)
    elif kind == "clusters":
        c = rng.random((7, 2)) * 1000
        xy = c[rng.integers(0, 7, n)] + rng.normal(0, 0.01, (n, 2))
    elif kind == "duplicates":
        base = (rng.random((40, 2)) * 100).astype(np.float32)
        xy = base[rng.integers(0, 40, n)]
    elif kind == "two_values":
        xy = rng.integers(0, 2, (n, 2)).astype(np.float32)
    elif kind == "circle":
        a = rng.random(n) * 2 * np.pi
        xy = np.stack([np.cos(a), np.sin(a)], 1) * 500 + 500
    else:  # integer lattice jittered by one ulp: |new - cur| at the rounding limit
        g = rng.integers(0, 30, (n, 2)).astype(np.float32)
        xy = np.nextafter(g, g + rng.choice([-1, 1], (n, 2))).astype(np.float32)
    xy = np.ascontiguousarray(xy, dtype=np.float32)
    run_both(ctx, xy)
    run_both(ctx, xy, O.restart_perm(n, 77, 3))


@pytest.mark.parametrize("seed", range(6))
def test_random_small_instances(ctx, seed):
    rng = np.random.default_rng(100 + seed)
    n = int(rng.integers(4, 900))
    mode = seed % 3
    if mode == 0:
        xy = rng.random((n, 2)) * 1000
    elif mode == 1:
        xy = rng.integers(0, 12, (n, 2))           # heavy ties
    else:
        xy = rng.normal(0, 1, (n, 2)) * 10.0 ** rng.integers(-6, 7, (n, 1))  # mixed magnitudes
    xy = np.ascontiguousarray(xy, dtype=np.float32)
    run_both(ctx, xy, O.restart_perm(n, seed, 0))
