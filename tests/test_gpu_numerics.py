"""GPU numerics contract (-m gpu): the correctly rounded f32 sqrt every kernel uses (= Rust's f32::sqrt, kdtree.rs:294)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_sqrt_rn_exhaustive_over_all_non_negative_floats(ctx):
    # all 2^31 bit patterns 0x00000000 .. 0x7FFFFFFF (zero, denormals, normals, +inf, NaNs): fast path == compiler expansion
    bad, first = C.c_uint64(), C.c_uint32()
    ctx.check(ctx.lib.tl_selftest_sqrt(ctx.handle, 0, 1 << 31, C.byref(bad), C.byref(first)))
    assert bad.value == 0, f"{bad.value} mismatches, first at bits {first.value:#010x}"


def test_sqrt_rn_matches_the_host(ctx):
    # and the compiler expansion is the IEEE result: spot-check the device against numpy's sqrt via the matrix build
    import teeline_amd as TA
    rng = np.random.default_rng(3)
    for scale in (1e-18, 1e-6, 1.0, 1e9, 1e18):
        xy = (rng.random((2000, 2)) * scale).astype(np.float32)
        dm = TA.distance_matrix.build(np.arange(2000), xy, ctx=ctx)
        i, j = np.tril_indices(2000, -1)
        dx, dy = xy[i, 0] - xy[j, 0], xy[i, 1] - xy[j, 1]
        ref = np.sqrt((dx * dx + dy * dy).astype(np.float32)).astype(np.float32)
        assert np.array_equal(dm.items.view(np.uint32), ref.view(np.uint32))
