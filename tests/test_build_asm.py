"""The linked library's machine code, checked on the CPU (no GPU needed): every inline-asm `ds_min_u32` (csrc/tl_device.h) is
followed by an `s_waitcnt lgkmcnt(0)` on every path to the next `s_barrier` (the compiler's wait insertion does not
count LDS operations issued through inline asm).  teeline_amd/build.py runs the same check at link time."""
import os

import pytest

import teeline_amd.build as B

pytestmark = pytest.mark.skipif(not os.path.exists(B.OBJDUMP), reason="llvm-objdump not present")

_GOOD = """
0000000000001000 <k_good>:
	v_mov_b32_e32 v1, s0                                       // 000000001000: 7E020200
	ds_min_u32 v34, v35                                        // 000000001004: D80E0000 00002322
	s_cbranch_scc1 3                                           // 00000000100C: BF850003 <k_good+0x1c>
	ds_write_b32 v1, v2                                        // 000000001010: D81A0000 00000201
	s_waitcnt lgkmcnt(0)                                       // 000000001018: BF8CC07F
	s_waitcnt vmcnt(0) lgkmcnt(0)                              // 00000000101C: BF8C0070
	s_barrier                                                  // 000000001020: BF8A0000
	s_endpgm                                                   // 000000001024: BF810000
"""
_BAD = """
0000000000001000 <k_bad>:
	ds_min_u32 v34, v35                                        // 000000001000: D80E0000 00002322
	s_cbranch_scc1 2                                           // 000000001008: BF850002 <k_bad+0x14>
	s_waitcnt lgkmcnt(0)                                       // 00000000100C: BF8CC07F
	s_barrier                                                  // 000000001010: BF8A0000
	s_barrier                                                  // 000000001014: BF8A0000
	s_endpgm                                                   // 000000001018: BF810000
"""


def test_checker_follows_both_sides_of_a_branch():
    import re
    assert B._check_ds_min_paths(_GOOD, re) == []
    assert B._check_ds_min_paths(_BAD, re) == [("k_bad", "0x1000")]  # the taken side skips the wait


def test_a_raw_waitcnt_immediate_is_decoded_not_prefix_matched():
    # `s_waitcnt 0x0f70` (lgkmcnt field [11:8] = 15) starts with a zero but drains nothing
    import re
    assert B._waitcnt_drains_lgkm("lgkmcnt(0)") and B._waitcnt_drains_lgkm("vmcnt(0) lgkmcnt(0)")
    assert not B._waitcnt_drains_lgkm("vmcnt(0)") and not B._waitcnt_drains_lgkm("lgkmcnt(1)")
    assert B._waitcnt_drains_lgkm("0") and B._waitcnt_drains_lgkm("0x0070") and B._waitcnt_drains_lgkm("0xc07f")
    assert not B._waitcnt_drains_lgkm("0x0f70") and not B._waitcnt_drains_lgkm("0x0170") and not B._waitcnt_drains_lgkm("garbage")
    raw_bad = _GOOD.replace("s_waitcnt lgkmcnt(0)     ", "s_waitcnt 0x0f70         ").replace("s_waitcnt vmcnt(0) lgkmcnt(0)", "s_waitcnt vmcnt(0)           ")
    assert B._check_ds_min_paths(raw_bad, re) == [("k_good", "0x1004")]
    raw_ok = _GOOD.replace("s_waitcnt lgkmcnt(0)     ", "s_waitcnt 0xc07f         ")
    assert B._check_ds_min_paths(raw_ok, re) == []


@pytest.mark.parametrize("lib", ["LIB", "JITTER_LIB", "TUNE_LIB"])
def test_every_ds_min_is_drained_before_the_next_barrier(lib):
    path = getattr(B, lib)
    if not os.path.exists(path):
        pytest.skip(f"{os.path.basename(path)} not built")
    assert B.verify_ds_min_waits(path) == []
