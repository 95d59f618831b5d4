"""Race stress (-m gpu): the randomized parity campaigns against the oracle, run on libteeline_gpu_jitter.so.

That library is the product's sources built with -DTL_JITTER (teeline_amd/build.py build_jitter, tl_device.h TL_SYNC):
after every workgroup barrier pseudo-randomly chosen waves sleep for 3-30 us, so the waves of a workgroup run far out of
step.  Results must not depend on that — every exchange through LDS has to be ordered by a barrier or be exact on its
own.  (The build found one that was not: the LDS 2-opt kernel skipped round 2 of a dense row whenever *some* key was
already posted, which a late wave could read as "nothing left to do" while holding the row's real first hit.)
Each campaign runs in a child process, because a process binds one library.
"""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JITTER_LIB = os.path.join(ROOT, "teeline_amd", "libteeline_gpu_jitter.so")


def test_the_stress_library_is_the_jitter_build():
    import ctypes as C
    assert os.path.exists(JITTER_LIB), "built by __graft_entry__.build() / python -m teeline_amd.build"
    lib = C.CDLL(JITTER_LIB)
    lib.tl_version.restype = C.c_char_p
    assert lib.tl_version().decode().endswith("+jitter")


@pytest.mark.parametrize("script,seconds,min_runs", [
    ("fuzz_campaign.py", 15, 8),          # LDS 2-opt (pruned, NO_PRUNE, 8- and 4-wave forms) and the matrix form
    ("fuzz_campaign_lk.py", 12, 6),       # LK, kd-tree candidate lists, NN seed
    ("fuzz_campaign_oropt.py", 8, 4),     # Or-opt and 3-opt scans / solves
])
def test_results_do_not_depend_on_wave_timing(script, seconds, min_runs):
    env = dict(os.environ, TEELINE_GPU_LIB=JITTER_LIB)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "probes", script), str(seconds)], env=env,
                       capture_output=True, text=True, timeout=seconds * 6 + 300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    m = re.search(r"(\d+) runs, (\d+) mismatches", r.stdout)
    assert m, r.stdout[-2000:]
    assert int(m.group(2)) == 0, r.stdout[-4000:]
    assert int(m.group(1)) >= min_runs, f"only {m.group(1)} runs in {seconds} s"
