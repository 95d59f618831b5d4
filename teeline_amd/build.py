"""Builds libteeline_gpu.so (hipcc, gfx950 only) in-tree next to this file.

`python -m teeline_amd.build` or `teeline_amd.build.build()`; __graft_entry__.build() calls this.
hipcc cross-compiles without a GPU.  The .so is git-ignored but travels with the tree to the GPU box.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ROOT = os.path.dirname(HERE)
LIB = os.path.join(HERE, "libteeline_gpu.so")

SOURCES = ["tl_api.hip", "tl_api_two_opt.hip", "tl_api_scans.hip", "tl_api_lk.hip", "two_opt_ref.hip", "two_opt_nl.hip", "two_opt_dm.hip", "dm_build.hip", "three_opt.hip", "lk.hip", "lk_deep.hip", "nn_knn.hip", "two_opt_best.hip", "or_opt.hip", "two_opt_large.hip", "kdtree.hip"]
HEADERS = ["tl_device.h", "tl_kernels.h", "two_opt_common.h", "tl_api_common.h", os.path.join(ROOT, "include", "teeline_gpu.h")]

# -ffp-contract=off: the reference never fuses mul+add (src/tsp/kdtree.rs:291-295); bit-exact parity
# depends on it.  Correctly rounded sqrt/div is hipcc's default and is spelled out here on purpose.
FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
    "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-fast-math",
    "-fno-gpu-rdc",
    "-Wall", "-Wno-unused-function", "-Wno-pass-failed",  # (pass-failed: `#pragma unroll` hints the deep LK build cannot honour)
]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found — libteeline_gpu cannot be built (there is no CPU fallback)")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HEADERS]
    deps.append(os.path.abspath(__file__))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, extra_flags=(), out=None):
    """out: alternative output path (tuning variants built with extra -D flags); default = the product library.
    Every source is compiled to an object of its own, side by side (the LDS 2-opt kernel's instantiations alone take ~20 s), then linked."""
    if out is None and not force and not needs_build():
        return LIB
    import concurrent.futures as cf
    import tempfile
    LIB_OUT = out or LIB
    hipcc = _hipcc()
    cflags = [f for f in FLAGS if f != "-shared"] + list(extra_flags) + ["-I", os.path.join(ROOT, "include")]
    with tempfile.TemporaryDirectory(prefix="teeline_gpu_obj_") as tmp:
        def cc(src):
            obj = os.path.join(tmp, src.replace(".hip", ".o"))
            cmd = [hipcc] + cflags + ["-c", os.path.join(CSRC, src), "-o", obj]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            subprocess.check_call(cmd)
            return obj
        with cf.ThreadPoolExecutor(max(1, min(len(SOURCES), (os.cpu_count() or 2)))) as ex:
            objs = list(ex.map(cc, SOURCES))
        link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-fno-gpu-rdc", "-o", LIB_OUT + ".tmp"] + objs
        if verbose:
            print(" ".join(link), file=sys.stderr)
        subprocess.check_call(link)
    try:
        bad = verify_ds_min_waits(LIB_OUT + ".tmp")
    except (RuntimeError, OSError, subprocess.CalledProcessError) as exc:  # no disassembler on this box: the check is skipped, loudly
        if os.environ.get("TL_BUILD_STRICT"):  # CI: an unverified library is not installed
            raise RuntimeError(f"TL_BUILD_STRICT: the ds_min_u32 wait check could not run ({exc})") from exc
        print(f"teeline_amd.build: ds_min_u32 wait check skipped ({exc})", file=sys.stderr)
        bad = []
    if bad:
        raise RuntimeError("ds_min_u32 reaches an s_barrier without `s_waitcnt lgkmcnt(0)` in: " + ", ".join(f"{k} @ {a}" for k, a in bad[:8]))
    os.replace(LIB_OUT + ".tmp", LIB_OUT)
    return LIB_OUT


OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def verify_ds_min_waits(lib=None):
    """Build-time check of the linked library's gfx950 code: on every control-flow path from a `ds_min_u32` to the next
    `s_barrier` there is an `s_waitcnt` that drains lgkmcnt to 0.  The hit keys of the 2-opt kernels are posted with an inline-asm
    ds_min_u32 (csrc/tl_device.h lds_min_u32), which the compiler's wait insertion does not count; the barrier behind the post is
    only safe if a tracked LDS operation (or the explicit wait of lds_min_u32_fenced) sits in between.  Returns the list of
    offending (kernel, address) pairs — empty = good; raises if the disassembler is missing."""
    import re
    import tempfile
    lib = lib or LIB
    if not os.path.exists(OBJDUMP):
        raise RuntimeError(f"{OBJDUMP} not found: cannot check the ds_min_u32 -> s_barrier waits")
    bad = []
    with tempfile.TemporaryDirectory(prefix="teeline_gpu_asm_") as tmp:
        local = os.path.join(tmp, os.path.basename(lib))
        shutil.copy(lib, local)
        subprocess.check_call([OBJDUMP, "--offloading", local], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=tmp)
        for name in sorted(os.listdir(tmp)):
            if "gfx950" not in name:
                continue
            text = subprocess.check_output([OBJDUMP, "-d", os.path.join(tmp, name)], text=True)
            if "ds_min_u32" not in text:
                continue
            bad += _check_ds_min_paths(text, re)
    return bad


def _check_ds_min_paths(text, re):
    sym_re = re.compile(r"^([0-9a-f]+) <([^>]+)>:")
    ins_re = re.compile(r"^\s+(\S+)(.*?)//\s*([0-9A-Fa-f]+):")
    tgt_re = re.compile(r"<([^>+]+)\+0x([0-9a-f]+)>\s*$")
    syms, ins = {}, []  # symbol -> start address; (addr, mnemonic, operands, target address or None, kernel)
    cur = None
    for line in text.splitlines():
        m = sym_re.match(line)
        if m:
            cur = m.group(2)
            syms[cur] = int(m.group(1), 16)
            continue
        m = ins_re.match(line)
        if not m:
            continue
        tgt = None
        t = tgt_re.search(line)
        if t and m.group(1).startswith(("s_branch", "s_cbranch")):
            tgt = syms.get(t.group(1), None)
            tgt = None if tgt is None else tgt + int(t.group(2), 16)
        ins.append((int(m.group(3), 16), m.group(1), m.group(2), tgt, cur))
    at = {a: k for k, (a, *_rest) in enumerate(ins)}
    bad = []
    for k0, (a0, mn0, _o, _t, kern) in enumerate(ins):
        if mn0 != "ds_min_u32":
            continue
        seen, stack, ok = set(), [k0 + 1], True
        while stack and ok:
            k = stack.pop()
            while k < len(ins) and k not in seen:
                seen.add(k)
                _a, mn, ops, tgt, _k = ins[k]
                if mn == "s_waitcnt" and _waitcnt_drains_lgkm(ops):
                    break  # drained on this path
                if mn == "s_barrier":
                    ok = False
                    break
                if mn in ("s_endpgm",):
                    break
                if mn in ("s_setpc_b64", "s_swappc_b64"):
                    ok = False  # an indirect jump: not followed, so not proven
                    break
                if mn.startswith("s_cbranch"):
                    if tgt not in at:
                        ok = False  # a target this parser could not resolve: not proven
                        break
                    stack.append(at[tgt])
                elif mn == "s_branch":
                    if tgt in at:
                        stack.append(at[tgt])
                    else:
                        ok = False
                    break
                k += 1
        if not ok:
            bad.append((kern, hex(a0)))
    return bad


def _waitcnt_drains_lgkm(ops):
    """True if this s_waitcnt operand text waits for lgkmcnt == 0.  Symbolic form: `lgkmcnt(0)` by name.  A raw immediate (how the
    disassembler prints encodings it does not decompose): gfx9 SOPP simm16 carries lgkmcnt in bits [11:8] — vmcnt is [3:0]+[15:14],
    expcnt [6:4] — so `0x0f70` (lgkmcnt 15) is NOT a drain although its text starts with a zero."""
    t = ops.strip()
    if "lgkmcnt(" in t:
        return "lgkmcnt(0)" in t
    if "cnt(" in t:            # symbolic, other counters only: lgkmcnt is not waited on
        return False
    try:
        imm = int(t.split()[0].rstrip(","), 0)
    except (ValueError, IndexError):
        return False
    return ((imm >> 8) & 0xF) == 0


JITTER_LIB = os.path.join(HERE, "libteeline_gpu_jitter.so")


def build_jitter(force=False):
    """The race-stress build of the same sources (-DTL_JITTER: waves leave every workgroup barrier far apart, tl_device.h).
    Test infrastructure (tests/test_gpu_race_stress.py loads it in a child process); never the product library."""
    build()
    if not force and os.path.exists(JITTER_LIB) and os.path.getmtime(JITTER_LIB) >= os.path.getmtime(LIB):
        return JITTER_LIB
    return build(extra_flags=["-DTL_JITTER"], out=JITTER_LIB)


TUNE_LIB = os.path.join(HERE, "libteeline_gpu_tune.so")


def build_tune(force=False):
    """The tuning build of the same sources (-DTL_TUNE): also carries the kernel forms that were measured and rejected (the
    TL_TUNE_ONLY_FLAGS of include/teeline_gpu.h) and reads TL_* knobs from the environment.  Test infrastructure: the variant
    parity tests (tests/test_gpu_lk.py) run against it in a child process; never the product library."""
    build()
    if not force and os.path.exists(TUNE_LIB) and os.path.getmtime(TUNE_LIB) >= os.path.getmtime(LIB):
        return TUNE_LIB
    return build(extra_flags=["-DTL_TUNE"], out=TUNE_LIB)


def build_all(force=False):
    """Product library first, then the two test-infrastructure builds of the same sources side by side (each is a full hipcc run)."""
    import concurrent.futures as cf
    lib = build(force=force)
    with cf.ThreadPoolExecutor(2) as ex:
        j, t = ex.submit(build_jitter, force), ex.submit(build_tune, force)
        return lib, j.result(), t.result()


CLI = os.path.join(HERE, "teeline-gpu")
CLI_SRC = os.path.join(HERE, "host_cpp", "teeline_gpu_cli.cpp")
CLI_HDR = os.path.join(HERE, "host_cpp", "teeline_gpu.hpp")


def build_cli(force=False):
    """C++ host mirror + CLI façade (g++, links the in-tree libteeline_gpu.so through an $ORIGIN rpath)."""
    build()
    deps = [CLI_SRC, CLI_HDR, os.path.join(ROOT, "include", "teeline_gpu.h"), LIB]
    if not force and os.path.exists(CLI) and all(os.path.getmtime(d) <= os.path.getmtime(CLI) for d in deps):
        return CLI
    cmd = ["g++", "-O2", "-std=c++17", "-Wall", "-I", os.path.join(ROOT, "include"), CLI_SRC, "-o", CLI,
           "-L", HERE, "-lteeline_gpu", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)
    return CLI


if __name__ == "__main__":
    if "--tune" in sys.argv:  # tuning variant: rejected kernel forms + TL_* knobs from the environment (never the product library)
        print(build_tune(force=True))
        sys.exit(0)
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_cli(force="--force" in sys.argv))
    print(build_jitter(force="--force" in sys.argv))
    print(build_tune(force="--force" in sys.argv))
