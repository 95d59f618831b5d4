#!/usr/bin/env python3
"""Static instruction census of one kernel instantiation, by source region.

Compiles a HIP source for gfx950 with line tables (-gline-tables-only -S), cuts the asm of ONE kernel symbol out of it and counts its
instructions by class (SALU / VALU / LDS / VMEM / SMEM / s_waitcnt / s_nop / s_barrier / branches) per source region: the lexically
enclosing function of the `.loc` line each instruction is attributed to (inlined code counts where it was written), and — inside the
kernel body itself — the blocks between marker comments.  Static counts: an instruction inside a loop counts once.

usage: python scripts/isa_census.py two_opt_ref.hip '<mangled kernel symbol substring>' [out.txt] [extra -D flags...]
"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "teeline_amd", "csrc")

# marker comments that cut the kernel body of two_opt_ref.hip into blocks (first match of each, in file order)
BODY_MARKS = [
    ("initial tour + tile boxes", "// ---------------------------------------------------------------- initial tour"),
    ("descent set-up", "// ---------------------------------------------------------------- descent"),
    ("control wave", "// ------------------------------------------------------------ control wave"),
    ("worker: descriptor (B0)", "// ------------------------------------------------------------ worker waves"),
    ("worker: dense step", "// ---- dense step"),
    ("worker: pruned step", "// ---- pruned step"),
    ("worker: step end (B2 + boundary call)", "rows_done:;"),
    ("results + ordered cost sum", "// ---------------------------------------------------------------- results"),
]


def functions_of(path):
    """[(first_line, last_line, name)] of the function definitions of a source file (brace matching from a definition's `{`)."""
    src = open(path).read().split("\n")
    out, depth, cur = [], 0, None
    head = re.compile(r"^\s*(?:template\s*<[^>]*>\s*)?(?:static\s+|inline\s+|constexpr\s+|__host__\s+|__device__\s+|__global__\s+|__forceinline__\s+|__launch_bounds__\([^)]*\)\s+|__attribute__\(\(.*?\)\)\s+)*[\w:<>\*&\s,]+?\b(\w+)\s*\(")
    pend = None
    for ln, text in enumerate(src, 1):
        s = text.split("//")[0]
        if cur is None and depth_ns_only(depth_stack := None) is None:
            pass
        if cur is None:
            m = head.match(s)
            if m and not s.strip().startswith(("namespace", "#", "return", "if", "for", "while", "}", "using", "typedef")) and "=" not in s.split("(")[0]:
                pend = (ln, m.group(1))
            if pend and "{" in s and s.rstrip().endswith("{") is False and ";" in s:
                pass
            if pend and "{" in s:
                cur = [pend[0], None, pend[1]]
                depth = s.count("{") - s.count("}")
                pend = None
                if depth <= 0:
                    cur[1] = ln
                    out.append(tuple(cur))
                    cur = None
                continue
            if pend and ";" in s and "{" not in s:
                pend = None
        else:
            depth += s.count("{") - s.count("}")
            if depth <= 0:
                cur[1] = ln
                out.append(tuple(cur))
                cur = None
    return out


def depth_ns_only(_):
    return None


def classify(mn):
    if mn == "s_waitcnt" or mn.startswith("s_waitcnt"):
        return "waitcnt"
    if mn == "s_barrier":
        return "barrier"
    if mn in ("s_nop", "s_sleep"):
        return "s_nop"
    if mn.startswith(("s_cbranch", "s_branch", "s_setpc", "s_swappc", "s_endpgm")):
        return "branch"
    if mn.startswith(("s_load", "s_buffer_load", "s_memtime", "s_memrealtime", "s_dcache")):
        return "SMEM"
    if mn.startswith("s_"):
        return "SALU"
    if mn.startswith("ds_"):
        return "LDS"
    if mn.startswith(("global_", "flat_", "buffer_", "scratch_")):
        return "VMEM"
    if mn.startswith("v_"):
        return "VALU"
    return "other"


def main():
    src, sym = sys.argv[1], sys.argv[2]
    out = sys.argv[3] if len(sys.argv) > 3 and not sys.argv[3].startswith("-") else None
    extra = [a for a in sys.argv[3:] if a.startswith("-")]
    asm = f"/tmp/{os.path.basename(src)}.census.s"
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt",
           "-fno-fast-math", "-I", os.path.join(ROOT, "include"), "--cuda-device-only", "-gline-tables-only", "-S", "-o", asm, os.path.join(CSRC, src)] + extra
    subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
    lines = open(asm).read().split("\n")
    files = {}
    for l in lines:
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"\s+"([^"]*)"', l)
        if m:
            files[int(m.group(1))] = os.path.join(m.group(2), m.group(3))
        else:
            m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"', l)
            if m:
                files[int(m.group(1))] = m.group(2)
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*:", l) and sym in l)
    name = lines[start].split(":")[0]
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    # the last s_endpgm of the function: go on to .Lfunc_end
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    fn_tables = {}
    body = None
    main_src = os.path.join(CSRC, src)
    text = open(main_src).read().split("\n")
    marks = []
    for label, needle in BODY_MARKS:
        for ln, t in enumerate(text, 1):
            if needle in t and (not marks or ln > marks[-1][0]):
                marks.append((ln, label))
                break
    census = collections.defaultdict(collections.Counter)
    cur = ("?", 0)
    for l in lines[start:end]:
        m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
        if m:
            cur = (files.get(int(m.group(1)), "?"), int(m.group(2)))
            continue
        s = l.strip()
        if not s or s.startswith((".", ";", "//")) or s.endswith(":"):
            continue
        mn = s.split()[0]
        if not re.match(r"^[a-z_0-9]+$", mn):
            continue
        f, ln = cur
        region = os.path.basename(f) + ":?"
        if os.path.exists(f):
            if f not in fn_tables:
                fn_tables[f] = functions_of(f)
            hit = [fn for fn in fn_tables[f] if fn[0] <= ln <= fn[1]]
            if hit:
                fn = min(hit, key=lambda x: x[1] - x[0])
                region = f"{os.path.basename(f)}:{fn[2]}"
                if os.path.abspath(f) == os.path.abspath(main_src) and fn[2].startswith("k_"):
                    blk = [lab for (mln, lab) in marks if mln <= ln]
                    region = f"{fn[2]} body: {blk[-1] if blk else 'prologue'}"
        census[region][classify(mn)] += 1
    cols = ["SALU", "VALU", "LDS", "VMEM", "SMEM", "waitcnt", "s_nop", "barrier", "branch", "other"]
    rows = sorted(census.items(), key=lambda kv: -sum(kv[1].values()))
    tot = collections.Counter()
    for _, c in rows:
        tot.update(c)
    w = max(len(r) for r, _ in rows) + 2
    o = [f"static instruction census of {name}", f"source {src}, flags {' '.join(extra) or '(product)'}; {sum(tot.values())} instructions; by lexical source region of the .loc line", "",
         "region".ljust(w) + "".join(c.rjust(9) for c in cols) + "    total"]
    for r, c in rows:
        o.append(r.ljust(w) + "".join(str(c.get(k, 0)).rjust(9) for k in cols) + str(sum(c.values())).rjust(9))
    o.append("TOTAL".ljust(w) + "".join(str(tot.get(k, 0)).rjust(9) for k in cols) + str(sum(tot.values())).rjust(9))
    txt = "\n".join(o) + "\n"
    if out:
        open(out, "w").write(txt)
    print(txt)


if __name__ == "__main__":
    main()
