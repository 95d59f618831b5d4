"""Rebuild the `src/tsp/gpu.rs` part of integration/patches/0001-gpu-feature.patch from integration/teeline-gpu/gpu.rs
(a new-file hunk: every line of the file with a leading '+'); the Cargo.toml and src/tsp/mod.rs hunks are kept as they are.
python scripts/refresh_patch.py [--check]   (--check: exit 1 if the patch does not carry the current file)"""
import os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATCH = os.path.join(ROOT, "integration", "patches", "0001-gpu-feature.patch")
SRC = os.path.join(ROOT, "integration", "teeline-gpu", "gpu.rs")
text = open(PATCH).read()
parts = re.split(r"(?m)^(?=diff )", text)
src = open(SRC).read()
lines = src.split("\n")
if lines and lines[-1] == "":
    lines.pop()
out = []
for part in parts:
    if "+++ b/src/tsp/gpu.rs" not in part:
        out.append(part)
        continue
    head = part.split("\n")
    keep = [l for l in head if l.startswith(("diff ", "--- ", "+++ ", "new file", "index "))]
    hunk = f"@@ -0,0 +1,{len(lines)} @@"
    out.append("\n".join(keep + [hunk] + ["+" + l for l in lines]) + "\n")
new = "".join(out)
if "--check" in sys.argv:
    sys.exit(0 if new == text else 1)
open(PATCH, "w").write(new)
print(f"gpu.rs hunk: {len(lines)} lines")
