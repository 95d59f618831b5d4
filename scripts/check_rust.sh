#!/bin/bash
# The Rust side of the drop-in boundary (integration/teeline-gpu: the FFI crate, gpu.rs, the patch against the reference) has never
# met a compiler in this image (no cargo / rustc, no network).  Wherever a toolchain exists this is the FIRST thing to run:
#   scripts/check_rust.sh [path-to-a-teeline-checkout]
# It type-checks the FFI crate against include/teeline_gpu.h's shapes (cargo check), and, given a checkout of the reference,
# applies integration/patches/0001-gpu-feature.patch to a scratch copy and checks that too.  Without cargo it says so, loudly,
# and exits 3: "uncompiled" is a result of its own, neither a pass nor a failure of the tree (tests/test_abi.py reports it as an
# expected failure, so it stays visible in every test summary).
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
if ! command -v cargo >/dev/null 2>&1; then
  echo "check_rust.sh: *** cargo NOT FOUND — integration/teeline-gpu (Rust FFI crate, gpu.rs, reference patch) is UNCOMPILED on this box ***" >&2
  echo "check_rust.sh: the extern \"C\" block is only checked textually against include/teeline_gpu.h (tests/test_abi.py)" >&2
  exit 3
fi
set -e
echo "check_rust.sh: cargo $(cargo --version)"
( cd "$ROOT/integration/teeline-gpu" && TEELINE_GPU_LIB_DIR="$ROOT/teeline_amd" cargo check --offline 2>&1 || cargo check )
if [ $# -ge 1 ] && [ -d "$1/src/tsp" ]; then
  TMP=$(mktemp -d)
  cp -r "$1" "$TMP/teeline"
  ( cd "$TMP/teeline" && patch -p1 < "$ROOT/integration/patches/0001-gpu-feature.patch" && mkdir -p src/tsp && cp "$ROOT/integration/teeline-gpu/gpu.rs" src/tsp/gpu.rs && cargo check --features gpu )
  rm -rf "$TMP"
fi
echo "check_rust.sh: ok"
