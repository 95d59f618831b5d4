#!/bin/bash
# Register / LDS / scratch usage of every kernel in one HIP source: scripts/kernel_regs.sh two_opt_ref.hip [extra -D flags]
src=$1; shift
out=/tmp/$(basename "$src" .hip).s
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math \
  -I "$(dirname "$0")/../include" --cuda-device-only -S "$@" -o "$out" "$(dirname "$0")/../teeline_amd/csrc/$src" || exit 1
awk '/^  - \.agpr_count/{a=$3} /\.name:/{name=$2} /\.sgpr_count/{s=$2} /\.vgpr_count/{v=$2} /\.vgpr_spill_count/{sp=$2} /\.private_segment_fixed_size/{p=$2} /\.group_segment_fixed_size/{l=$2} /\.wavefront_size/{print name, "vgpr="v, "sgpr="s, "spill="sp, "scratch="p, "lds="l}' "$out"
echo "asm: $out"
