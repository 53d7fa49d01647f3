#!/bin/bash
# Register / spill / LDS figures of every kernel in an object file (or the built library), read from the gfx950 code object's
# metadata:  scripts/kernel_regs.sh diffusionkit_amd/csrc/build/gemm256v3.o [name-filter]
set -e
obj=${1:-diffusionkit_amd/libdk_hip.so}
flt=${2:-.}
tmp=$(mktemp -d)
B=/opt/rocm/lib/llvm/bin
$B/llvm-objcopy --dump-section .hip_fatbin=$tmp/fb.bin "$obj" $tmp/copy.o
$B/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$tmp/fb.bin --output=$tmp/k.co --unbundle
$B/llvm-readelf --notes $tmp/k.co | awk -v f="$flt" '
  /\.name:/ {name=$2}
  /\.vgpr_count:/ {v=$2} /\.agpr_count:/ {a=$2} /\.sgpr_count:/ {s=$2} /\.vgpr_spill_count:/ {sp=$2}
  /\.group_segment_fixed_size:/ {l=$2} /\.private_segment_fixed_size:/ {pr=$2}
  /\.wavefront_size:/ { if (name ~ f) printf "%-80s vgpr %3s agpr %3s sgpr %3s spill %3s scratch %5s lds %6s\n", name, v, a, s, sp, pr, l }'
rm -rf "$tmp"
