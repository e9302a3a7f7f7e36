#!/bin/bash
# Registers, scratch and LDS of the kernels in ptk.hip for the current sources: scripts/kernel_resources.sh [extra hipcc flags...] 
# (device-only compile, unbundled, read from the code object's notes)
set -e
cd "$(dirname "$0")/.."
tmp=$(mktemp -d)
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize --cuda-device-only -c "$@" -o $tmp/ptk.co pathtracing_amd/csrc/ptk.hip
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$tmp/ptk.co --output=$tmp/ptk.elf --unbundle
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $tmp/ptk.elf | grep -E "^\s+\.name:|\.vgpr_count|private_segment_fixed|group_segment_fixed|vgpr_spill" | paste - - - - - | sed 's/ \+/ /g; s/_ZN12_GLOBAL__N_1//' | awk '{print $4, "lds="$2, "scratch="$6, "vgpr="$8, "spill="$10}' | cut -c1-160
rm -rf $tmp
