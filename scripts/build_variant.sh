#!/bin/bash
# Builds an experiment variant of the library from the current sources: scripts/build_variant.sh <out.so> [extra hipcc flags...]
# e.g. scripts/build_variant.sh scripts/libptk_stats.so -DPTK_STATS ; scripts/build_variant.sh scripts/libptk_exp.so -DPTK_EXP
set -e
cd "$(dirname "$0")/.."
out=$1; shift
flags=$(python3 -c "import __graft_entry__ as g; print(' '.join(g.HIPCC_FLAGS))")
c=pathtracing_amd/csrc
/opt/rocm/bin/hipcc $flags "$@" -o "$out" $c/ptk.hip $c/ptk_group.hip $c/kd_build_dev.hip $c/kd_build.cpp
