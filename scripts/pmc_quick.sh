#!/bin/bash
# Quick A/B counter pass on the GPU box: one SQ PMC run of bench.py, summarised.
# Usage: scripts/pmc_quick.sh <tag> <bench args...>
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof
mkdir -p $out
rm -rf $out/${tag}_q1 $out/${tag}_q2
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_LDS -d $out/${tag}_q1 --output-format csv -- python3 bench.py "$@" --no-cpu-baseline --steps 3 --warmup 1 > $out/${tag}_q1.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM -d $out/${tag}_q2 --output-format csv -- python3 bench.py "$@" --no-cpu-baseline --steps 3 --warmup 1 > $out/${tag}_q2.log 2>&1
python3 scripts/pmc_summary.py $out/${tag}_q1 $out/${tag}_q2
