#!/bin/bash
# Latency counters of one short bench.py run.  Usage: scripts/pmc_lat.sh <tag> <bench args...>
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof; mkdir -p $out; rm -rf $out/${tag}_lat
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_INST_LEVEL_SMEM -d $out/${tag}_lat --output-format csv -- python3 bench.py "$@" --no-cpu-baseline --steps 3 --warmup 1 > $out/${tag}_lat.log 2>&1
python3 scripts/pmc_summary.py $out/${tag}_lat
