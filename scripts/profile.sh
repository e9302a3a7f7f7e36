#!/bin/bash
# Profiles one bench.py workload on the GPU box: kernel trace + three PMC passes (each in its own
# run, as the pool requires).  Usage: scripts/profile.sh <tag> <bench args...>
# Outputs under gpurun_out/prof/<tag>_{kt,sq,fetch,write}; summarise with scripts/pmc_summary.py.
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof
mkdir -p $out
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $out/${tag}_kt --output-format csv -- python3 bench.py "$@" --no-cpu-baseline > $out/${tag}_kt.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD -d $out/${tag}_sq --output-format csv -- python3 bench.py "$@" --no-cpu-baseline > $out/${tag}_sq.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE TCP_TCC_READ_REQ_sum -d $out/${tag}_fetch --output-format csv -- python3 bench.py "$@" --no-cpu-baseline > $out/${tag}_fetch.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $out/${tag}_write --output-format csv -- python3 bench.py "$@" --no-cpu-baseline > $out/${tag}_write.log 2>&1
grep -h '"metric"' $out/${tag}_kt.log | cut -c1-200
python3 scripts/pmc_summary.py $out/${tag}_sq $out/${tag}_fetch $out/${tag}_write
