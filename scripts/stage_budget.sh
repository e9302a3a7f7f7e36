#!/bin/bash
# Per-stage instruction budget of k_bounce<false> on Room: runs the PTK_EXP build of the library
# with the bounce-2 launch cut short after a stage (PTK_EXP_STAGE: 1 = load + cull loop, 2 = + ray->local / sphere / box
# stage, 3 = + triangle stage = whole cast, 0 = whole kernel with shading and compaction) and reads that launch's
# per-dispatch SQ counters.  Differences of successive rows = the stages.  Usage (GPU box): scripts/stage_budget.sh [workload]
set -e
wl=${1:-room}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof/stage
mkdir -p $out
export PTK_LIB=scripts/libptk_exp.so
for stage in 1 2 3 0; do
  rm -rf $out/s$stage
  PTK_EXP_STAGE=$stage timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES \
      -d $out/s$stage --output-format csv -- python3 bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline > $out/s$stage.log 2>&1 || true   # the stage-limited image is garbage: bench.py's own check fails, the counters are what matters
done
python3 scripts/stage_budget.py $out > gpurun_out/stage_budget_$wl.txt
cat gpurun_out/stage_budget_$wl.txt
