#!/bin/bash
# Duration of the bounce-2 launch of k_bounce<false> with the cast cut short after a stage (PTK_EXP build, see stage_budget.sh).
# Usage (GPU box): scripts/stage_time.sh [workload]
wl=${1:-room}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof/stagetime
mkdir -p $out
export PTK_LIB=scripts/libptk_exp.so
for stage in 1 2 3 0; do
  rm -rf $out/s$stage
  PTK_EXP_STAGE=$stage timeout -s KILL 120 rocprofv3 --kernel-trace -d $out/s$stage --output-format csv -- python3 bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline > $out/s$stage.log 2>&1 || true
  f=$(find $out/s$stage -name "*kernel_trace.csv" | head -1)
  python3 - "$f" $stage <<'P'
import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if "k_bounce<false" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
d=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in rows]
b2=d[0::4]
print(f"stage {sys.argv[2]}: bounce-2 launch {sum(b2)/len(b2):8.1f} us over {len(b2)} launches; all bounces {[round(sum(d[k::4])/len(d[k::4]),1) for k in range(4)]}")
P
done
