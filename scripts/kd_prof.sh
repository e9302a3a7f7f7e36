#!/bin/bash
# Kernel-trace summary of one GPU KD build.  Usage: scripts/kd_prof.sh <triangles>
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof
mkdir -p $out; rm -rf $out/kd_kt
timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d $out/kd_kt --output-format csv -- python3 scripts/kd_build_time.py $1 dev > $out/kd_kt.log 2>&1
f=$(find $out/kd_kt -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'P'
import csv,sys
tot=0
for r in csv.DictReader(open(sys.argv[1])):
    n=r["Name"].replace("void ","").replace("(anonymous namespace)::","").split("(")[0][:60]
    tot+=float(r["TotalDurationNs"])
    print(f'{n:60s} calls={r["Calls"]:>5s} total_ms={float(r["TotalDurationNs"])/1e6:8.2f} avg_us={float(r["AverageNs"])/1e3:9.1f}')
print("sum ms", tot/1e6)
P
tail -3 $out/kd_kt.log
