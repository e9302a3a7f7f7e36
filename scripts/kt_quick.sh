#!/bin/bash
# Kernel-trace summary of one short bench.py run on the GPU box.  Usage: scripts/kt_quick.sh <tag> <bench args...>
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof
mkdir -p $out; rm -rf $out/${tag}_kt
timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d $out/${tag}_kt --output-format csv -- python3 bench.py "$@" --no-cpu-baseline --steps 5 --warmup 2 > $out/${tag}_kt.log 2>&1
f=$(find $out/${tag}_kt -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'P'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r["Name"].replace("void ","").replace("(anonymous namespace)::","").split("(")[0]
    if float(r["Percentage"])>0.3: print(f'{n:40s} calls={r["Calls"]:>5s} avg_us={float(r["AverageNs"])/1e3:9.1f} pct={r["Percentage"]}')
P
