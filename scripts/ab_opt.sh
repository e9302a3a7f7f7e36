#!/bin/bash
# A/B of bench.py argument sets on the GPU box: scripts/ab_opt.sh "<workloads>" <steps> "<args A>" "<args B>" ...
wl=$1; steps=$2; shift 2
mkdir -p gpurun_out
for rep in 1 2; do for a in "$@"; do for w in $wl; do
  timeout -k 10 300 python bench.py --workload $w --steps $steps --warmup 6 --no-cpu-baseline $a > gpurun_out/ab.json 2>gpurun_out/ab.err || { echo "FAILED $a $w"; tail -3 gpurun_out/ab.err; exit 1; }
  python - "$a" "$w" <<'P'
import json,sys
d=json.load(open("gpurun_out/ab.json"))
print(f"{sys.argv[1]:34s} {sys.argv[2]:10s} {d['value']:9.1f} Mpaths/s {d['mrays_per_s']:9.1f} Mrays/s {d['ms_per_step']:8.3f} ms/step")
P
done; done; done
