#!/bin/bash
# A/B/... of several builds of the library on the GPU box: scripts/abn.sh "<workloads>" "<lib1> <lib2> ..." [steps] [extra bench args]
wl=${1:-room}; libs=$2; steps=${3:-40}; extra=$4
mkdir -p gpurun_out
for rep in 1 2; do for lib in $libs; do for w in $wl; do
  PTK_LIB=$lib timeout -k 10 300 python bench.py --workload $w --steps $steps --warmup 6 --no-cpu-baseline $extra > gpurun_out/ab.json 2>gpurun_out/ab.err || { echo "FAILED $lib $w"; tail -3 gpurun_out/ab.err; exit 1; }
  python - "$lib" "$w" <<'P'
import json,sys
d=json.load(open("gpurun_out/ab.json"))
print(f"{sys.argv[1]:34s} {sys.argv[2]:10s} {d['value']:9.1f} Mpaths/s {d['mrays_per_s']:9.1f} Mrays/s {d['ms_per_step']:8.3f} ms/step")
P
done; done; done
