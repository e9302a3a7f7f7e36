#!/bin/bash
# A/B of two builds of the library on the GPU box: scripts/ab.sh "<workloads>" <libA> <libB> [steps]
wl=${1:-room}; a=${2:-scripts/libptk_base.so}; b=${3:-pathtracing_amd/libptk.so}; steps=${4:-40}
mkdir -p gpurun_out
for rep in 1 2; do for lib in $a $b; do for w in $wl; do
  PTK_LIB=$lib timeout -k 10 200 python bench.py --workload $w --steps $steps --warmup 5 --no-cpu-baseline > gpurun_out/ab.json 2>gpurun_out/ab.err || { echo "FAILED $lib $w"; tail -3 gpurun_out/ab.err; exit 1; }
  python - "$lib" "$w" <<'P'
import json,sys
d=json.load(open("gpurun_out/ab.json"))
print(f"{sys.argv[1]:34s} {sys.argv[2]:10s} {d['value']:9.1f} Mpaths/s {d['mrays_per_s']:9.1f} Mrays/s {d['ms_per_step']:8.3f} ms/step")
P
done; done; done
