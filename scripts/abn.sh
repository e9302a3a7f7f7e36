#!/bin/bash
# A/B/C... of several builds of the library on the GPU box: scripts/abn.sh "<workloads>" steps lib1 lib2 ...
wl=$1; steps=$2; shift 2
mkdir -p gpurun_out
for rep in 1 2; do for lib in "$@"; do for w in $wl; do
  extra=""; [ "$w" = synthetic ] && extra="--triangles 1000000"
  PTK_LIB=$lib timeout -k 10 300 python bench.py --workload $w --steps $steps --warmup 3 --no-cpu-baseline $extra > gpurun_out/ab.json 2>gpurun_out/ab.err || { echo "FAILED $lib $w"; tail -3 gpurun_out/ab.err; exit 1; }
  python - "$lib" "$w" <<'P'
import json,sys
d=json.load(open("gpurun_out/ab.json"))
print(f"{sys.argv[1]:34s} {sys.argv[2]:10s} {d['value']:9.1f} Mpaths/s {d['ms_per_step']:8.3f} ms/step")
P
done; done; done
