#!/bin/bash
# On the GPU box: rocprofv3 kernel trace + PMC passes per workload, summarised into profiles/<round>_* (scripts/collect_profiles.py),
# then a bench line per workload (with cpu_baseline; its roofline.traffic reads the PMC summary just written).
# Everything is copied to gpurun_out/profiles_new/ for the way home.  Usage: scripts/profile_all.sh [round tag, default r02]
set -e
rnd=${1:-r02}
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/bench gpurun_out/profiles_new
scripts/profile.sh room --steps 10
scripts/profile.sh spheres --workload spheres --steps 5
scripts/profile.sh mario --workload mario --steps 5
scripts/profile.sh synth5m --workload synthetic --steps 2 --warmup 1
python3 scripts/collect_profiles.py $rnd room spheres mario synth5m > gpurun_out/collect.log 2>&1
timeout -k 10 200 python3 bench.py > gpurun_out/bench/room.json 2> gpurun_out/bench/room.err
timeout -k 10 200 python3 bench.py --workload spheres --steps 40 --cpu-baseline > gpurun_out/bench/spheres.json 2> gpurun_out/bench/spheres.err
timeout -k 10 200 python3 bench.py --workload mario --steps 20 --cpu-baseline > gpurun_out/bench/mario.json 2> gpurun_out/bench/mario.err
timeout -k 10 400 python3 bench.py --workload synthetic --steps 2 --warmup 1 --cpu-baseline > gpurun_out/bench/synth5m.json 2> gpurun_out/bench/synth5m.err
for w in room spheres mario synth5m; do cp gpurun_out/bench/$w.json profiles/${rnd}_${w}_bench.json; done
cp profiles/${rnd}_* gpurun_out/profiles_new/
cut -c1-150 gpurun_out/bench/*.json
