#!/bin/bash
# On the GPU box: rocprofv3 kernel trace + PMC passes and a bench line (with cpu_baseline) per workload.
# Outputs under gpurun_out/prof and gpurun_out/bench; summarise here with scripts/collect_profiles.py.
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/bench
scripts/profile.sh room --steps 10
scripts/profile.sh spheres --workload spheres --steps 5
scripts/profile.sh mario --workload mario --steps 5
scripts/profile.sh synth5m --workload synthetic --steps 2 --warmup 1
timeout -k 10 200 python3 bench.py > gpurun_out/bench/room.json 2> gpurun_out/bench/room.err
timeout -k 10 200 python3 bench.py --workload spheres --steps 10 --cpu-baseline > gpurun_out/bench/spheres.json 2> gpurun_out/bench/spheres.err
timeout -k 10 200 python3 bench.py --workload mario --steps 10 --cpu-baseline > gpurun_out/bench/mario.json 2> gpurun_out/bench/mario.err
timeout -k 10 400 python3 bench.py --workload synthetic --steps 2 --warmup 1 --cpu-baseline > gpurun_out/bench/synth5m.json 2> gpurun_out/bench/synth5m.err
cut -c1-150 gpurun_out/bench/*.json
