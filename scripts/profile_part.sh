#!/bin/bash
# One part of scripts/profile_all.sh (a gpurun call is limited to 20 minutes): scripts/profile_part.sh <round> <tag>...
# tags: room spheres mario synth5m.  Profiles (kernel trace + PMC passes), summaries into profiles/<round>_*, a bench line each.
set -e
rnd=$1; shift
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/bench gpurun_out/profiles_new
for tag in "$@"; do
  case $tag in
    room)    pargs="--steps 10";                                   bargs="";;
    spheres) pargs="--workload spheres --steps 5";                 bargs="--workload spheres --steps 40 --cpu-baseline";;
    # (KD scenes: the profiled runs use one launch chain, --overlap 0, so that a traced launch is the launch bench.py's
    # roofline times -- its timed steps run as one chain, kernels alone on the GPU; the bench line itself runs as shipped)
    mario)   pargs="--workload mario --steps 5 --overlap 0";                   bargs="--workload mario --steps 20 --cpu-baseline";;
    synth5m) pargs="--workload synthetic --steps 2 --warmup 1 --overlap 0";    bargs="--workload synthetic --steps 2 --warmup 1 --cpu-baseline";;
  esac
  scripts/profile.sh $tag $pargs > gpurun_out/profile_$tag.log 2>&1 || { tail -5 gpurun_out/profile_$tag.log; exit 1; }
  python3 scripts/collect_profiles.py $rnd $tag > gpurun_out/collect_$tag.log 2>&1
  timeout -k 10 400 python3 bench.py $bargs > gpurun_out/bench/$tag.json 2> gpurun_out/bench/$tag.err
  cp gpurun_out/bench/$tag.json profiles/${rnd}_${tag}_bench.json
  cp profiles/${rnd}_${tag}_* gpurun_out/profiles_new/
  cut -c1-160 gpurun_out/bench/$tag.json
done
