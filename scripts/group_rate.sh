#!/bin/bash
# Rate of the C++ host (rt_render) on Room 256x256: one context at 256 spp per iteration (the CLI's shape), and a group of
# four contexts on GPU 0 at the same 256 spp per iteration (64 per member: the strong-scaling shape of config 4 on one GPU).
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/host
python3 - <<'P'
import sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from host_scene_text import write_scene_text
write_scene_text("tests/golden/assets/res/scenes/room.json", "gpurun_out/host/room.txt", "gpurun_out/host")
P
for devs in "0" "0,0,0,0" "0,0,0,0,0,0,0,0"; do
  timeout -k 10 120 pathtracing_amd/host/rt_render --scene gpurun_out/host/room.txt --out gpurun_out/host/room.bin --resolution 256 256 --iterations ${1:-200} --spp-per-iteration 256 --devices $devs > gpurun_out/host/room_$devs.log 2>&1
  python3 - "$devs" <<'P'
import re, sys
rows=[(int(m.group(1)), float(m.group(2))) for m in re.finditer(r"iteration (\d+) / \d+, ([0-9.]+) spp/sec", open(f"gpurun_out/host/room_{sys.argv[1]}.log").read())]
it,rate=rows[-1]
t_end=it*256/rate
i0,r0=rows[15]
t0=i0*256/r0
steady=(it-i0)*256/(t_end-t0)
print(f"rt_render --devices {sys.argv[1]}: {it} iterations x 256 spp, cumulative {rate*65536/1e6:.0f} Mpaths/s; steady state {steady*65536/1e6:.0f} Mpaths/s; {1e3*(t_end-t0)/(it-i0):.3f} ms per iteration")
P
done
