#!/bin/bash
# Throughput of the C++ rt:: host API (rt_render) on Room 256x256, 256 spp per iteration, from the per-iteration
# spp/sec lines: (a) cumulative at the end, (b) steady state between iteration 16 and the last one.
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/host
python3 - <<'P'
import sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from host_scene_text import write_scene_text
write_scene_text("tests/golden/assets/res/scenes/room.json", "gpurun_out/host/room.txt", "gpurun_out/host")
P
timeout -k 10 120 pathtracing_amd/host/rt_render --scene gpurun_out/host/room.txt --out gpurun_out/host/room.bin --resolution 256 256 --iterations ${1:-200} --spp-per-iteration 256 > gpurun_out/host/room.log 2>&1
python3 - <<'P'
import re
rows=[(int(m.group(1)), float(m.group(2))) for m in re.finditer(r"iteration (\d+) / \d+, ([0-9.]+) spp/sec", open("gpurun_out/host/room.log").read())]
it,rate=rows[-1]
t_end=it*256/rate
i0,r0=rows[15]
t0=i0*256/r0
steady=(it-i0)*256/(t_end-t0)
print(f"rt_render: {it} iterations, cumulative {rate:.0f} spp/s = {rate*65536/1e6:.0f} Mpaths/s; steady state {steady:.0f} spp/s = {steady*65536/1e6:.0f} Mpaths/s; {1e3*(t_end-t0)/(it-i0):.3f} ms per iteration; first 16 iterations {t0*1e3:.0f} ms")
P
