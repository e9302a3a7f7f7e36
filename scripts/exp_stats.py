#!/usr/bin/env python3
"""Work counters of the balanced traversal (experiment build: hipcc ... -DPTK_STATS -o scripts/libptk_stats.so).
Usage: PTK_LIB=scripts/libptk_stats.so python3 scripts/exp_stats.py [workload] [w] [spp]"""
import ctypes as C
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import numpy as np  # noqa: E402
import torch  # noqa: E402
from pathtracing_amd import Context, _abi  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "room"
w = int(sys.argv[2]) if len(sys.argv) > 2 else 256
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 64
scene = bench.load_workload(wl, 1_000_000)
ctx = Context(0)
ctx.set_option(4, 1)
if len(sys.argv) > 4:
    ctx.set_option(5, int(sys.argv[4]))
ctx.upload(scene)
ctx.render_begin(w, w, math.pi / 4)
lib = _abi.load()
lib.ptk_debug_stats.restype = C.c_int
lib.ptk_debug_stats.argtypes = [C.c_void_p, C.c_void_p]
out = (C.c_ulonglong * 64)()
lib.ptk_debug_stats(ctx._h, out)
total = torch.zeros(w, w, 3, device="cuda")
ctx.render_samples_dev(0, 0, spp, total.data_ptr())
torch.cuda.synchronize()
lib.ptk_debug_stats(ctx._h, out)
v = list(out)
waves = max(1, v[0])
print(f"waves {v[0]}  candidates/wave {v[1]/waves:.1f}  batches/wave {v[2]/waves:.2f}  tri stages/wave {v[3]/waves:.2f}  "
      f"tri iterations/wave {v[4]/waves:.2f}  P/stage {v[6]/max(1,v[3]):.1f}  slots/stage {v[7]/max(1,v[3]):.1f}")
print(f"cycles/wave: kernel(to shade end) {v[12]/waves:.0f}  intersect {v[11]/waves:.0f}  batches {v[9]/waves:.0f} (of which tri stages {v[10]/waves:.0f})  "
      f"=> cull loop {(v[11]-v[9])/waves:.0f}  batch excl tri {(v[9]-v[10])/waves:.0f}  shade+io {(v[12]-v[11])/waves:.0f}")
if v[22]:
    rays = max(1, v[24])
    print(f"walker: parked rays {v[24]} in {v[25]} refills; descend steps {v[22]} (avg lanes {v[23]/v[22]:.1f}, {v[23]/rays:.1f} per ray); "
          f"leaf stages {v[26]} (avg leaf lanes {v[27]/max(1,v[26]):.1f}), triangle items {v[28]} ({v[28]/rays:.1f} per ray) in {v[18]} iterations "
          f"({v[28]/max(1,v[18]):.1f} per iteration); pop passes {v[29]} (avg lanes {v[30]/max(1,v[29]):.1f}); "
          f"items the side tests alone reject: {v[31]} ({100.0*v[31]/max(1,v[28]):.1f} %)")
if v[21]:
    if True:
        wv = v[21]
        print(f"walker clocks per persistent wave ({wv} sampled): total {v[20]/wv:.0f}; descend {v[13]/wv:.0f} ({v[13]/max(1,v[22]):.0f} per pass); "
              f"pop {v[14]/wv:.0f} ({v[14]/max(1,v[29]):.0f} per pass); leaf {v[15]/wv:.0f} ({v[15]/max(1,v[26]):.0f} per stage); "
              f"refill {v[17]/wv:.0f}; fin/start {v[19]/wv:.0f}")
        if os.environ.get("PTK_STATS_COARSE"):
            print(f"  (coarse build) of the refill: {v[37]/wv:.0f} clocks in {v[38]/wv:.1f} grabs per wave ({v[37]/max(1,v[38]):.0f} per grab)")
            for b in range(5):
                if v[40 + b]:
                    print(f"  bounce {b}: {v[40+b]} waves, mean {v[32+b]/v[40+b]:.0f} clocks, slowest {v[48+b]} ({v[48+b]*v[40+b]/max(1,v[32+b]):.2f} x mean); longest drain {v[57+b]}")
            print(f"  stackless sweeps: {v[62]} node visits in all, longest single sweep call {v[63]}")
            print(f"  drain (dry -> end) mean {v[56]/max(1,sum(v[40:45])):.0f} clocks")
ctx.close()
