#!/usr/bin/env python3
"""Does overlapping two launch chains on one GPU pay?  k contexts on device 0, each on a stream of its own, render
disjoint pixel ranges of the same frame at the same time; compared with one context rendering all of it.
Usage: python3 scripts/exp_overlap.py [workload] [width] [spp per step] [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torch  # noqa: E402
from pathtracing_amd import Context  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "mario"
w = int(sys.argv[2]) if len(sys.argv) > 2 else 128
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 256
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 12
scene = bench.load_workload(wl, 5_000_000)
total = torch.zeros(w, w, 3, device="cuda")
for k in (1, 2, 3, 4):
    ctxs, streams = [], []
    for j in range(k):
        c = Context(0)
        c.upload(scene)
        c.render_begin(w, w)
        s = torch.cuda.Stream()
        c.set_stream(s.cuda_stream)
        ctxs.append(c)
        streams.append(s)
    npx = w * w
    cuts = [npx * j // k for j in range(k + 1)]

    def step(i):
        for j, c in enumerate(ctxs):
            c.render_samples_dev(1, i * spp, spp, total.data_ptr(), accumulate=True, pixel_begin=cuts[j], pixel_count=cuts[j + 1] - cuts[j])

    for i in range(3):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(3, 3 + steps):
        step(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{wl} {w}x{w} {spp} spp/step: {k} chains at once: {steps * npx * spp / dt / 1e6:.1f} Mpaths/s ({1e3 * dt / steps:.3f} ms per step)", flush=True)
    for c in ctxs:
        c.close()
