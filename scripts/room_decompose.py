#!/usr/bin/env python3
"""Room PNG residual study, part 2: which light carries the residual?  Renders room.json with one object at a time
made black (albedo 0, metallic 0: every path through it ends), so base - black_X is the radiance that touched X, and
regresses the PNG's residual on those contribution maps.  Writes gpurun_out/room_decomp_<tag>.npy (8x8 block means)."""
import os, sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests")); os.makedirs("gpurun_out", exist_ok=True)
from conftest import scene_path
from pathtracing_amd import Context, load_scene, scene as S
ITER = int(os.environ.get("ROOM_ITER", "250"))
ctx = Context(0)
names = {"base": None, "ball": 3, "lbox": 1, "rbox": 2, "floor": 4, "ceil": 5, "back": 6, "front": 7, "left": 8, "right": 9}
for tag, node in names.items():
    sc = load_scene(scene_path("room"))
    if node is not None:
        sc.nodes[node].material.albedo = S.ColorSampler([0, 0, 0])
        sc.nodes[node].material.metallic = S.ColorSampler([0, 0, 0])
    ctx.upload(sc); ctx.render_begin(256, 256)
    img = ctx.render_image(seed=0, iterations=ITER, spi=256)
    np.save(f"gpurun_out/room_decomp_{tag}.npy", img.reshape(128, 2, 128, 2, 3).mean(axis=(1, 3)).astype(np.float32))
    print(tag, float(img.mean()), flush=True)
