#!/usr/bin/env python3
"""Room PNG residual study: renders room.json and source-level variants of it on the GPU at
the reference image's sample count (250 x 256 spp, 256x256) and compares each, in LINEAR radiance, with
results/room-256-i250.png (tests/golden/reference_results/), the only output of the Room scene the reference holds.

The PNG is inverted through the CLI tone map (CLI/main.cpp:93-103): v = floor(255 (1-exp(-L))^(1/2.2)); blocks whose
reference radiance exceeds 2.5 are excluded (the tone map saturates there: the lamp).  Per variant:
  scale   mean over valid 8x8-pixel blocks of GPU / PNG radiance
  spread  standard deviation of that ratio over the blocks (noise floor + structure)
  roi     the same ratio over the region where the committed scene deviates most: the back wall above the right box
          (pixel rows 40-145, columns 178-223)
  band    ... and over the band above the chrome sphere (rows 150-190, columns 100-170)
Writes gpurun_out/room_residual.json and the ratio maps (gpurun_out/room_residual_<tag>.npy, 32x32 blocks).
"""
import copy
import json
import os
import sys

import numpy as np

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
os.makedirs("gpurun_out", exist_ok=True)

from PIL import Image  # noqa: E402

from conftest import GOLDEN, scene_path  # noqa: E402
from pathtracing_amd import Context, load_scene, scene as S  # noqa: E402

ITER = int(os.environ.get("ROOM_ITER", "250"))
B = 8

ref = np.asarray(Image.open(f"{GOLDEN}/reference_results/room-256-i250.png")).astype(np.float64)[..., :3]
c = ((ref + 0.5) / 255.0) ** 2.2
Lref = -np.log(np.clip(1.0 - c, 1e-6, 1.0))


def blk(a):
    return a.reshape(256 // B, B, 256 // B, B, 3).mean(axis=(1, 3)).mean(axis=2)


Rb = blk(Lref)
valid = Rb < 2.5
ctx = Context(0)


def fresh():
    return load_scene(scene_path("room"))


def rough(sc, node, v):
    sc.nodes[node].material.roughness = S.ColorSampler([v, 0.0, 0.0])


def metal(sc, node, v):
    sc.nodes[node].material.metallic = S.ColorSampler([v, 0.0, 0.0])


LAMP, LBOX, RBOX, BALL, FLOOR, CEIL, BACK, FRONT, LEFT, RIGHT = range(10)


def v_base(sc):
    pass


VARIANTS = {"base": v_base}
for name, node in (("ball", BALL), ("lbox", LBOX), ("rbox", RBOX), ("floor", FLOOR), ("back", BACK), ("right", RIGHT), ("left", LEFT),
                   ("ceil", CEIL)):
    for val in (0.3, 0.9):
        VARIANTS[f"{name}_r{int(val * 10):02d}"] = (lambda node, val: lambda sc: rough(sc, node, val))(node, val)
VARIANTS["boxes_r09"] = lambda sc: (rough(sc, LBOX, 0.9), rough(sc, RBOX, 0.9))
VARIANTS["lbox_r03_rbox_r09"] = lambda sc: (rough(sc, LBOX, 0.3), rough(sc, RBOX, 0.9))
VARIANTS["lbox_r09_rbox_r03"] = lambda sc: (rough(sc, LBOX, 0.9), rough(sc, RBOX, 0.3))
VARIANTS["ball_r03_metal0"] = lambda sc: (rough(sc, BALL, 0.3), metal(sc, BALL, 0.0))
VARIANTS["ball_metal09"] = lambda sc: metal(sc, BALL, 0.9)
VARIANTS["walls_r09"] = lambda sc: [rough(sc, n, 0.9) for n in (FLOOR, CEIL, BACK, FRONT, LEFT, RIGHT)]


def lamp_emission(sc, v):
    sc.nodes[LAMP].material.emission = S.ColorSampler([v, v, v])


VARIANTS["lamp_e24"] = lambda sc: lamp_emission(sc, 24.0)

only = os.environ.get("ROOM_VARIANTS")
rows = {}
for tag, fn in VARIANTS.items():
    if only and tag not in only.split(","):
        continue
    sc = fresh()
    fn(sc)
    ctx.upload(sc)
    ctx.render_begin(256, 256)
    img = ctx.render_image(seed=0, iterations=ITER, spi=256).astype(np.float64)
    ratio = blk(img) / Rb
    np.save(f"gpurun_out/room_residual_{tag}.npy", ratio.astype(np.float32))
    rv = ratio[valid]
    px = lambda r0, r1, c0, c1: ratio[r0 // B:r1 // B, c0 // B:c1 // B]
    rows[tag] = dict(scale=float(rv.mean()), spread=float(rv.std()), spread_rel=float((rv / rv.mean()).std()),
                     roi=float(px(40, 145, 178, 223).mean()), band=float(px(150, 190, 100, 170).mean()),
                     left_corner=float(px(40, 145, 40, 80).mean()))
    print(tag, json.dumps(rows[tag]), flush=True)
with open("gpurun_out/room_residual.json", "w") as f:
    json.dump(rows, f, indent=1)
