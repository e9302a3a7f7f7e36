#!/usr/bin/env python3
"""KD build time, host (ptk_kd_build) vs GPU (ptk_kd_build_dev), and whole ptk_scene_upload with either.
Usage: python3 scripts/kd_build_time.py [triangles]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from pathtracing_amd import Context, _abi, workloads, scene as S  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
dev_only = len(sys.argv) > 2 and sys.argv[2] == "dev"  # for rocprofv3: one GPU build, nothing else
tris = np.ascontiguousarray(workloads.displaced_sphere(n))
lib = _abi.load()
ctx = Context(0)
info = _abi.KdInfo()
if not dev_only:
    t = time.time(); assert lib.ptk_kd_build(tris.ctypes.data, n, C.byref(info), None, 0, None, 0) == 0; host = time.time() - t
    print(f"triangles {n}: nodes {info.num_nodes} leaf prims {info.num_leaf_prims} depth {info.max_depth}", flush=True)
    print(f"host ptk_kd_build      {host*1e3:9.1f} ms", flush=True)
for rep in range(1 if dev_only else 3):
    t = time.time(); ctx._check(lib.ptk_kd_build_dev(ctx._h, tris.ctypes.data, n, None, 0, C.byref(info), None, 0, None, 0)); dev = time.time() - t
    print(f"gpu  ptk_kd_build_dev  {dev*1e3:9.1f} ms (run {rep}; includes the 96 B/triangle upload and the tree download)", flush=True)
if dev_only:
    ctx.close()
    sys.exit(0)
sc = S.Scene()
node = S.SceneNode()
node.shape = S.Mesh(tris)
sc.nodes.append(node)
for mode in (0, 1):
    ctx.set_option(_abi.PTK_OPT_DEVICE_KD, mode)
    t = time.time(); ctx.upload(sc); up = time.time() - t
    print(f"ptk_scene_upload, PTK_OPT_DEVICE_KD={mode}: {up*1e3:9.1f} ms (includes the Python-side flattening)", flush=True)
ctx.close()
