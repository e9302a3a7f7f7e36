"""BASELINE.json config 5 at its full size (5 000 000 triangles), where the CPU oracle is out of reach (the
reference's tree copies triangles into leaves: 23.8 M x 160 B): size-independent properties instead.
  1. the KD-tree built on the GPU equals the host builder's byte for byte (the host builder is pinned to the
     oracle's pointer tree at small sizes, tests/test_host.py);
  2. every traversal mapping returns identical hits, among them the reference-order walk (PTK_OPT_KD_PRUNE = 0:
     the literal recursion of mesh::intersect_internal, scene.cpp:188-212, no pruning, no near-first order) and
     the scene uploaded with the host-built tree;
  3. each hit is self-consistent: position = origin + t * direction, and position = M (v0 + u e0 + v e1) with the
     reported triangle's vertices (the primitive id names the triangle that was hit);
  4. no hit is reported from behind (one-sided triangles, scene.cpp:149-158)."""
import numpy as np
import pytest

from conftest import random_rays
from test_gpu_kd_build import assert_same_tree

pytestmark = pytest.mark.gpu

MODES = {  # PTK_OPT_TRAVERSAL, PTK_OPT_KD_PRUNE, PTK_OPT_NODE_CULL, PTK_OPT_BALANCE, PTK_OPT_SPLIT, PTK_OPT_DEVICE_KD
    "split (default)": (2, 1, 2, 1, 1, 2),
    "balanced": (2, 1, 2, 1, 0, 2),
    "per-lane lists": (2, 1, 2, 0, 0, 2),
    "reference-order walk": (2, 0, 2, 1, 0, 2),
    "uniform node loop, host-built tree": (0, 1, 2, 0, 0, 0),
}


def test_config5_mesh_at_full_size(_gpu_ctx_session):
    from pathtracing_amd import _abi, kd_build, workloads
    ctx = _gpu_ctx_session
    sc = workloads.synthetic_scene(5_000_000, 256)
    node = sc.nodes[0]
    tris = node.shape.triangles
    assert tris.shape[0] == 5_000_000

    host = kd_build(tris)
    assert_same_tree(ctx.kd_build(tris), host)
    assert host[0].num_leaf_prims > 20_000_000 and host[0].max_depth == 100

    rng = np.random.default_rng(99)
    cam = np.asarray(sc.camera.position, np.float32)
    tgt = rng.uniform(-1.3, 1.3, (262_144, 3)).astype(np.float32)
    d_cam = tgt - cam
    d_cam = (d_cam / np.linalg.norm(d_cam, axis=1, keepdims=True)).astype(np.float32)
    o_rand, d_rand = random_rays(1_000_000, 5, -1.6, 1.6)
    o = np.concatenate([np.broadcast_to(cam, d_cam.shape), o_rand]).astype(np.float32)
    d = np.concatenate([d_cam, d_rand]).astype(np.float32)

    outs = {}
    options = (_abi.PTK_OPT_TRAVERSAL, _abi.PTK_OPT_KD_PRUNE, _abi.PTK_OPT_NODE_CULL, _abi.PTK_OPT_BALANCE, _abi.PTK_OPT_SPLIT,
               _abi.PTK_OPT_DEVICE_KD)
    try:
        for name, values in MODES.items():
            for opt, val in zip(options, values):
                ctx.set_option(opt, val)
            ctx.upload(sc)
            outs[name] = ctx.intersect_batch(o, d)
    finally:
        for opt, val in zip(options, (2, 1, 2, 2, 2, 2)):
            ctx.set_option(opt, val)

    ref = outs["reference-order walk"]
    hit = ref["prim"] >= 0
    assert 0.2 < hit.mean() < 0.95, hit.mean()
    for name, got in outs.items():
        for key in ref:
            assert np.array_equal(got[key], ref[key], equal_nan=True), (name, key, int((got[key] != ref[key]).sum()))

    # 3. self-consistency of the hits (float64 reconstruction from the inputs)
    o64, d64 = o[hit].astype(np.float64), d[hit].astype(np.float64)
    pos = ref["position"][hit].astype(np.float64)
    t = ref["t"][hit].astype(np.float64)
    assert np.abs(o64 + d64 * t[:, None] - pos).max() < 2e-4
    tri = tris[ref["prim"][hit]][:, :, 0:3].astype(np.float64)
    u, v = ref["u"][hit].astype(np.float64), ref["v"][hit].astype(np.float64)
    assert (u >= 0).all() and (v >= 0).all() and (u + v <= 1.0 + 1e-5).all()
    local = tri[:, 0] + u[:, None] * (tri[:, 1] - tri[:, 0]) + v[:, None] * (tri[:, 2] - tri[:, 0])
    m = node.transform.T.astype(np.float64)  # column-major storage -> math matrix
    world = local @ m[:3, :3].T + m[:3, 3]
    assert np.abs(world - pos).max() < 2e-4, np.abs(world - pos).max()
    # 4. one-sided: the face normal (v1 - v0) x (v2 - v1) in world space faces the ray
    n_local = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 1])
    n_world = n_local @ np.linalg.inv(m[:3, :3])  # (M^-T n)^T = n^T M^-1
    assert (np.einsum("ij,ij->i", n_world, d64) < 0).all()
