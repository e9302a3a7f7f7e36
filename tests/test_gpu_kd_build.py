"""the KD build on the GPU (ptk_kd_build_dev) against the host builder (ptk_kd_build),
which tests/test_host.py pins node for node to the oracle's pointer tree (oracle/pt_oracle.c, kd_tree_node
ctor scene.cpp:214-286).  The bar is byte equality of the whole ptk_kd_node array (cut boxes, skip links,
first_prim / num_prims, depth, padded triangle boxes), of the leaf-ordered primitive list and of the info."""
import ctypes as C

import numpy as np
import pytest

from conftest import get_scene, random_rays, random_soup

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx(_gpu_ctx_session):
    return _gpu_ctx_session


def as_bytes(nodes, count):
    return np.frombuffer(nodes, dtype=np.uint8, count=count * 64).reshape(count, 64)


def assert_same_tree(dev, host):
    (di, dn, dp), (hi, hn, hp) = dev, host
    assert (di.num_nodes, di.num_leaves, di.num_leaf_prims, di.max_depth) == (hi.num_nodes, hi.num_leaves, hi.num_leaf_prims, hi.max_depth)
    assert np.array_equal(np.array(list(di.bmin) + list(di.bmax), np.float32).view(np.uint32),
                          np.array(list(hi.bmin) + list(hi.bmax), np.float32).view(np.uint32))
    assert np.array_equal(dp, hp)
    a, b = as_bytes(dn, di.num_nodes), as_bytes(hn, hi.num_nodes)
    bad = np.nonzero((a != b).any(axis=1))[0]
    assert bad.size == 0, f"first differing node {bad[0]}: {a[bad[0]].view(np.uint32)} != {b[bad[0]].view(np.uint32)}"


@pytest.mark.parametrize("n,seed", [(1, 1), (2, 2), (3, 3), (7, 8), (200, 4), (5000, 5), (120_000, 6)])
def test_random_soups(ctx, n, seed):
    from pathtracing_amd import kd_build
    tris = random_soup(n, seed)
    assert_same_tree(ctx.kd_build(tris), kd_build(tris))


def test_mario(ctx):
    from pathtracing_amd import kd_build
    mario = get_scene("mario").nodes[1].shape
    assert_same_tree(ctx.kd_build(mario.triangles), kd_build(mario.triangles))


def test_displaced_sphere_300k(ctx):
    """The C5 mesh generator at 300 000 triangles: long runs of triangles sharing vertices, so many entries sit
    exactly on a split and go both ways; chains of two-triangle nodes run into the depth-100 stop (scene.cpp:220)."""
    from pathtracing_amd import kd_build, workloads
    tris = workloads.displaced_sphere(300_000)
    dev, host = ctx.kd_build(tris), kd_build(tris)
    assert_same_tree(dev, host)
    assert host[0].max_depth > 20


def test_lattice_coordinates_many_ties(ctx):
    """Vertices on a coarse integer lattice: splits land exactly on vertex coordinates (<= and >= both hold),
    many identical triangles, many nodes kept whole by the 1.5x rule."""
    from pathtracing_amd import kd_build
    rng = np.random.default_rng(11)
    tris = random_soup(20_000, 12)
    tris[:, :, 0:3] = rng.integers(-4, 5, (20_000, 3, 3)).astype(np.float32)
    assert_same_tree(ctx.kd_build(tris), kd_build(tris))


def test_identical_triangles_root_leaf(ctx):
    from pathtracing_amd import kd_build
    tris = np.repeat(random_soup(1, 9), 1000, axis=0)
    dev = ctx.kd_build(tris)
    assert dev[0].num_nodes == 1 and dev[0].num_leaf_prims == 1000
    assert_same_tree(dev, kd_build(tris))


def test_two_overlapping_triangles_reach_depth_limit(ctx):
    """Two triangles that no split separates (2 -> 1 + 2 <= 3): a chain down to depth 100."""
    from pathtracing_amd import kd_build
    tris = random_soup(2, 14)
    tris[1, :, 0:3] = tris[0, :, 0:3] * np.float32(1.5)
    dev, host = ctx.kd_build(tris), kd_build(tris)
    assert_same_tree(dev, host)


def test_non_finite_and_signed_zero_vertices(ctx):
    """NaN coordinates fail every comparison (a triangle can drop out of both children, scene.cpp:257-265, and a
    NaN split empties a node); infinities and -0.0 go through the bounds and the padding as on the host."""
    from pathtracing_amd import kd_build
    for seed, poison in ((31, np.nan), (32, np.inf), (33, -np.inf), (34, -0.0)):
        tris = random_soup(3000, seed)
        rng = np.random.default_rng(seed)
        rows = rng.integers(0, 3000, 40)
        tris[rows, rng.integers(0, 3, 40), rng.integers(0, 3, 40)] = np.float32(poison)
        if poison == -0.0:
            tris[:, :, 1] = np.abs(tris[:, :, 1])  # no negative y: mesh bounds keep +0.0 from the default box
            tris[rows, 0, 1] = np.float32(-0.0)
        assert_same_tree(ctx.kd_build(tris), kd_build(tris))


def test_wide_dynamic_range_sum_order(ctx):
    """Coordinates spanning 12 orders of magnitude: the split is only reproduced by summing in the reference's
    order (scene.cpp:240-249); any regrouping changes the rounding and then the tree."""
    from pathtracing_amd import kd_build
    rng = np.random.default_rng(77)
    tris = random_soup(50_000, 78)
    scale = np.float32(10.0) ** rng.integers(-6, 7, (50_000, 1, 1)).astype(np.float32)
    tris[:, :, 0:3] *= scale
    assert_same_tree(ctx.kd_build(tris), kd_build(tris))


def test_bounded_build_with_root_depth(ctx):
    """The public kd_tree_node constructor (scene.h:159): caller-supplied root box and root depth."""
    from pathtracing_amd import _abi
    lib = _abi.load()
    tris = random_soup(4000, 41)
    bounds = np.array([-2, -3, -4, 2.5, 3.5, 4.5], np.float32)
    for root_depth in (1, 2, 97):
        dev = ctx.kd_build(tris, bounds=bounds, root_depth=root_depth)
        info = _abi.KdInfo()
        assert lib.ptk_kd_build_bounded(tris.ctypes.data, 4000, bounds.ctypes.data, root_depth, C.byref(info), None, 0, None, 0) == 0
        nodes = (_abi.KdNode * info.num_nodes)()
        prims = np.empty(info.num_leaf_prims, np.uint32)
        assert lib.ptk_kd_build_bounded(tris.ctypes.data, 4000, bounds.ctypes.data, root_depth, C.byref(info),
                                        C.cast(nodes, C.c_void_p), info.num_nodes, prims.ctypes.data, info.num_leaf_prims) == 0
        assert_same_tree(dev, (info, nodes, prims))


def test_empty_mesh_is_refused(ctx):
    from pathtracing_amd.renderer import PtkError
    with pytest.raises(PtkError):
        ctx.kd_build(np.zeros((0, 3, 8), np.float32))


def test_scene_upload_uses_either_builder_with_same_hits(ctx):
    """ptk_scene_upload with PTK_OPT_DEVICE_KD = 1 and = 0: same tree info, same hits on random rays."""
    from pathtracing_amd import _abi
    sc = get_scene("mario")
    o, d = random_rays(200_000, 17, -3, 3)
    outs, infos = [], []
    try:
        for mode in (0, 1):
            ctx.set_option(_abi.PTK_OPT_DEVICE_KD, mode)
            ctx.upload(sc)
            i = ctx.kd_info(0)
            infos.append((i.num_nodes, i.num_leaves, i.num_leaf_prims, i.max_depth, list(i.bmin), list(i.bmax)))
            outs.append(ctx.intersect_batch(o, d))
    finally:
        ctx.set_option(_abi.PTK_OPT_DEVICE_KD, 2)
    assert infos[0] == infos[1]
    for key in outs[0]:
        assert np.array_equal(outs[0][key], outs[1][key], equal_nan=True), key
    assert (outs[0]["node"] >= 0).sum() > 1000


def test_scene_with_host_and_device_built_meshes(ctx):
    """One scene holding one-leaf meshes (host, pair table), a host-built tree and two device-built trees (one of
    them planar: the facing cull applies to it): tables of the device-built meshes sit behind the host-built ones.
    Hits with PTK_OPT_DEVICE_KD = 2 (threshold), 1 (all on the GPU, the one-leaf meshes come back through the
    download path) and 0 (all on the host) are the same, and equal to the oracle's on a sample."""
    from oracle import oracle
    from pathtracing_amd import _abi, scene as S
    sc = S.Scene()
    specs = [("plane", None), ("soup", random_soup(3000, 51)), ("big", random_soup(40_000, 52, extent=1.5)),
             ("flat", None), ("cube", None)]
    flat = random_soup(36_000, 53, extent=2.0)
    flat[:, :, 1] = 0.0           # every vertex in the plane y = 0 ...
    flat[:, :, 3:6] = [0.0, 1.0, 0.0]
    a, b = flat[:, 1, 0:3] - flat[:, 0, 0:3], flat[:, 2, 0:3] - flat[:, 1, 0:3]
    flip = np.cross(a, b)[:, 1] < 0  # ... and every face normal +y
    flat[flip] = flat[flip][:, [0, 2, 1], :]
    base = get_scene("room")
    plane = next(n.shape for n in base.nodes if isinstance(n.shape, S.Mesh) and n.shape.triangles.shape[0] == 2)
    cube = next(n.shape for n in base.nodes if isinstance(n.shape, S.Mesh) and n.shape.triangles.shape[0] == 12)
    rng = np.random.default_rng(5)
    for name, tris in specs:
        node = S.SceneNode()
        node.translate(rng.uniform(-1, 1, 3).astype(np.float32))
        node.rotate(rng.normal(0, 1, 3).astype(np.float32), np.float32(rng.uniform(-2, 2)))
        node.scale(rng.uniform(0.6, 1.4, 3).astype(np.float32))
        node.shape = {"plane": plane, "cube": cube, "flat": S.Mesh(flat)}.get(name) or S.Mesh(tris)
        sc.nodes.append(node)
    o, d = random_rays(150_000, 23, -4, 4)
    outs = []
    try:
        for mode in (2, 1, 0):
            ctx.set_option(_abi.PTK_OPT_DEVICE_KD, mode)
            ctx.upload(sc)
            outs.append(ctx.intersect_batch(o, d))
    finally:
        ctx.set_option(_abi.PTK_OPT_DEVICE_KD, 2)
    for other in outs[1:]:
        for key in outs[0]:
            assert np.array_equal(outs[0][key], other[key], equal_nan=True), key
    hit_nodes = set(np.unique(outs[0]["node"]).tolist())
    assert hit_nodes >= {0, 1, 2, 3, 4}
    ora = oracle.OracleScene(sc)
    ref = ora.intersect_batch(o[:20_000], d[:20_000])
    assert np.array_equal(ref["node"], outs[0]["node"][:20_000]) and np.array_equal(ref["prim"], outs[0]["prim"][:20_000])
