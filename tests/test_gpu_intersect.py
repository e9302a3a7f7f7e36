"""GPU parity, closest hit: ptk_intersect_batch (HIP) vs the CPU oracle on the same rays.

Bar (BASELINE.json north_star): node and primitive ids bit-exact; t, u, v within 1e-5
relative.  Batches follow SURVEY §8d: (i) primary rays through pixel centres, (ii) first-
bounce secondary rays of the Philox render, (iii) random rays in the scene box.
"""
import numpy as np
import pytest

from conftest import get_oracle, get_scene, planar_cluster, random_rays, random_soup, single_child_nodes

pytestmark = pytest.mark.gpu

SCENES = ["room", "materials", "mario", "furnace"]
REL = 1e-5


def check_batch(gpu, ora, exact_attrs=False):
    assert np.array_equal(gpu["node"], ora["node"]), f"{(gpu['node'] != ora['node']).sum()} node ids differ"
    assert np.array_equal(gpu["prim"], ora["prim"]), f"{(gpu['prim'] != ora['prim']).sum()} primitive ids differ"
    hit = ora["node"] >= 0
    for k in ("t", "u", "v"):
        a, b = gpu[k][hit], ora[k][hit]
        err = np.abs(a - b) / np.maximum(np.abs(b), 1.0 if k != "t" else 1e-30)
        assert err.max(initial=0.0) <= REL, (k, err.max())
    np.testing.assert_allclose(gpu["position"][hit], ora["position"][hit], rtol=REL, atol=1e-5)
    np.testing.assert_allclose(gpu["normal"][hit], ora["normal"][hit], rtol=0, atol=2e-5)
    if exact_attrs:
        assert np.array_equal(gpu["t"][hit], ora["t"][hit])
        assert np.array_equal(gpu["position"][hit], ora["position"][hit])


def centre_rays(ora, w, h):
    ys, xs = np.mgrid[0:h, 0:w]
    xy = np.stack([xs.ravel(), ys.ravel()], axis=1).astype(np.uint32)
    u12 = np.full((w * h, 2), 0.5, np.float32)
    return ora.primary_rays(w, h, np.pi / 4, xy, u12)


@pytest.mark.parametrize("name", SCENES)
def test_primary_centre_rays(gpu_ctx, name):
    ora = get_oracle(name)
    gpu_ctx.upload(get_scene(name))
    o, d = centre_rays(ora, 256, 256)
    check_batch(gpu_ctx.intersect_batch(o, d), ora.intersect_batch(o, d), exact_attrs=True)


@pytest.mark.parametrize("name", SCENES)
def test_secondary_rays(gpu_ctx, name):
    ora = get_oracle(name)
    gpu_ctx.upload(get_scene(name))
    n = 65536
    pixel = np.arange(n, dtype=np.uint32) % (256 * 256)
    sample = np.arange(n, dtype=np.uint64) // 7
    o, d, valid = ora.secondary_rays(256, 256, np.pi / 4, 1234, pixel, sample)
    o, d = o[valid], d[valid]
    assert len(o) > 1000 or name == "furnace"
    check_batch(gpu_ctx.intersect_batch(o, d), ora.intersect_batch(o, d))


@pytest.mark.parametrize("name,lo,hi", [("room", -10, 10), ("materials", -8, 8), ("mario", -10, 10), ("furnace", -3, 3)])
def test_random_rays(gpu_ctx, name, lo, hi):
    ora = get_oracle(name)
    gpu_ctx.upload(get_scene(name))
    n = 1_000_000 if name != "mario" else 200_000
    o, d = random_rays(n, 12345, lo, hi)
    check_batch(gpu_ctx.intersect_batch(o, d), ora.intersect_batch(o, d))


def test_deep_kd_soup(gpu_ctx):
    """A 30k-triangle soup: deep KD-tree, duplicated leaf triangles, scaled + rotated node."""
    from oracle import oracle
    from pathtracing_amd import scene as S
    sc = S.Scene()
    node = S.SceneNode()
    node.translate([0.3, -0.2, 0.1])
    node.rotate([0.2, 1.0, 0.4], 0.7)
    node.scale([1.5, 0.8, 1.2])
    node.shape = S.Mesh(random_soup(30000, 7))
    sc.nodes.append(node)
    sph = S.SceneNode()
    sph.translate([0.0, 0.0, -0.5])
    sph.scale([0.4, 0.4, 0.4])
    sph.shape = S.Sphere()
    sc.nodes.append(sph)
    ora = oracle.OracleScene(sc)
    gpu_ctx.upload(sc)
    info = gpu_ctx.kd_info(0)
    assert info.max_depth >= 8 and info.num_leaf_prims > 30000
    o, d = random_rays(200_000, 99, -2.5, 2.5)
    g, r = gpu_ctx.intersect_batch(o, d), ora.intersect_batch(o, d)
    check_batch(g, r)
    assert (r["prim"] >= 0).sum() > 20000


def test_single_child_kd_nodes(gpu_ctx):
    """Interior KD nodes with only a left or only a right child (planar clusters at non-dyadic x, see
    conftest.planar_cluster): the pruned walk derives a child's box from its parent's by moving one plane to the
    split, so it has to know WHICH child the next preorder slot holds.  Pruned, reference-order and oracle ids agree;
    rays graze the clusters from both sides so triangle parts outside the cut boxes are exercised."""
    from oracle import oracle
    from pathtracing_amd import kd_build, scene as S
    sc = S.Scene()
    seen = [0, 0]
    for k, (x, n) in enumerate([(0.1, 400), (0.7, 400), (-0.1, 401), (-0.7, 1000), (0.3, 1000)]):
        tris = planar_cluster(n, 21 + k, x)
        info, nodes, _ = kd_build(tris)
        lo, ro = single_child_nodes(nodes, info)
        seen[0] += lo
        seen[1] += ro
        node = S.SceneNode()
        node.translate([0.0, 2.5 * (k - 2), 0.0])
        node.rotate([0.0, 0.0, 1.0], 0.05 * k)
        node.shape = S.Mesh(tris)
        sc.nodes.append(node)
    assert seen[0] > 0 and seen[1] > 0, seen
    ora = oracle.OracleScene(sc)
    gpu_ctx.upload(sc)
    rng = np.random.default_rng(5)
    n = 120_000
    o = np.stack([rng.uniform(-3, 3, n), rng.uniform(-6.5, 6.5, n), rng.uniform(-1.5, 1.5, n)], axis=1).astype(np.float32)
    target = np.stack([rng.uniform(-0.8, 0.8, n), o[:, 1] + rng.normal(0, 0.6, n), rng.uniform(-1.2, 1.2, n)], axis=1)
    d = target - o
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    g, r = gpu_ctx.intersect_batch(o, d), ora.intersect_batch(o, d)
    check_batch(g, r)
    assert (r["prim"] >= 0).sum() > 10_000


@pytest.mark.parametrize("limit", [0, 1, 3])
def test_short_stack_overflow_sweeps_the_subtree(gpu_ctx, limit):
    """The pruned KD walk keeps a short stack of pending far children (12 entries per lane in LDS in the walker, 16 in
    the per-lane loops).  When it is full the walk sweeps the current subtree in preorder through the skip links
    instead of spilling.  PTK_OPT_WALK_STACK forces that path: with a limit of 0, 1 or 3 entries nearly every ray
    sweeps, with subtrees nested inside swept subtrees, and the hits must still be the oracle's, id for id."""
    from oracle import oracle
    from pathtracing_amd import _abi, scene as S
    sc = S.Scene()
    for k, (n, seed) in enumerate([(12000, 61), (3000, 62)]):
        node = S.SceneNode()
        node.translate([1.2 * k, 0.1 * k, -0.4 * k])
        node.rotate([0.3, 1.0, 0.2], 0.5 + k)
        node.shape = S.Mesh(random_soup(n, seed, extent=1.0, size=0.1))
        sc.nodes.append(node)
    two = random_soup(2, 14)            # two triangles no split separates: a chain down to depth 100
    two[1, :, 0:3] = two[0, :, 0:3] * np.float32(1.5)
    chain = S.SceneNode()
    chain.translate([-1.5, 0.0, 0.0])
    chain.shape = S.Mesh(two)
    sc.nodes.append(chain)
    ora = oracle.OracleScene(sc)
    o, d = random_rays(150_000, 77, -2.5, 2.5)
    ref = ora.intersect_batch(o, d)
    try:
        gpu_ctx.set_option(_abi.PTK_OPT_WALK_STACK, limit)
        gpu_ctx.upload(sc)
        got = gpu_ctx.intersect_batch(o, d)
    finally:
        gpu_ctx.set_option(_abi.PTK_OPT_WALK_STACK, 64)
    check_batch(got, ref)
    assert (ref["prim"] >= 0).sum() > 10_000 and (ref["node"] == 2).sum() > 0


def test_ragged_and_empty_batches(gpu_ctx):
    ora = get_oracle("room")
    gpu_ctx.upload(get_scene("room"))
    for n in (1, 63, 64, 65, 255, 257, 1000):
        o, d = random_rays(n, n, -10, 10)
        check_batch(gpu_ctx.intersect_batch(o, d), ora.intersect_batch(o, d))
    out = gpu_ctx.intersect_batch(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32))
    assert out["node"].shape == (0,)


def test_axis_aligned_and_degenerate_rays(gpu_ctx):
    """Zero direction components (inf/NaN slab terms, scene.cpp:60-67) and origins on box faces."""
    ora = get_oracle("room")
    gpu_ctx.upload(get_scene("room"))
    dirs = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1],
                     [1, 1, 0], [0, 1, 1], [1, 0, -1]], np.float32)
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    origins = np.array([[0, 0, 0], [0, 0, 29], [-6.5, -5, -6.5], [6, -6, -6], [0, -10, 0], [10, 0, 0],
                        [0, -7, 0], [0, 10, 0], [-10, -10, -10]], np.float32)
    o = np.repeat(origins, len(dirs), axis=0)
    d = np.tile(dirs, (len(origins), 1))
    check_batch(gpu_ctx.intersect_batch(o, d), ora.intersect_batch(o, d))


def test_empty_scene_misses(gpu_ctx):
    from pathtracing_amd import scene as S
    gpu_ctx.upload(S.Scene())
    o, d = random_rays(1000, 3, -1, 1)
    out = gpu_ctx.intersect_batch(o, d)
    assert (out["node"] == -1).all() and (out["prim"] == -1).all()


def test_deep_kd_axis_aligned_and_on_plane_rays(gpu_ctx):
    """Zero direction components and origins exactly on KD split planes / box faces of a deep tree:
    the 0/0 and +-inf slab terms (scene.cpp:60-67) must steer the pruned walk like the reference's."""
    from oracle import oracle
    from pathtracing_amd import kd_build, scene as S
    tris = random_soup(4000, 21)
    sc = S.Scene()
    node = S.SceneNode()
    node.shape = S.Mesh(tris)
    sc.nodes.append(node)
    ora = oracle.OracleScene(sc)
    gpu_ctx.upload(sc)
    info, nodes, _ = kd_build(tris)
    # origins on the first few split planes (left child's max = the split) and on the root box faces
    planes = []
    for i in range(min(200, info.num_nodes)):
        n = nodes[i]
        if n.num_prims == 0 and i + 1 < info.num_nodes:
            ax = n.depth % 3
            planes.append((ax, float(nodes[i + 1].bmax[ax])))
    for ax in range(3):
        planes += [(ax, float(info.bmin[ax])), (ax, float(info.bmax[ax]))]
    rng = np.random.default_rng(5)
    o = rng.uniform(-1.3, 1.3, (len(planes) * 24, 3)).astype(np.float32)
    d = np.zeros_like(o)
    axes = np.eye(3, dtype=np.float32)
    for j in range(len(o)):
        ax, val = planes[j // 24]
        o[j, ax] = np.float32(val)
        k = j % 24
        if k < 6:
            d[j] = axes[k % 3] * (1 if k < 3 else -1)                      # along an axis
        elif k < 12:
            d[j] = axes[(ax + 1) % 3] * (1 if k < 9 else -1)               # inside the split plane
        else:
            v = rng.normal(0, 1, 3).astype(np.float32)
            v[(ax + k) % 3] = 0.0                                          # one zero component
            d[j] = v / np.linalg.norm(v)
    check_batch(gpu_ctx.intersect_batch(o, d), ora.intersect_batch(o, d))


def test_synthetic_displaced_sphere(gpu_ctx):
    """BASELINE.json config 5's mesh generator at a size the oracle handles in seconds (50k triangles):
    primary, inside-out and random rays."""
    from oracle import oracle
    from pathtracing_amd import workloads
    sc = workloads.synthetic_scene(50_000, 256)
    ora = oracle.OracleScene(sc)
    gpu_ctx.upload(sc)
    info = gpu_ctx.kd_info(0)
    assert info.num_leaf_prims >= 50_000 and info.max_depth >= 12
    o, d = centre_rays(ora, 192, 192)
    g, r = gpu_ctx.intersect_batch(o, d), ora.intersect_batch(o, d)
    check_batch(g, r, exact_attrs=True)
    assert (r["prim"] >= 0).mean() > 0.2
    o, d = random_rays(200_000, 77, -1.6, 1.6)
    check_batch(gpu_ctx.intersect_batch(o, d), ora.intersect_batch(o, d))


def test_many_random_nodes(gpu_ctx):
    """40 nodes with random rotations, non-uniform scales (up to 30:1), tiny far-away objects and shared
    meshes: the conservative world-space node cull (PTK_OPT_NODE_CULL) must never drop a node the reference
    reports a hit on, whatever the transform."""
    from oracle import oracle
    from pathtracing_amd import scene as S
    rng = np.random.default_rng(2024)
    sc = S.Scene()
    cube = get_scene("room").nodes  # borrow the cube / plane meshes
    meshes = [n.shape for n in cube if isinstance(n.shape, S.Mesh)]
    small = [meshes[0], meshes[-1], S.Mesh(random_soup(300, 3, extent=1.0, size=0.3))]
    for i in range(40):
        node = S.SceneNode()
        far = 40.0 if i % 7 == 0 else 6.0
        node.translate(rng.uniform(-far, far, 3).astype(np.float32))
        axis = rng.normal(0, 1, 3).astype(np.float32)
        node.rotate(axis, np.float32(rng.uniform(0, 6.28)))
        base = 0.02 if i % 5 == 0 else rng.uniform(0.3, 2.0)
        node.scale((base * rng.uniform(1.0, 30.0 if i % 3 == 0 else 1.5, 3)).astype(np.float32))
        node.shape = S.Sphere() if i % 2 == 0 else small[i % 3]
        sc.nodes.append(node)
    ora = oracle.OracleScene(sc)
    gpu_ctx.upload(sc)
    o, d = random_rays(300_000, 8, -45, 45)
    # aim half of the rays at node centres so that the small / thin ones are actually hit
    centres = np.stack([n.transform[3][:3] for n in sc.nodes]).astype(np.float32)
    tgt = centres[rng.integers(0, len(centres), len(o) // 2)] + rng.normal(0, 0.3, (len(o) // 2, 3)).astype(np.float32)
    dd = tgt - o[: len(tgt)]
    d[: len(tgt)] = (dd / np.linalg.norm(dd, axis=1, keepdims=True)).astype(np.float32)
    g, r = gpu_ctx.intersect_batch(o, d), ora.intersect_batch(o, d)
    check_batch(g, r)
    assert (r["node"] >= 0).mean() > 0.25 and len(np.unique(r["node"])) > 30


def _soup_node(S, tris, translate, axis, angle, scale):
    node = S.SceneNode()
    node.translate(np.asarray(translate, np.float32))
    node.rotate(np.asarray(axis, np.float32), np.float32(angle))
    node.scale(np.asarray(scale, np.float32))
    node.shape = S.Mesh(tris)
    return node


def test_overlapping_trees_and_big_leaves(gpu_ctx):
    """Split mode's walker (k_walk) on its awkward inputs: three overlapping KD-tree nodes (a parked ray
    carries several tree bits, non-uniform scales make local and world distance orders differ), one of
    them with 150 coincident triangles that the 1.5x rule keeps in one leaf (more than the 64 triangles
    a lane contributes per pooled leaf stage), plus spheres and a cube the first kernel resolves."""
    from oracle import oracle
    from pathtracing_amd import scene as S
    rng = np.random.default_rng(5)
    sc = S.Scene()
    cube = [n.shape for n in get_scene("room").nodes if isinstance(n.shape, S.Mesh)][-1]
    clump = random_soup(1, 3, extent=0.2, size=0.5)
    clump = np.repeat(clump, 150, axis=0)
    clump[:, :, :3] += rng.normal(0, 1e-4, (150, 1, 3)).astype(np.float32)
    big_leaf = np.concatenate([random_soup(4000, 11, extent=1.0, size=0.05), clump]).astype(np.float32)
    sc.nodes.append(_soup_node(S, random_soup(20000, 21), [0.2, 0.0, -0.1], [0, 1, 0], 0.4, [1.0, 2.5, 0.6]))
    sc.nodes.append(_soup_node(S, big_leaf, [-0.3, 0.2, 0.0], [1, 0.3, 0], 1.1, [1.4, 0.7, 1.0]))
    sc.nodes.append(_soup_node(S, random_soup(6000, 31), [0.0, -0.1, 0.3], [0.2, 0.1, 1], 2.0, [0.5, 0.5, 3.0]))
    for i in range(3):
        node = S.SceneNode()
        node.translate(rng.uniform(-1, 1, 3).astype(np.float32))
        node.scale(np.full(3, 0.3, np.float32))
        node.shape = S.Sphere() if i < 2 else cube
        sc.nodes.append(node)
    ora = oracle.OracleScene(sc)
    gpu_ctx.upload(sc)
    o, d = random_rays(300_000, 99, -2.5, 2.5)
    g, r = gpu_ctx.intersect_batch(o, d), ora.intersect_batch(o, d)
    check_batch(g, r)
    assert len(np.unique(r["node"][r["node"] >= 0])) == 6
    # a second, much smaller batch reuses the walker's buffers
    o, d = random_rays(777, 100, -2.5, 2.5)
    check_batch(gpu_ctx.intersect_batch(o, d), ora.intersect_batch(o, d))


def test_more_than_64_tree_nodes(gpu_ctx):
    """70 KD-tree nodes do not fit the walker's 64-bit tree mask: the library falls back to the
    in-kernel walk; ids stay exact."""
    from oracle import oracle
    from pathtracing_amd import scene as S
    rng = np.random.default_rng(6)
    sc = S.Scene()
    mesh = S.Mesh(random_soup(400, 41, extent=1.0, size=0.1))
    for i in range(70):
        node = S.SceneNode()
        node.translate(rng.uniform(-6, 6, 3).astype(np.float32))
        node.rotate(rng.normal(0, 1, 3).astype(np.float32), np.float32(rng.uniform(0, 6.28)))
        node.scale(rng.uniform(0.5, 1.5, 3).astype(np.float32))
        node.shape = mesh
        sc.nodes.append(node)
    ora = oracle.OracleScene(sc)
    gpu_ctx.upload(sc)
    o, d = random_rays(200_000, 101, -8, 8)
    check_batch(gpu_ctx.intersect_batch(o, d), ora.intersect_batch(o, d))


def test_mirrored_and_sheared_planes(gpu_ctx):
    """The planar-mesh facing cull (DCull::fn, a ray direction that faces every triangle of a one-sided
    planar mesh from behind cannot hit it) under transforms that flip orientation or shear: negative
    scales, and a node matrix set directly to a sheared one.  Rays graze the planes at every angle."""
    from oracle import oracle
    from pathtracing_amd import scene as S
    rng = np.random.default_rng(77)
    room = get_scene("room")
    meshes = [n.shape for n in room.nodes if isinstance(n.shape, S.Mesh)]
    plane, cube = meshes[0], meshes[-1]
    sc = S.Scene()
    for i in range(24):
        node = S.SceneNode()
        node.translate(rng.uniform(-4, 4, 3).astype(np.float32))
        node.rotate(rng.normal(0, 1, 3).astype(np.float32), np.float32(rng.uniform(0, 6.28)))
        sgn = np.where(rng.uniform(0, 1, 3) < 0.4, -1.0, 1.0)
        node.scale((sgn * rng.uniform(0.3, 3.0, 3)).astype(np.float32))
        if i % 4 == 0:
            shear = np.eye(4, dtype=np.float32)
            shear[1][0] = np.float32(rng.uniform(-0.8, 0.8))  # column-major [col][row]
            shear[2][1] = np.float32(rng.uniform(-0.8, 0.8))
            node.multiply(shear)
        node.shape = plane if i % 3 else cube
        sc.nodes.append(node)
    ora = oracle.OracleScene(sc)
    gpu_ctx.upload(sc)
    o, d = random_rays(400_000, 55, -6, 6)
    # half of the rays start next to a plane and leave it at a shallow angle
    centres = np.stack([n.transform[3][:3] for n in sc.nodes]).astype(np.float32)
    k = len(o) // 2
    o[:k] = centres[rng.integers(0, len(centres), k)] + rng.normal(0, 0.5, (k, 3)).astype(np.float32)
    g, r = gpu_ctx.intersect_batch(o, d), ora.intersect_batch(o, d)
    check_batch(g, r)
    assert (r["node"] >= 0).mean() > 0.2


def test_rays_through_edges_and_corners_of_one_leaf_meshes(gpu_ctx):
    """The root box test of a one-leaf mesh (bounding_box::intersect, scene.cpp:58-75) is decided from approximate
    quotients when their error bounds allow it and by the six IEEE divides otherwise (box_intersect_fast).  Rays aimed
    at the corners, edge points and face borders of a cube and a plane, exactly and a few ulps to either side, rays that
    run inside a face plane, parallel to an axis, or start on a box plane, and rays whose entry distance is exactly -1
    (the reference's `t9 != -1` quirk) sit on that boundary: ids must still be the oracle's, one for one."""
    import os
    from conftest import ASSETS
    from oracle import oracle
    from pathtracing_amd import scene as S
    shapes = []
    for f in ("cube.obj", "plane.obj"):
        arrays = list(S.load_meshes_from_wavefront(os.path.join(ASSETS, "res", "meshes", f)).values())
        shapes.append(S.Mesh(arrays[0]))
    sc = S.Scene()
    for k, shape in enumerate(shapes[:2]):
        node = S.SceneNode()                      # identity: local space == world space, so the targets below are exact
        node.shape = shape
        sc.nodes.append(node)
    tilted = S.SceneNode()
    tilted.translate([4.0, 0.5, -1.0])
    tilted.rotate([0.3, 1.0, 0.2], 0.7)
    tilted.scale([0.5, 1.5, 0.75])
    tilted.shape = shapes[0]
    sc.nodes.append(tilted)
    ora = oracle.OracleScene(sc)
    gpu_ctx.upload(sc)
    rng = np.random.default_rng(23)
    lo = np.min([t.reshape(-1, 8)[:, 0:3].min(axis=0) for t in (shapes[0].triangles,)], axis=0)
    hi = np.max([t.reshape(-1, 8)[:, 0:3].max(axis=0) for t in (shapes[0].triangles,)], axis=0)
    grid = np.array([lo, (lo + hi) / 2, hi], dtype=np.float32)       # per axis: min, mid, max
    pts = np.array([[grid[i, 0], grid[j, 1], grid[k, 2]] for i in range(3) for j in range(3) for k in range(3)], dtype=np.float32)
    o_list, d_list = [], []
    for rep in range(400):
        o = rng.uniform(-4, 4, 3).astype(np.float32)
        if rep % 5 == 0:
            o[rng.integers(3)] = grid[rng.integers(3), 0]            # origin on a box plane
        for p in pts:
            t = p.copy()
            for ax in range(3):                                       # exact, or 1..3 ulps off
                steps = int(rng.integers(-3, 4))
                for _ in range(abs(steps)):
                    t[ax] = np.nextafter(t[ax], np.float32(np.inf if steps > 0 else -np.inf))
            d = (t.astype(np.float64) - o.astype(np.float64))
            n = np.linalg.norm(d)
            if n < 1e-3:
                continue
            o_list.append(o)
            d_list.append((d / n).astype(np.float32))
    for rep in range(3000):                                           # axis-parallel and in-plane rays
        o = rng.uniform(-3, 3, 3).astype(np.float32)
        d = rng.normal(size=3)
        ax = int(rng.integers(3))
        d[ax] = 0.0
        if rep % 2 == 0:
            o[ax] = grid[rng.integers(3), ax]
        if rep % 3 == 0:
            d[(ax + 1) % 3] = 0.0
        d /= np.linalg.norm(d)
        o_list.append(o)
        d_list.append(d.astype(np.float32))
    for rep in range(2000):                                           # entry distance exactly -1 on some axis
        d = rng.normal(size=3)
        d /= np.linalg.norm(d)
        d = d.astype(np.float32)
        ax = int(rng.integers(3))
        o = rng.uniform(-0.9, 0.9, 3).astype(np.float32)
        o[ax] = np.float32(grid[0 if d[ax] > 0 else 2, ax] + d[ax])  # (plane - o) / d == -1 up to rounding
        o_list.append(o)
        d_list.append(d)
    o = np.array(o_list, dtype=np.float32)
    d = np.array(d_list, dtype=np.float32)
    g, r = gpu_ctx.intersect_batch(o, d), ora.intersect_batch(o, d)
    check_batch(g, r)
    assert (r["node"] >= 0).sum() > 2000
