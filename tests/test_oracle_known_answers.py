"""Pins the CPU oracle (oracle/pt_oracle.c) without a reference build: the reference cannot be
compiled here (glm/spdlog absent) and ships no tests, so the oracle is checked against
(1) published known answers of the generators it restates, (2) hand-derived geometry cases
read off the reference source, (3) the furnace scene's closed form, and (4) the tonemapped
PNG the reference itself holds.  Anything beyond these is "parity unpinned" (DESIGN.md)."""
import numpy as np
import pytest
from PIL import Image

from conftest import GOLDEN, get_oracle, get_scene
from oracle import oracle
from pathtracing_amd import scene as S

FOV = np.pi / 4


def one_node_scene(shape, setup=None):
    sc = S.Scene()
    n = S.SceneNode()
    if setup:
        setup(n)
    n.shape = shape
    sc.nodes.append(n)
    return sc


def test_philox_known_answers():
    """Random123 kat_vectors, philox4x32-10."""
    assert [hex(x) for x in oracle.philox(0, 0, 0, 0)] == ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]
    # all-ones counter and key: pixel, sample(lo,hi), block all 0xffffffff
    lib = oracle.load()
    out = np.empty(4, np.uint32)
    lib.orc_philox(0xFFFFFFFFFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFFFFFFFFFF, 0xFFFFFFFF, out.ctypes.data)
    assert [hex(x) for x in out] == ["0x408f276d", "0x41c83b0e", "0xa20bc7c6", "0x6d5451fd"]


def test_mt19937_known_answer():
    """C++ standard [rand.predef]: the 10000th output of a default-constructed mt19937 is 4123659995."""
    lib = oracle.load()
    assert lib.orc_mt19937_nth(5489, 10000) == 4123659995
    f = np.empty(1000, np.float32)
    lib.orc_mt19937_floats(5489, 1000, f.ctypes.data)
    assert f.min() >= 0.0 and f.max() < 1.0
    # first draw of the default engine is 3499211612 -> float(3499211612)/2^32
    assert f[0] == np.float32(3499211612) / np.float32(4294967296.0)


def test_sphere_analytic_cases():
    """sphere::intersect, scene.cpp:401-441, on the unit sphere with an identity node."""
    ora = oracle.OracleScene(one_node_scene(S.Sphere()))
    o = np.array([[0, 0, 5], [0, 0, 0], [0, 0, 5], [0, 2, 5], [0, 0, -5], [0, 1, 5]], np.float32)
    d = np.array([[0, 0, -1], [0, 0, -1], [0, 0, 1], [0, 0, -1], [0, 0, -1], [0, 0, -1]], np.float32)
    r = ora.intersect_batch(o, d)
    assert r["node"].tolist() == [0, 0, -1, -1, -1, 0]
    assert r["prim"].tolist() == [-1] * 6
    assert r["t"][0] == 4.0 and np.allclose(r["position"][0], [0, 0, 1])
    assert np.allclose(r["normal"][0], [0, 0, 1])
    # uv = (atan2(n.x, n.z)/pi + 0.5, n.y*0.5 + 0.5): front pole -> (0.5, 0.5)
    assert np.allclose(r["uv"][0], [0.5, 0.5])
    # origin inside: t1 < 0 <= t2, exits through the back, normal still points outward
    assert r["t"][1] == 1.0 and np.allclose(r["position"][1], [0, 0, -1]) and np.allclose(r["normal"][1], [0, 0, -1])
    # back of the sphere: u = atan2(0,-1)/pi + 0.5 = 1.5 (u spans [-0.5, 1.5])
    assert np.allclose(r["uv"][1], [1.5, 0.5])
    # grazing ray at sq_distance == 1 still hits (test is `> 1`)
    assert r["t"][5] == 5.0


def test_triangle_is_one_sided_and_node_transform():
    """mesh::intersect_triangle, scene.cpp:142-186: back faces and origins behind the plane miss;
    cast_ray works in node-local space and reports world space (scene.cpp:371-383)."""
    tri = np.zeros((1, 3, 8), np.float32)
    tri[0, :, :3] = [[-1, -1, 0], [1, -1, 0], [0, 1, 0]]  # CCW seen from +z -> face normal +z
    tri[0, :, 3:6] = [0, 0, 1]
    tri[0, :, 6:8] = [[0, 0], [1, 0], [0, 1]]

    def setup(n):
        n.translate([10, 0, 0])
        n.scale([2, 2, 2])

    ora = oracle.OracleScene(one_node_scene(S.Mesh(tri), setup))
    o = np.array([[10, 0, 5], [10, 0, -5], [10, -1, 5], [14, 0, 5]], np.float32)
    d = np.array([[0, 0, -1], [0, 0, 1], [0, 0, -1], [0, 0, -1]], np.float32)
    r = ora.intersect_batch(o, d)
    assert r["node"].tolist() == [0, -1, 0, -1]
    assert r["prim"].tolist() == [0, -1, 0, -1]
    assert r["t"][0] == 5.0 and np.allclose(r["position"][0], [10, 0, 0])
    # local hit (0,0,0): v2 = (1,1,0); barycentrics (u,v,w) = (0.25, 0.25, 0.5); (u,v) := (bar.y, bar.z)
    assert np.allclose([r["u"][0], r["v"][0]], [0.25, 0.5])
    assert np.allclose(r["uv"][0], [0.25, 0.5])  # uv = bar.x*uv0 + bar.y*uv1 + bar.z*uv2
    assert np.allclose(r["normal"][0], [0, 0, 1])


def test_slab_quirks():
    """bounding_box::intersect, scene.cpp:58-75: a box entered at distance exactly -1 is reported
    as a miss (`t9 != -1`); zero direction components go through inf/NaN-ignoring fmin/fmax."""
    # one big triangle in the plane z = 0 spanning the box [-2,2]^2 x {0}; mesh bounds include the origin
    tri = np.zeros((2, 3, 8), np.float32)
    tri[0, :, :3] = [[-2, -2, 0], [2, -2, 0], [-2, 2, 0]]
    tri[1, :, :3] = [[2, -2, 0], [2, 2, 0], [-2, 2, 0]]
    tri[:, :, 3:6] = [0, 0, 1]
    ora = oracle.OracleScene(one_node_scene(S.Mesh(tri)))
    # axis-parallel ray (dx = dy = 0): hits
    r = ora.intersect_batch(np.array([[0.5, 0.5, 3]], np.float32), np.array([[0, 0, -1]], np.float32))
    assert r["node"][0] == 0 and r["t"][0] == 3.0
    # flat box z in [0,0]: origin at z = +1 moving +z: t = (0-1)/1 = -1 for both z planes, t8 = -1 < 0 -> miss anyway
    r = ora.intersect_batch(np.array([[0.5, 0.5, 1]], np.float32), np.array([[0, 0, 1]], np.float32))
    assert r["node"][0] == -1
    # a thick box entered at exactly t7 = -1 with the origin inside: make the mesh thick by a far vertex
    tri2 = np.concatenate([tri, tri[:1]], axis=0).copy()
    tri2[2, :, :3] = [[-2, -2, 4], [2, -2, 4], [-2, 2, 4]]   # z = 4, faces +z: culled for a ray going +z
    ora2 = oracle.OracleScene(one_node_scene(S.Mesh(tri2)))
    info, nodes, prims = ora2.kd_export(0)
    assert info.num_nodes == 1  # 3 triangles: 3+... the 1.5x rule keeps the root a leaf
    # origin at z=1 looking -z: box z-range [0,4], entry t7 = (4-1)/-1 = -3, exit 1 -> hit the z=0 plane at t=1
    r = ora2.intersect_batch(np.array([[0.5, 0.5, 1]], np.float32), np.array([[0, 0, -1]], np.float32))
    assert r["node"][0] == 0 and r["t"][0] == 1.0
    # origin at z=3 looking -z: entry t7 = (4-3)/-1 = -1 exactly -> the quirk turns a real hit into a miss
    r = ora2.intersect_batch(np.array([[0.5, 0.5, 3]], np.float32), np.array([[0, 0, -1]], np.float32))
    assert r["node"][0] == -1


def test_plane_and_cube_kd_trees_collapse_to_one_leaf():
    """plane (2 tris: 2+2 > 3) and cube (12 tris: 10+10 > 18) stay a single root leaf; mesh
    bounds grow from the default (0,0,0) box (scene.cpp:125-136)."""
    ora = get_oracle("room")
    info, nodes, prims = ora.kd_export(0)  # cube is referenced first (node 1)
    assert (info.num_nodes, info.num_leaves, info.num_leaf_prims, info.max_depth) == (1, 1, 12, 0)
    assert list(info.bmin) == [-1, -1, -1] and list(info.bmax) == [1, 1, 1]
    assert prims.tolist() == list(range(12))
    info, nodes, prims = ora.kd_export(1)
    assert (info.num_nodes, info.num_leaf_prims) == (1, 2)
    assert list(info.bmin) == [-10, 0, -10] and list(info.bmax) == [10, 0, 10]


def test_kd_hand_built_split():
    """Four well separated triangles along x: mean split at depth 0 (axis x) puts two on each side,
    children at depth 1 split on y where all centroids coincide -> 2+2 > 1.5*2 -> leaves."""
    tri = np.zeros((4, 3, 8), np.float32)
    for i, x in enumerate([-3.0, -2.0, 2.0, 3.0]):
        tri[i, :, :3] = [[x, 0, 0], [x + 0.5, 0, 0], [x, 0.5, 0]]
    ora = oracle.OracleScene(one_node_scene(S.Mesh(tri)))
    info, nodes, prims = ora.kd_export(0)
    assert info.num_nodes == 3 and info.num_leaves == 2 and info.max_depth == 1
    # mean of all 12 x-coordinates = (4*(-3-2+2+3) + 4*0.5... ) / 12 = (0 + 2)/12
    split = np.float32(0)
    for x in [-3.0, -2.0, 2.0, 3.0]:
        for v in (x, x + 0.5, x):
            split = np.float32(split + np.float32(v))
    split = np.float32(split / np.float32(12))
    assert nodes[0].num_prims == 0 and nodes[0].skip == 3
    assert nodes[1].bmax[0] == split and nodes[2].bmin[0] == split
    assert nodes[1].bmin[0] == -3.0 and nodes[2].bmax[0] == 3.5  # children boxes are cuts of the parent box
    assert prims.tolist() == [0, 1, 2, 3]
    assert (nodes[1].first_prim, nodes[1].num_prims, nodes[2].first_prim, nodes[2].num_prims) == (0, 2, 2, 2)


def test_image_sampler_semantics():
    """image::sample, sampler.cpp:36-61, on a 2x2 image with distinct texels."""
    px = np.array([[[1, 0, 0], [0, 1, 0]], [[0, 0, 1], [1, 1, 1]]], np.float32)  # px[y][x]
    uv = np.array([[0.0, 0.0], [0.5, 0.0], [0.25, 0.0], [0.75, 0.5], [1.25, 2.0], [-0.75, -1.0], [0.5, 0.5]], np.float32)
    lin = oracle.sample_image(px, 0, uv)
    # x = u*W with no half-texel offset: u=0 -> texel 0 exactly, u=0.5 -> texel 1 exactly
    assert lin[0].tolist() == [1, 0, 0] and lin[1].tolist() == [0, 1, 0]
    assert np.allclose(lin[2], [0.5, 0.5, 0])            # halfway between texel 0 and 1
    # u=0.75: x=1.5 -> x0=1, x1=ceil(1.5)%2=0 (wraps), weight 0.5; v=0.5 -> y=1 exactly -> row 1
    assert np.allclose(lin[3], [0.5, 0.5, 1.0])
    assert np.allclose(lin[4], lin[2]) and np.allclose(lin[5], lin[2])  # fract() wrap, negative too
    assert lin[6].tolist() == [1, 1, 1]
    near = oracle.sample_image(px, 1, uv)
    assert near[0].tolist() == [1, 0, 0] and near[2].tolist() == [0, 1, 0]  # round(0.5) = 1 (half away from zero)
    assert near[3].tolist() == [0, 0, 1]                                     # round(1.5)=2 -> %2 = 0, row 1


def test_equirect_mapping():
    """equirectangular_map::sample, sampler.cpp:116-124: +z -> u=0.5; v = asin(y)/pi - 0.5 wraps to +0.5."""
    w, h = 8, 4
    px = np.zeros((h, w, 3), np.float32)
    px[..., 0] = np.arange(w)[None, :]
    px[..., 1] = np.arange(h)[:, None]
    out = oracle.sample_equirect(px, 1, np.array([[0, 0, 1], [1, 0, 0], [0, 0, -1]], np.float32))
    # +z: u=0.5 -> x=4; v=-0.5 -> fract=0.5 -> y=2
    assert out[0].tolist() == [4, 2, 0]
    assert out[1].tolist() == [6, 2, 0]      # +x: u = 0.25 + 0.5
    assert out[2][0] in (0.0, 8.0 % 8)       # -z: u = 1.0 -> wraps to column 0


def test_hemisphere_mapping():
    """rng::hemisphere, rng.cpp:12-36: z=u1 along the normal, uniform in the hemisphere, not renormalised."""
    n = np.tile(np.array([[0, 0, 1]], np.float32), (4, 1))
    u = np.array([[1.0, 0.0], [0.0, 0.0], [0.0, 0.25], [0.5, 0.5]], np.float32)
    d = oracle.hemisphere(n, u)
    assert np.allclose(d[0], [0, 0, 1])
    # n=(0,0,1): |n.x| > |n.y| false -> t = normalize(0,-1,0) ; b = cross(n,t) = (1,0,0)
    assert np.allclose(d[1], [0, -1, 0], atol=1e-6)
    assert np.allclose(d[2], [1, 0, 0], atol=1e-6)
    assert np.allclose(np.linalg.norm(d, axis=1), 1, atol=1e-6) and np.all(d[:, 2] >= 0)


def test_primary_ray_convention():
    """abstract_pathtracer.cpp:20-25,74-83: h2 = atan(fov/2) (not tan), pixel (0,0) is top-left, jitter 0.5 = centre."""
    ora = get_oracle("furnace")
    w = h = 4
    o, d = ora.primary_rays(w, h, FOV, np.array([[0, 0], [3, 3], [2, 2]], np.uint32), np.full((3, 2), 0.5, np.float32))
    assert np.allclose(o, [[0, 0, 5]] * 3)
    h2 = np.arctan(np.float32(FOV) / 2)
    # centre of pixel (0,0): fx = 0 -> x_factor = -1, y_factor = +1
    e = np.array([-h2, h2, -1.0]); e /= np.linalg.norm(e)
    assert np.allclose(d[0], e, atol=1e-6)
    e = np.array([0.5 * h2, -0.5 * h2, -1.0]); e /= np.linalg.norm(e)  # pixel (3,3): fx=3 -> 0.5, fy -> -0.5
    assert np.allclose(d[1], e, atol=1e-6)


def test_furnace_closed_form():
    """furnace.json: E[L] = 0.18 on the sphere, exactly 1 on the background."""
    ora = get_oracle("furnace")
    w = h = 32
    img, rays = ora.render_image(w, h, FOV, 11, 2, 256, threads=8)
    ys, xs = np.mgrid[0:h, 0:w]
    r = np.hypot(xs - w / 2 + 0.5, ys - h / 2 + 0.5)
    assert np.all(img[r > 10.5] == 1.0)
    assert abs(img[r < 7].mean() - 0.18) < 0.003


def test_mt19937_and_philox_renders_agree_statistically():
    ora = get_oracle("furnace")
    a, _ = ora.render_image(24, 24, FOV, 5, 1, 128, threads=4)
    b = ora.render_image_mt19937(24, 24, FOV, 1, 128)
    assert abs(a.mean() - b.mean()) < 0.01


def test_room_against_reference_png():
    """Coarse pin against the reference's own output (results/room-256-i250.png, CLI tone map
    1-exp(-c), gamma 1/2.2, truncation to 8 bit; CLI/main.cpp:93-103).  The PNG is inverted to linear
    radiance, box-filtered 4x4 and compared with a 1024-spp 64x64 oracle render in the LINEAR domain
    (a sample mean is unbiased there; after the concave tone map a noisy image reads darker).
    Saturated texels (the light) are excluded."""
    ora = get_oracle("room")
    img, _ = ora.render_image(64, 64, FOV, 0, 4, 256, threads=8)
    ref = np.asarray(Image.open(f"{GOLDEN}/reference_results/room-256-i250.png")).astype(np.float64)[..., :3]
    lin = -np.log(np.clip(1 - np.power((ref + 0.5) / 255, 2.2), 1e-6, 1))
    ok = (ref < 250).all(axis=2).reshape(64, 4, 64, 4).all(axis=(1, 3))
    lin64 = lin.reshape(64, 4, 64, 4, 3).mean(axis=(1, 3))
    assert ok.mean() > 0.9
    a, b = img.astype(np.float64)[ok], lin64[ok]
    rel = np.abs(a.mean(axis=0) - b.mean(axis=0)) / b.mean(axis=0)
    # measured: +1.5..2.5 % (oracle brighter), with structured deviations of up to -7 % in concave
    # corners (DESIGN.md "What pins the oracle"); the PNG may predate the committed scene / code
    assert np.all(rel < 0.035), rel
    # per texel the 1024-spp side is dominated by Monte-Carlo noise (radiance-25 light, ~16 % median)
    assert np.median(np.abs(a - b) / (b + 0.05)) < 0.3


@pytest.mark.parametrize("name,lo,hi,n", [("mario", -8, 8, 150_000), ("room", -10, 10, 100_000)])
def test_pruned_walk_model_equals_reference_walk(name, lo, hi, n):
    """The oracle's model of the library's pruned near-first KD walk must return exactly what
    the reference's visit-everything recursion returns — every output bit — while touching fewer nodes."""
    from conftest import get_oracle, random_rays
    from oracle import oracle as O
    ora = get_oracle(name)
    o, d = random_rays(n, 4242, lo, hi)
    pix = np.arange(40_000, dtype=np.uint32) % (128 * 128)
    so, sd, valid = ora.secondary_rays(128, 128, np.pi / 4, 9, pix, np.arange(40_000, dtype=np.uint64) // 3)
    o = np.concatenate([o, so[valid]])
    d = np.concatenate([d, sd[valid]])
    try:
        O.set_walk_model(0)
        ref = ora.intersect_batch(o, d, stats=True)
        O.set_walk_model(1)
        mod = ora.intersect_batch(o, d, stats=True)
    finally:
        O.set_walk_model(0)
    for k in ("node", "prim", "t", "u", "v", "position", "normal", "uv"):
        assert np.array_equal(ref[k], mod[k]), k
    assert mod["kd_nodes_tested"] <= ref["kd_nodes_tested"] and mod["tris_tested"] <= ref["tris_tested"]
    if name == "mario":
        assert mod["tris_tested"] < 0.4 * ref["tris_tested"]
