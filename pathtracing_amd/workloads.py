"""Synthetic workload for BASELINE.json config 5: a displaced UV-sphere
with an exact triangle count, per-vertex normals and uvs, a procedural albedo texture and an
equirectangular HDR background.  The reference would build it through mesh::add_triangle
(scene.h:268); here the triangles go straight into the harness's Mesh.  Everything is a pure
function of the arguments (fixed seeds), so the CPU oracle and the GPU see identical inputs.
"""
import numpy as np

from . import scene as S

SEED = 0x5EED


def _value_noise(u, v, cells, rng_seed):
    """Bilinear value noise on a periodic-in-u lattice of `cells` x `cells` random values."""
    rng = np.random.default_rng(rng_seed)
    lat = rng.random((cells + 1, cells), dtype=np.float32)
    x = u * cells
    y = v * cells
    x0 = np.floor(x).astype(np.int64)
    y0 = np.minimum(np.floor(y).astype(np.int64), cells - 1)
    fx = (x - x0).astype(np.float32)
    fy = (y - y0).astype(np.float32)
    fx = fx * fx * (3 - 2 * fx)
    fy = fy * fy * (3 - 2 * fy)
    xa, xb = x0 % cells, (x0 + 1) % cells
    a = lat[y0, xa] * (1 - fx) + lat[y0, xb] * fx
    b = lat[y0 + 1, xa] * (1 - fx) + lat[y0 + 1, xb] * fx
    return (a * (1 - fy) + b * fy).astype(np.float32)


def displaced_sphere(num_triangles, amplitude=0.12, seed=SEED):
    """(T,3,8) float32 triangles (position, normal, uv) of a UV-sphere whose radius is displaced by
    three octaves of value noise.  num_triangles = 2*nu*nv with nu ~ 2.5*nv (5M -> 2500 x 1000)."""
    target = np.sqrt(num_triangles / 5.0)
    fits = [n for n in range(2, int(4 * target) + 3) if num_triangles % (2 * n) == 0 and num_triangles // (2 * n) >= 3]
    if not fits:
        raise ValueError(f"num_triangles={num_triangles} cannot be written as 2*nu*nv")
    nv = min(fits, key=lambda n: abs(n - target))
    nu = num_triangles // (2 * nv)
    # latitude stops short of the poles so no triangle is degenerate
    eps = 0.5 * np.pi / nv
    theta = np.linspace(eps, np.pi - eps, nv + 1, dtype=np.float64)            # polar angle
    phi = np.linspace(0.0, 2.0 * np.pi, nu + 1, dtype=np.float64)             # azimuth (last = first)
    uu = (np.arange(nu + 1, dtype=np.float64) / nu)[None, :].repeat(nv + 1, 0)
    vv = (np.arange(nv + 1, dtype=np.float64) / nv)[:, None].repeat(nu + 1, 1)
    un = (uu % 1.0).astype(np.float32)
    r = 1.0 + amplitude * (_value_noise(un, vv.astype(np.float32), 8, seed) - 0.5) \
        + 0.5 * amplitude * (_value_noise(un, vv.astype(np.float32), 32, seed + 1) - 0.5) \
        + 0.25 * amplitude * (_value_noise(un, vv.astype(np.float32), 128, seed + 2) - 0.5)
    r[:, -1] = r[:, 0]  # close the seam exactly
    st, ct = np.sin(theta)[:, None], np.cos(theta)[:, None]
    sp, cp = np.sin(phi)[None, :], np.cos(phi)[None, :]
    pos = np.stack([r * st * sp, r * ct * np.ones_like(sp), r * st * cp], axis=-1)
    pos[:, -1] = pos[:, 0]
    # smooth normals from central differences of the displaced surface
    du = np.roll(pos[:, :-1], -1, axis=1) - np.roll(pos[:, :-1], 1, axis=1)
    du = np.concatenate([du, du[:, :1]], axis=1)
    dv = np.gradient(pos, axis=0)
    nrm = np.cross(du, dv)   # (phi, theta) order: outward for this parametrisation
    nrm /= np.linalg.norm(nrm, axis=-1, keepdims=True)
    outward = np.sum(nrm * pos, axis=-1, keepdims=True) < 0
    nrm = np.where(outward, -nrm, nrm)
    vert = np.concatenate([pos, nrm, np.stack([uu, 1.0 - vv], axis=-1)], axis=-1).astype(np.float32)  # (nv+1, nu+1, 8)
    a = vert[:-1, :-1]
    b = vert[:-1, 1:]
    c = vert[1:, 1:]
    d = vert[1:, :-1]
    # counter-clockwise seen from outside (the reference's triangles are one-sided, scene.cpp:149-158)
    t0 = np.stack([a, d, c], axis=2)
    t1 = np.stack([a, c, b], axis=2)
    tris = np.stack([t0, t1], axis=2).reshape(-1, 3, 8)
    return np.ascontiguousarray(tris, dtype=np.float32)


def checker_noise_texture(size, seed=SEED):
    """size x size float RGB: a 32x32 checker tinted by value noise (linear colour, in [0,1])."""
    y, x = np.mgrid[0:size, 0:size]
    u = (x.astype(np.float32) + 0.5) / size
    v = (y.astype(np.float32) + 0.5) / size
    check = (((x * 32 // size) + (y * 32 // size)) & 1).astype(np.float32)
    n = _value_noise(u, v, 64, seed + 7)
    base = 0.25 + 0.6 * check
    rgb = np.stack([base * (0.7 + 0.3 * n), base * (0.9 - 0.3 * n), base * (0.5 + 0.5 * n)], axis=-1)
    return np.ascontiguousarray(np.clip(rgb, 0.0, 1.0), dtype=np.float32)


def synthetic_scene(num_triangles=5_000_000, texture_size=4096, background_image=None, seed=SEED):
    """The C5 scene: one mesh node (textured albedo, roughness 0.7), equirect HDR background
    (or a uniform grey sky when no image is given), camera outside looking at the sphere."""
    sc = S.Scene()
    sc.camera.position = np.array([0.0, 0.6, 3.6], dtype=np.float32)
    sc.camera.set_direction([0.0, -0.15, -1.0])
    if background_image is not None:
        sc.background = S.EquirectangularMap(background_image)
    else:
        sc.background = S.ColorSampler([0.7, 0.7, 0.7])
    node = S.SceneNode()
    node.rotate([0.0, 1.0, 0.0], np.float32(0.4))
    node.scale([1.1, 1.1, 1.1])
    node.shape = S.Mesh(displaced_sphere(num_triangles, seed=seed))
    node.material.albedo = S.Image(checker_noise_texture(texture_size, seed))
    node.material.roughness = S.ColorSampler([0.7, 0.7, 0.7])
    sc.nodes.append(node)
    return sc
