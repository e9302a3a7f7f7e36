"""ctypes wrapper of oracle/libpt_oracle.so (CPU restatement of the reference).

TEST INFRASTRUCTURE: imported only by tests/, bench.py's cpu_baseline leg and
__graft_entry__.smoke().  Never import this from pathtracing_amd/.
"""
import ctypes as C
import os

import numpy as np

from pathtracing_amd import _abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpt_oracle.so")
_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} not built: run `make -C oracle`")
        lib = C.CDLL(LIB_PATH)
        lib.orc_scene_create.restype = C.c_void_p
        lib.orc_scene_create.argtypes = [C.POINTER(_abi.SceneDesc)]
        lib.orc_scene_destroy.argtypes = [C.c_void_p]
        lib.orc_intersect_batch.argtypes = [C.c_void_p, C.c_uint64] + [C.c_void_p] * 11
        lib.orc_render_samples.restype = C.c_uint64
        lib.orc_render_samples.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_float, C.c_uint64, C.c_uint64,
                                           C.c_uint64, C.c_uint64, C.c_uint32, C.c_void_p, C.c_int, C.c_int]
        lib.orc_render_samples_stats.restype = C.c_uint64
        lib.orc_render_samples_stats.argtypes = lib.orc_render_samples.argtypes + [C.c_void_p]
        lib.orc_mt19937_nth.restype = C.c_uint32
        lib.orc_mt19937_nth.argtypes = [C.c_uint32, C.c_uint32]
        lib.orc_mt19937_floats.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p]
        lib.orc_resolve_iteration.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64]
        lib.orc_render_image.restype = C.c_uint64
        lib.orc_render_image.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_float, C.c_uint64, C.c_uint64,
                                         C.c_uint32, C.c_void_p, C.c_int]
        lib.orc_render_image_mt19937.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_float, C.c_uint64,
                                                 C.c_uint32, C.c_void_p]
        lib.orc_kd_export.restype = C.c_int
        lib.orc_kd_export.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(_abi.KdInfo), C.c_void_p, C.c_uint64,
                                      C.c_void_p, C.c_uint64]
        lib.orc_sample_image.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p]
        lib.orc_sample_equirect.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p]
        lib.orc_hemisphere.argtypes = [C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.orc_primary_rays.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_float, C.c_uint64, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_void_p]
        lib.orc_philox.argtypes = [C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint32, C.c_void_p]
        lib.orc_secondary_rays.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_float, C.c_uint64, C.c_uint64,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.orc_set_walk_model.argtypes = [C.c_int]
        lib.orc_render_aov.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_float, C.c_uint64, C.c_uint64, C.c_uint32, C.c_void_p]
        lib.orc_tonemap.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
        lib.orc_shade_probe.argtypes = [C.c_void_p, C.c_uint64] + [C.c_void_p] * 8
        _lib = lib
    return _lib


def set_walk_model(model):
    """0 = the reference's visit-everything KD walk (default), 1 = the model of the library's pruned
    near-first walk (same results by construction; used for equality tests and traffic counts)."""
    load().orc_set_walk_model(int(model))


STAT_NAMES = ["kd_box", "tri_a", "tri_b", "tri_full", "tri_hit", "sphere", "sphere_hit", "node", "node_hit", "ray",
              "shade", "miss", "path", "mesh_win", "texel"]


class OracleScene:
    def __init__(self, scene):
        self._lib = load()
        desc, keep = scene.to_desc()
        self._h = self._lib.orc_scene_create(C.byref(desc))
        del keep

    def close(self):
        if self._h:
            self._lib.orc_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def intersect_batch(self, origins, directions, stats=False):
        o = np.ascontiguousarray(np.asarray(origins, dtype=np.float32).T)
        d = np.ascontiguousarray(np.asarray(directions, dtype=np.float32).T)
        n = o.shape[1]
        out = {
            "node": np.empty(n, np.int32), "prim": np.empty(n, np.int32), "t": np.empty(n, np.float32),
            "u": np.empty(n, np.float32), "v": np.empty(n, np.float32), "position": np.empty((3, n), np.float32),
            "normal": np.empty((3, n), np.float32), "uv": np.empty((2, n), np.float32),
        }
        st = np.zeros(2, np.uint64)
        self._lib.orc_intersect_batch(self._h, n, o.ctypes.data, d.ctypes.data, out["node"].ctypes.data,
                                      out["prim"].ctypes.data, out["t"].ctypes.data, out["u"].ctypes.data,
                                      out["v"].ctypes.data, out["position"].ctypes.data, out["normal"].ctypes.data,
                                      out["uv"].ctypes.data, st.ctypes.data if stats else None)
        for k in ("position", "normal", "uv"):
            out[k] = np.ascontiguousarray(out[k].T)
        if stats:
            out["kd_nodes_tested"], out["tris_tested"] = int(st[0]), int(st[1])
        return out

    def render_samples(self, w, h, fov_y, seed, sample_begin, sample_count, pixel_begin=0, pixel_count=None,
                       out=None, accumulate=False, threads=1):
        if pixel_count is None:
            pixel_count = w * h - pixel_begin
        if out is None:
            out = np.zeros((h, w, 3), np.float32)
        rays = self._lib.orc_render_samples(self._h, w, h, float(np.float32(fov_y)), seed, pixel_begin, pixel_count,
                                            sample_begin, sample_count, out.ctypes.data, 1 if accumulate else 0, threads)
        return out, int(rays)

    def render_samples_stats(self, w, h, fov_y, seed, sample_begin, sample_count, threads=1):
        """(sum image, rays, {op name: count}) — operation counts for the flop model."""
        out = np.zeros((h, w, 3), np.float32)
        st = np.zeros(16, np.uint64)
        rays = self._lib.orc_render_samples_stats(self._h, w, h, float(np.float32(fov_y)), seed, 0, w * h, sample_begin,
                                                  sample_count, out.ctypes.data, 0, threads, st.ctypes.data)
        return out, int(rays), {k: int(st[i]) for i, k in enumerate(STAT_NAMES)}

    def render_image(self, w, h, fov_y, seed, iterations, spi, threads=1):
        img = np.zeros((h, w, 3), np.float32)
        rays = self._lib.orc_render_image(self._h, w, h, float(np.float32(fov_y)), seed, iterations, spi,
                                          img.ctypes.data, threads)
        return img, int(rays)

    def render_image_mt19937(self, w, h, fov_y, iterations, spi):
        img = np.zeros((h, w, 3), np.float32)
        self._lib.orc_render_image_mt19937(self._h, w, h, float(np.float32(fov_y)), iterations, spi, img.ctypes.data)
        return img

    def kd_export(self, mesh):
        info = _abi.KdInfo()
        self._lib.orc_kd_export(self._h, mesh, C.byref(info), None, 0, None, 0)
        nodes = (_abi.KdNode * max(1, info.num_nodes))()
        prims = np.empty(max(1, info.num_leaf_prims), np.uint32)
        self._lib.orc_kd_export(self._h, mesh, C.byref(info), C.cast(nodes, C.c_void_p), info.num_nodes,
                                prims.ctypes.data, info.num_leaf_prims)
        return info, nodes, prims[: info.num_leaf_prims]

    def primary_rays(self, w, h, fov_y, xy, u12):
        xy = np.ascontiguousarray(xy, np.uint32)
        u12 = np.ascontiguousarray(u12, np.float32)
        n = xy.shape[0]
        o = np.empty((3, n), np.float32)
        d = np.empty((3, n), np.float32)
        self._lib.orc_primary_rays(self._h, w, h, float(np.float32(fov_y)), n, xy.ctypes.data, u12.ctypes.data,
                                   o.ctypes.data, d.ctypes.data)
        return np.ascontiguousarray(o.T), np.ascontiguousarray(d.T)

    def secondary_rays(self, w, h, fov_y, seed, pixel, sample):
        pixel = np.ascontiguousarray(pixel, np.uint32)
        sample = np.ascontiguousarray(sample, np.uint64)
        n = pixel.shape[0]
        o = np.empty((3, n), np.float32)
        d = np.empty((3, n), np.float32)
        valid = np.empty(n, np.int32)
        self._lib.orc_secondary_rays(self._h, w, h, float(np.float32(fov_y)), seed, n, pixel.ctypes.data,
                                     sample.ctypes.data, o.ctypes.data, d.ctypes.data, valid.ctypes.data)
        return np.ascontiguousarray(o.T), np.ascontiguousarray(d.T), valid.astype(bool)


def _shade_probe(self, origins, directions, u12):
    """One level of trace_recursive with supplied draws -> dict like Context.probe_shade."""
    o = np.ascontiguousarray(np.asarray(origins, np.float32).reshape(-1, 3).T)
    d = np.ascontiguousarray(np.asarray(directions, np.float32).reshape(-1, 3).T)
    u = np.ascontiguousarray(np.asarray(u12, np.float32).reshape(-1, 2).T)
    n = o.shape[1]
    out = {"node": np.empty(n, np.int32)}
    bufs = {k: np.empty((3, n), np.float32) for k in ("emission", "weight", "next_origin", "next_direction")}
    self._lib.orc_shade_probe(self._h, n, o.ctypes.data, d.ctypes.data, u.ctypes.data, out["node"].ctypes.data,
                              *[bufs[k].ctypes.data for k in ("emission", "weight", "next_origin", "next_direction")])
    for k, v in bufs.items():
        out[k] = np.ascontiguousarray(v.T)
    return out


OracleScene.shade_probe = _shade_probe


def _render_aov(self, aov, w, h, fov_y, seed, spp):
    """Mean of debug_pathtracer::trace over spp jittered primary rays per pixel -> (h, w, 3)."""
    out = np.zeros((h, w, 3), np.float32)
    self._lib.orc_render_aov(self._h, aov, w, h, float(np.float32(fov_y)), seed, 0, spp, out.ctypes.data)
    return out / np.float32(spp)


OracleScene.render_aov = _render_aov


def tonemap(image_rgb):
    """CLI/main.cpp:91-103 -> (rgba uint32 per pixel, the float products c * 255 before truncation)."""
    lib = load()
    img = np.ascontiguousarray(image_rgb, np.float32).reshape(-1, 3)
    rgba = np.empty(img.shape[0], np.uint32)
    scaled = np.empty_like(img)
    lib.orc_tonemap(img.ctypes.data, img.shape[0], rgba.ctypes.data, scaled.ctypes.data)
    return rgba, scaled


def resolve_iteration(sum_rgb, image_rgb, spi, iteration):
    lib = load()
    s = np.ascontiguousarray(sum_rgb, np.float32)
    lib.orc_resolve_iteration(s.ctypes.data, image_rgb.ctypes.data, s.size // 3, spi, iteration)


def sample_image(pixels, mode, uv):
    lib = load()
    px = np.ascontiguousarray(pixels, np.float32)
    uv = np.ascontiguousarray(uv, np.float32)
    out = np.empty((uv.shape[0], 3), np.float32)
    lib.orc_sample_image(px.ctypes.data, px.shape[1], px.shape[0], mode, uv.shape[0], uv.ctypes.data, out.ctypes.data)
    return out


def sample_equirect(pixels, mode, dirs):
    lib = load()
    px = np.ascontiguousarray(pixels, np.float32)
    dirs = np.ascontiguousarray(dirs, np.float32)
    out = np.empty((dirs.shape[0], 3), np.float32)
    lib.orc_sample_equirect(px.ctypes.data, px.shape[1], px.shape[0], mode, dirs.shape[0], dirs.ctypes.data, out.ctypes.data)
    return out


def hemisphere(normals, u12):
    lib = load()
    n = np.ascontiguousarray(normals, np.float32)
    u = np.ascontiguousarray(u12, np.float32)
    out = np.empty_like(n)
    lib.orc_hemisphere(n.shape[0], n.ctypes.data, u.ctypes.data, out.ctypes.data)
    return out


def philox(seed, pixel, sample, block):
    lib = load()
    out = np.empty(4, np.uint32)
    lib.orc_philox(seed, pixel, sample, block, out.ctypes.data)
    return out
