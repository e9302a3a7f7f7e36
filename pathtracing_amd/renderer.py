"""Python mirror of the reference's renderer API over the C ABI (include/ptk.h):
view_parameters / trace_parameters / pathtracer_result / abstract_pathtracer::run
(reference src/Pathtracing/abstract_pathtracer.h:90-105,111-174,190 and
abstract_pathtracer.cpp:11-119).  Same names, argument meaning and event order as the
reference; the per-pixel loop runs on the GPU through libptk.so.  There is no CPU path
here: without the HIP library every call raises.
"""
import ctypes as C
import math
import threading
import time

import numpy as np

from . import _abi


class PtkError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"ptk error {code}: {message}")
        self.code = code


class Group:
    """Owns one ptk_group: several contexts (device ids may repeat) behind one driver (include/ptk.h, ptk_group_*)."""

    def __init__(self, devices):
        self._lib = _abi.load()
        ids = (C.c_int * len(devices))(*[int(d) for d in devices])
        h = C.c_void_p()
        rc = self._lib.ptk_group_create(ids, len(devices), C.byref(h))
        if rc != _abi.PTK_OK:
            raise PtkError(rc, self._lib.ptk_group_last_error(None).decode())
        self._h = h
        self.width = self.height = 0

    def close(self):
        if getattr(self, "_h", None):
            self._lib.ptk_group_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc, allow=()):
        if rc != _abi.PTK_OK and rc not in allow:
            raise PtkError(rc, self._lib.ptk_group_last_error(self._h).decode())
        return rc

    def upload(self, scene):
        desc, keep = scene.to_desc()
        self._check(self._lib.ptk_group_scene_upload(self._h, C.byref(desc)))

    def render_begin(self, width, height, fov_y=math.pi / 4.0, max_paths=0):
        self._check(self._lib.ptk_group_render_begin(self._h, width, height, float(np.float32(fov_y)), max_paths))
        self.width, self.height = width, height

    def render_iteration(self, seed, iteration, spi):
        """Blocking iteration; returns True if it ran to the end, False if ptk_group_request_interrupt cut it short."""
        return self._check(self._lib.ptk_group_render_iteration(self._h, seed, iteration, spi), allow=(_abi.PTK_INTERRUPTED,)) == _abi.PTK_OK

    def request_interrupt(self):
        self._check(self._lib.ptk_group_request_interrupt(self._h))

    def progress(self):
        return float(self._lib.ptk_group_progress(self._h))

    def slices(self, spi):
        k = C.c_uint32()
        self._check(self._lib.ptk_group_iteration_slices(self._h, spi, C.byref(k)))
        return k.value

    def submit_slice(self, seed, iteration, spi, slice_index, num_slices, aov=-1):
        t = C.c_uint64()
        self._check(self._lib.ptk_group_submit_slice(self._h, aov, seed, iteration, spi, slice_index, num_slices, C.byref(t)))
        return t.value

    def wait_slice(self, ticket):
        self._check(self._lib.ptk_group_wait_slice(self._h, ticket))

    def finish_iteration(self, iteration, spi, slices_done, num_slices, download=True):
        self._check(self._lib.ptk_group_finish_iteration(self._h, iteration, spi, slices_done, num_slices, 1 if download else 0))

    def wait_iteration(self, iteration, want_image=True):
        img = np.empty((self.height, self.width, 3), np.float32) if want_image else None
        cnt = _abi.Counters()
        self._check(self._lib.ptk_group_wait_iteration(self._h, iteration, img.ctypes.data if want_image else None, C.byref(cnt)))
        return img, (cnt.paths, cnt.rays)

    def read_image(self):
        img = np.empty((self.height, self.width, 3), np.float32)
        self._check(self._lib.ptk_group_read_image(self._h, img.ctypes.data))
        return img

    def counters(self):
        cnt = _abi.Counters()
        self._check(self._lib.ptk_group_get_counters(self._h, C.byref(cnt)))
        return cnt.paths, cnt.rays


class Context:
    """Owns one ptk_ctx (one GPU, one stream)."""

    def __init__(self, device=0):
        self._lib = _abi.load()
        h = C.c_void_p()
        rc = self._lib.ptk_ctx_create(int(device), C.byref(h))
        if rc != _abi.PTK_OK:
            raise PtkError(rc, self._lib.ptk_last_error(None).decode())
        self._h = h
        self.device = int(device)
        self._keep = None
        self.width = self.height = 0

    def close(self):
        if getattr(self, "_h", None):
            self._lib.ptk_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc):
        if rc != _abi.PTK_OK:
            raise PtkError(rc, self._lib.ptk_last_error(self._h).decode())

    def set_stream(self, stream_handle):
        self._check(self._lib.ptk_ctx_set_stream(self._h, C.c_void_p(stream_handle or 0)))

    def synchronize(self):
        self._check(self._lib.ptk_ctx_synchronize(self._h))

    def upload(self, scene):
        desc, keep = scene.to_desc()
        self._check(self._lib.ptk_scene_upload(self._h, C.byref(desc)))
        self._keep = None  # the library copied everything

    def upload_desc(self, desc):
        self._check(self._lib.ptk_scene_upload(self._h, C.byref(desc)))

    def kd_info(self, mesh):
        info = _abi.KdInfo()
        self._check(self._lib.ptk_scene_kd_info(self._h, mesh, C.byref(info)))
        return info

    def intersect_batch(self, origins, directions):
        """Host arrays (n,3) -> dict of numpy outputs (ptk_intersect_batch)."""
        o = np.ascontiguousarray(np.asarray(origins, dtype=np.float32).T)
        d = np.ascontiguousarray(np.asarray(directions, dtype=np.float32).T)
        n = o.shape[1]
        out = {
            "node": np.empty(n, np.int32), "prim": np.empty(n, np.int32), "t": np.empty(n, np.float32),
            "u": np.empty(n, np.float32), "v": np.empty(n, np.float32), "position": np.empty((3, n), np.float32),
            "normal": np.empty((3, n), np.float32), "uv": np.empty((2, n), np.float32),
        }
        self._check(self._lib.ptk_intersect_batch(
            self._h, n, o.ctypes.data, d.ctypes.data, out["node"].ctypes.data, out["prim"].ctypes.data,
            out["t"].ctypes.data, out["u"].ctypes.data, out["v"].ctypes.data, out["position"].ctypes.data,
            out["normal"].ctypes.data, out["uv"].ctypes.data))
        for k in ("position", "normal", "uv"):
            out[k] = np.ascontiguousarray(out[k].T)
        return out

    def intersect_batch_dev(self, n, o_ptr, d_ptr, node=0, prim=0, t=0, u=0, v=0, position=0, normal=0, uv=0):
        self._check(self._lib.ptk_intersect_batch_dev(self._h, n, o_ptr, d_ptr, node, prim, t, u, v, position, normal, uv))

    def render_begin(self, width, height, fov_y=math.pi / 4.0, max_paths=0):
        self._check(self._lib.ptk_render_begin(self._h, width, height, float(np.float32(fov_y)), max_paths))
        self.width, self.height = width, height

    def render_samples_dev(self, seed, sample_begin, sample_count, sum_ptr, accumulate=False, pixel_begin=0, pixel_count=None):
        if pixel_count is None:
            pixel_count = self.width * self.height - pixel_begin
        self._check(self._lib.ptk_render_samples_dev(self._h, seed, pixel_begin, pixel_count, sample_begin, sample_count,
                                                     sum_ptr, 1 if accumulate else 0))

    def render_samples_noalloc_dev(self, seed, sample_begin, sample_count, sum_ptr, accumulate=False, pixel_begin=0, pixel_count=None):
        """ptk_render_samples_noalloc_dev: same as render_samples_dev on the library's static scratch (no allocation)."""
        if pixel_count is None:
            pixel_count = self.width * self.height - pixel_begin
        self._check(self._lib.ptk_render_samples_noalloc_dev(self._h, seed, pixel_begin, pixel_count, sample_begin, sample_count,
                                                             sum_ptr, 1 if accumulate else 0))

    def render_aov_dev(self, aov, seed, sample_begin, sample_count, sum_ptr, accumulate=False, pixel_begin=0, pixel_count=None):
        """ptk_render_aov_dev: debug_pathtracer's AOVs (albedo, emission, roughness, metallic, normal) summed like samples."""
        if pixel_count is None:
            pixel_count = self.width * self.height - pixel_begin
        self._check(self._lib.ptk_render_aov_dev(self._h, aov, seed, pixel_begin, pixel_count, sample_begin, sample_count,
                                                 sum_ptr, 1 if accumulate else 0))

    def render_aov_image(self, aov, seed, spp):
        """(H, W, 3) mean AOV over spp jittered primary rays per pixel (host array)."""
        n = self.width * self.height * 3
        d_sum = self._to_dev(np.zeros(n, np.float32))
        try:
            self.render_aov_dev(aov, seed, 0, spp, d_sum)
            return self._from_dev(d_sum, (self.height, self.width, 3), np.float32) / np.float32(spp)
        finally:
            self._free_dev(d_sum)

    def resolve_iteration_dev(self, sum_ptr, image_ptr, spi, iteration):
        self._check(self._lib.ptk_resolve_iteration_dev(self._h, sum_ptr, image_ptr, spi, iteration))

    def render_image(self, seed, iterations, spi):
        img = np.empty((self.height, self.width, 3), np.float32)
        self._check(self._lib.ptk_render_image(self._h, seed, iterations, spi, img.ctypes.data))
        return img

    def tonemap_dev(self, image_ptr, rgba_ptr, num_pixels):
        self._check(self._lib.ptk_tonemap_dev(self._h, image_ptr, rgba_ptr, num_pixels))

    # ---- probes of single device functions (ptk_probe_*_dev); host arrays in, host arrays out ----
    def _to_dev(self, arr):
        a = np.ascontiguousarray(arr)
        p = C.c_void_p()
        self._check(self._lib.ptk_dev_alloc(self._h, max(1, a.nbytes), C.byref(p)))
        if a.nbytes:
            self._check(self._lib.ptk_dev_upload(self._h, p, a.ctypes.data, a.nbytes))
        return p

    def _from_dev(self, p, shape, dtype):
        out = np.empty(shape, dtype)
        self._check(self._lib.ptk_dev_download(self._h, out.ctypes.data, p, out.nbytes))
        return out

    def _free_dev(self, *ptrs):
        for p in ptrs:
            self._lib.ptk_dev_free(self._h, p)

    def probe_image_sample(self, texture, mode, uv):
        """image::sample of scene texture `texture` at uv (n,2) -> (n,3)."""
        uv = np.asarray(uv, np.float32).reshape(-1, 2)
        n = uv.shape[0]
        d_uv, d_out = self._to_dev(uv.T), self._to_dev(np.zeros((3, n), np.float32))
        try:
            self._check(self._lib.ptk_probe_image_sample_dev(self._h, texture, mode, n, d_uv, d_out))
            return np.ascontiguousarray(self._from_dev(d_out, (3, n), np.float32).T)
        finally:
            self._free_dev(d_uv, d_out)

    def probe_background(self, directions):
        d = np.asarray(directions, np.float32).reshape(-1, 3)
        n = d.shape[0]
        d_d, d_out = self._to_dev(d.T), self._to_dev(np.zeros((3, n), np.float32))
        try:
            self._check(self._lib.ptk_probe_background_dev(self._h, n, d_d, d_out))
            return np.ascontiguousarray(self._from_dev(d_out, (3, n), np.float32).T)
        finally:
            self._free_dev(d_d, d_out)

    def probe_hemisphere(self, normals, u12):
        nr = np.asarray(normals, np.float32).reshape(-1, 3)
        u = np.asarray(u12, np.float32).reshape(-1, 2)
        n = nr.shape[0]
        d_n, d_u, d_out = self._to_dev(nr.T), self._to_dev(u.T), self._to_dev(np.zeros((3, n), np.float32))
        try:
            self._check(self._lib.ptk_probe_hemisphere_dev(self._h, n, d_n, d_u, d_out))
            return np.ascontiguousarray(self._from_dev(d_out, (3, n), np.float32).T)
        finally:
            self._free_dev(d_n, d_u, d_out)

    def probe_shade(self, origins, directions, u12):
        """One level of trace_recursive -> dict(node, emission, weight, next_origin, next_direction)."""
        o = np.asarray(origins, np.float32).reshape(-1, 3)
        d = np.asarray(directions, np.float32).reshape(-1, 3)
        u = np.asarray(u12, np.float32).reshape(-1, 2)
        n = o.shape[0]
        z3 = np.zeros((3, n), np.float32)
        ptrs = [self._to_dev(o.T), self._to_dev(d.T), self._to_dev(u.T), self._to_dev(np.zeros(n, np.int32)),
                self._to_dev(z3), self._to_dev(z3), self._to_dev(z3), self._to_dev(z3)]
        try:
            self._check(self._lib.ptk_probe_shade_dev(self._h, n, *ptrs))
            out = {"node": self._from_dev(ptrs[3], (n,), np.int32)}
            for k, p in zip(("emission", "weight", "next_origin", "next_direction"), ptrs[4:]):
                out[k] = np.ascontiguousarray(self._from_dev(p, (3, n), np.float32).T)
            return out
        finally:
            self._free_dev(*ptrs)

    def counters(self):
        c = _abi.Counters()
        self._check(self._lib.ptk_get_counters(self._h, C.byref(c)))
        return int(c.paths), int(c.rays)

    def set_option(self, option, value):
        self._check(self._lib.ptk_ctx_set_option(self._h, option, value))

    def enable_timing(self, on=True):
        """on = True / 1: every launch is bracketed by HIP events; n > 1: the launches of every n-th render_samples_dev call."""
        self._check(self._lib.ptk_ctx_enable_timing(self._h, int(on)))

    def timing(self):
        """{kernel name: (launches, total_ms)} since enable_timing / the last call."""
        arr = (_abi.KernelTiming * len(_abi.PTK_KERNEL_NAMES))()
        self._check(self._lib.ptk_get_timing(self._h, C.cast(arr, C.c_void_p)))
        return {n: (int(arr[i].launches), float(arr[i].total_ms)) for i, n in enumerate(_abi.PTK_KERNEL_NAMES)}

    def reset_counters(self):
        self._check(self._lib.ptk_reset_counters(self._h))

    def kd_build(self, triangles, bounds=None, root_depth=0):
        """ptk_kd_build_dev: the KD build on this context's GPU -> (KdInfo, nodes ctypes array, prims ndarray),
        byte-identical to kd_build() below."""
        tris = np.ascontiguousarray(triangles, dtype=np.float32).reshape(-1, 3, 8)
        b = None if bounds is None else np.ascontiguousarray(bounds, dtype=np.float32).reshape(6)
        bp = None if b is None else b.ctypes.data
        info = _abi.KdInfo()
        self._check(self._lib.ptk_kd_build_dev(self._h, tris.ctypes.data, tris.shape[0], bp, root_depth, C.byref(info), None, 0, None, 0))
        nodes = (_abi.KdNode * max(1, info.num_nodes))()
        prims = np.empty(max(1, info.num_leaf_prims), np.uint32)
        self._check(self._lib.ptk_kd_build_dev(self._h, tris.ctypes.data, tris.shape[0], bp, root_depth, C.byref(info),
                                               C.cast(nodes, C.c_void_p), info.num_nodes, prims.ctypes.data, info.num_leaf_prims))
        return info, nodes, prims[: info.num_leaf_prims]


def kd_build(triangles):
    """ptk_kd_build on a (T,3,8) float32 array -> (KdInfo, nodes ctypes array, prims ndarray). Host only."""
    lib = _abi.load()
    tris = np.ascontiguousarray(triangles, dtype=np.float32).reshape(-1, 3, 8)
    info = _abi.KdInfo()
    rc = lib.ptk_kd_build(tris.ctypes.data, tris.shape[0], C.byref(info), None, 0, None, 0)
    if rc != _abi.PTK_OK:
        raise PtkError(rc, "ptk_kd_build")
    nodes = (_abi.KdNode * max(1, info.num_nodes))()
    prims = np.empty(max(1, info.num_leaf_prims), np.uint32)
    rc = lib.ptk_kd_build(tris.ctypes.data, tris.shape[0], C.byref(info), C.cast(nodes, C.c_void_p), info.num_nodes,
                          prims.ctypes.data, info.num_leaf_prims)
    if rc != _abi.PTK_OK:
        raise PtkError(rc, "ptk_kd_build")
    return info, nodes, prims[: info.num_leaf_prims]


# ---- reference-shaped driver -------------------------------------------------

class ViewParameters:
    """rt::view_parameters, abstract_pathtracer.h:90-95."""

    def __init__(self, width=800, height=600, fov_y=math.pi / 4.0):
        self.width, self.height, self.fov_y = width, height, fov_y


class TraceParameters:
    """rt::trace_parameters, abstract_pathtracer.h:100-105.  num_threads is accepted and
    ignored (the GPU replaces the scan-line thread pool)."""

    def __init__(self, num_threads=4, iterations=1, samples_per_iteration=1):
        self.num_threads, self.iterations, self.samples_per_iteration = num_threads, iterations, samples_per_iteration


class EventEmitter:
    """rt::event_emitter, abstract_pathtracer.h:27-85."""

    def __init__(self):
        self._handlers = []

    def subscribe(self, fn):
        self._handlers.append(fn)

    def emit(self, *args):
        for h in list(self._handlers):
            h(*args)

    __call__ = emit


class PathtracerResult:
    """rt::pathtracer_result, abstract_pathtracer.h:111-174: the driver thread starts in
    the constructor; events fire on that thread."""

    def __init__(self, fn):
        self.progress = 0.0
        self.iteration = 0
        self.samples_per_pixel = 0
        self.on_iteration_start = EventEmitter()
        self.on_iteration_end = EventEmitter()
        self.on_end = EventEmitter()
        self.error = None
        self._interrupted = False
        self._start_time = time.time()

        def body():
            try:
                fn(self)
            except Exception as e:  # surfaced by wait()
                self.error = e

        self._thread = threading.Thread(target=body, daemon=True)
        self._thread.start()

    def wait(self):
        self._thread.join()
        if self.error is not None:
            raise self.error

    def interrupt(self):
        self._interrupted = True

    def is_interrupted(self):
        return self._interrupted

    def get_elapsed_time(self):
        return time.time() - self._start_time


class Pathtracer:
    """rt::pathtracer (pathtracer.h:13-22) driven like abstract_pathtracer::run
    (abstract_pathtracer.cpp:11-119): iterations==0 runs until interrupt();
    on_iteration_end and on_end still fire after an interrupt."""

    def __init__(self, device=0, seed=0, max_paths=0):
        self.device, self.seed, self.max_paths = device, seed, max_paths

    def run(self, view_params, trace_params, scene):
        import torch  # device memory plumbing only

        ctx = Context(self.device)
        ctx.upload(scene)  # scene.compile(), abstract_pathtracer.cpp:13 (caller thread)
        w, h, fov = view_params.width, view_params.height, view_params.fov_y
        iterations, spi = trace_params.iterations, trace_params.samples_per_iteration
        seed = self.seed

        def driver(self_):
            dev = torch.device("cuda", self.device)
            with torch.cuda.device(dev):
                stream = torch.cuda.current_stream(dev)
                ctx.set_stream(stream.cuda_stream)
                ctx.render_begin(w, h, fov, self.max_paths)
                image = torch.zeros(h, w, 3, dtype=torch.float32, device=dev)
                total = torch.empty(h, w, 3, dtype=torch.float32, device=dev)
                it = 0
                while (iterations == 0 or it < iterations) and not self_.is_interrupted():
                    self_.on_iteration_start(it)
                    ctx.render_samples_dev(seed, it * spi, spi, total.data_ptr())
                    ctx.resolve_iteration_dev(total.data_ptr(), image.data_ptr(), spi, it)
                    stream.synchronize()
                    self_.progress = 1.0
                    self_.iteration = it
                    self_.samples_per_pixel += spi
                    self_.on_iteration_end(image.cpu().numpy(), it)
                    it += 1
                self_.on_end(image.cpu().numpy())
            ctx.close()

        return PathtracerResult(driver)
