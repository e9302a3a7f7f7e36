"""ctypes mirror of include/ptk.h and the loader for libptk.so.

The product path is the HIP library: if libptk.so is missing this module raises —
there is no CPU fallback (the CPU oracle under oracle/ is test infrastructure and
is never imported from here).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# PTK_LIB: an experiment build of the same library (scripts/), never a different implementation
LIB_PATH = os.environ.get("PTK_LIB") or os.path.join(_HERE, "libptk.so")

PTK_OK = 0
PTK_INTERRUPTED = 5
PTK_ERR_INVALID, PTK_ERR_HIP, PTK_ERR_STATE, PTK_ERR_NOMEM = 1, 2, 3, 4
PTK_MAX_DEPTH = 5
PTK_AOV_ALBEDO, PTK_AOV_EMISSION, PTK_AOV_ROUGHNESS, PTK_AOV_METALLIC, PTK_AOV_NORMAL = range(5)
PTK_NOALLOC_MAX_PATHS = 1 << 18
PTK_SAMPLER_COLOR, PTK_SAMPLER_IMAGE = 0, 1
PTK_MODE_LINEAR, PTK_MODE_NEAREST = 0, 1
PTK_SHAPE_SPHERE, PTK_SHAPE_MESH = 0, 1
PTK_BACKGROUND_COLOR, PTK_BACKGROUND_EQUIRECT = 0, 1


class Vertex(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("normal", C.c_float * 3), ("uv", C.c_float * 2)]


class Triangle(C.Structure):
    _fields_ = [("v", Vertex * 3)]


class MeshDesc(C.Structure):
    _fields_ = [("triangles", C.c_void_p), ("num_triangles", C.c_uint64)]


class TextureDesc(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("rgb", C.c_void_p)]


class SamplerDesc(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("color", C.c_float * 3), ("texture", C.c_uint32), ("mode", C.c_uint32)]


class NodeDesc(C.Structure):
    _fields_ = [
        ("transform", C.c_float * 16),
        ("inv_transform", C.c_float * 16),
        ("normal_transform", C.c_float * 16),
        ("shape_kind", C.c_uint32),
        ("mesh", C.c_uint32),
        ("albedo", C.c_uint32),
        ("emission", C.c_uint32),
        ("roughness", C.c_uint32),
        ("metallic", C.c_uint32),
    ]


class BackgroundDesc(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("color", C.c_float * 3), ("texture", C.c_uint32), ("mode", C.c_uint32)]


class CameraDesc(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("direction", C.c_float * 3)]


class SceneDesc(C.Structure):
    _fields_ = [
        ("nodes", C.c_void_p),
        ("num_nodes", C.c_uint32),
        ("meshes", C.c_void_p),
        ("num_meshes", C.c_uint32),
        ("samplers", C.c_void_p),
        ("num_samplers", C.c_uint32),
        ("textures", C.c_void_p),
        ("num_textures", C.c_uint32),
        ("background", BackgroundDesc),
        ("camera", CameraDesc),
    ]


class KdNode(C.Structure):
    _fields_ = [
        ("bmin", C.c_float * 3),
        ("bmax", C.c_float * 3),
        ("skip", C.c_uint32),
        ("first_prim", C.c_uint32),
        ("num_prims", C.c_uint32),
        ("depth", C.c_uint32),
        ("tmin", C.c_float * 3),
        ("tmax", C.c_float * 3),
    ]


class KdInfo(C.Structure):
    _fields_ = [
        ("num_nodes", C.c_uint32),
        ("num_leaves", C.c_uint32),
        ("num_leaf_prims", C.c_uint64),
        ("max_depth", C.c_uint32),
        ("bmin", C.c_float * 3),
        ("bmax", C.c_float * 3),
    ]


class Counters(C.Structure):
    _fields_ = [("paths", C.c_uint64), ("rays", C.c_uint64)]


class KernelTiming(C.Structure):
    _fields_ = [("launches", C.c_uint64), ("total_ms", C.c_double)]


PTK_OPT_TRAVERSAL = 1
PTK_OPT_KD_PRUNE = 2
PTK_OPT_NODE_CULL = 3
PTK_OPT_BALANCE = 4
PTK_OPT_SPLIT = 5
PTK_OPT_DEVICE_KD = 6
PTK_OPT_WALK_STACK = 7
PTK_OPT_COMPACT = 8
PTK_OPT_FUSE = 9
PTK_OPT_OVERLAP = 10
PTK_KERNEL_NAMES = ["k_bounce<true>", "k_bounce<false>", "k_accumulate", "k_intersect_batch", "k_walk", "k_shade_parked", "k_bounce_fused"]

# name -> (restype, argtypes); every symbol include/ptk.h declares.
SYMBOLS = {
    "ptk_ctx_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "ptk_ctx_destroy": (None, [C.c_void_p]),
    "ptk_last_error": (C.c_char_p, [C.c_void_p]),
    "ptk_ctx_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ptk_ctx_synchronize": (C.c_int, [C.c_void_p]),
    "ptk_ctx_get_stream": (C.c_void_p, [C.c_void_p]),
    "ptk_request_interrupt": (C.c_int, [C.c_void_p]),
    "ptk_resolve_range_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64]),
    "ptk_counters_download_async": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ptk_group_request_interrupt": (C.c_int, [C.c_void_p]),
    "ptk_group_progress": (C.c_float, [C.c_void_p]),
    "ptk_group_iteration_slices": (C.c_int, [C.c_void_p, C.c_uint64, C.POINTER(C.c_uint32)]),
    "ptk_group_submit_slice": (C.c_int, [C.c_void_p, C.c_int, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64)]),
    "ptk_group_wait_slice": (C.c_int, [C.c_void_p, C.c_uint64]),
    "ptk_group_finish_iteration": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int]),
    "ptk_group_wait_iteration": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p, C.POINTER(Counters)]),
    "ptk_scene_upload": (C.c_int, [C.c_void_p, C.POINTER(SceneDesc)]),
    "ptk_scene_set_camera": (C.c_int, [C.c_void_p, C.POINTER(CameraDesc)]),
    "ptk_kd_build": (C.c_int, [C.c_void_p, C.c_uint64, C.POINTER(KdInfo), C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64]),
    "ptk_kd_build_bounded": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, C.POINTER(KdInfo), C.c_void_p, C.c_uint64,
                                      C.c_void_p, C.c_uint64]),
    "ptk_kd_build_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, C.POINTER(KdInfo), C.c_void_p, C.c_uint64,
                                  C.c_void_p, C.c_uint64]),
    "ptk_intersect_batch_dev": (C.c_int, [C.c_void_p, C.c_uint64] + [C.c_void_p] * 10),
    "ptk_intersect_batch": (C.c_int, [C.c_void_p, C.c_uint64] + [C.c_void_p] * 10),
    "ptk_render_begin": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_float, C.c_uint64]),
    "ptk_render_samples_dev": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_void_p, C.c_int]),
    "ptk_render_samples_noalloc_dev": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_void_p, C.c_int]),
    "ptk_render_aov_dev": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_void_p, C.c_int]),
    "ptk_resolve_iteration_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64]),
    "ptk_render_image": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_void_p]),
    "ptk_tonemap_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]),
    "ptk_probe_image_sample_dev": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p]),
    "ptk_probe_background_dev": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]),
    "ptk_probe_hemisphere_dev": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ptk_probe_shade_dev": (C.c_int, [C.c_void_p, C.c_uint64] + [C.c_void_p] * 8),
    "ptk_get_counters": (C.c_int, [C.c_void_p, C.POINTER(Counters)]),
    "ptk_reset_counters": (C.c_int, [C.c_void_p]),
    "ptk_ctx_set_option": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "ptk_ctx_enable_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "ptk_get_timing": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ptk_dev_alloc": (C.c_int, [C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p)]),
    "ptk_dev_free": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ptk_dev_memset": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_uint64]),
    "ptk_dev_download": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]),
    "ptk_dev_upload": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]),
    "ptk_group_create": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]),
    "ptk_group_destroy": (None, [C.c_void_p]),
    "ptk_group_last_error": (C.c_char_p, [C.c_void_p]),
    "ptk_group_size": (C.c_int, [C.c_void_p]),
    "ptk_group_ctx": (C.c_void_p, [C.c_void_p, C.c_int]),
    "ptk_group_scene_upload": (C.c_int, [C.c_void_p, C.POINTER(SceneDesc)]),
    "ptk_group_scene_set_camera": (C.c_int, [C.c_void_p, C.POINTER(CameraDesc)]),
    "ptk_group_render_begin": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_float, C.c_uint64]),
    "ptk_group_render_iteration": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64]),
    "ptk_group_render_aov_iteration": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint64]),
    "ptk_group_read_image": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ptk_group_get_counters": (C.c_int, [C.c_void_p, C.POINTER(Counters)]),
    "ptk_scene_kd_info": (C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(KdInfo)]),
}

_lib = None


def load():
    """Loads libptk.so (once). Raises RuntimeError if the HIP extension was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()'); there is no CPU fallback"
        )
    # torch (used by the harness for device buffers, streams and torch.distributed) ships
    # its own libamdhip64.so.7; load it first so libptk.so binds to that same HIP runtime
    # instead of bringing a second one into the process (two runtimes cannot share a GPU).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
