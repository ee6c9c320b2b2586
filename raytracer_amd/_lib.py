"""ctypes binding of librt_hip.so (include/rt_hip.h).  Fails loudly when the library is missing or
no GPU is visible: there is no CPU fallback in this package."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RT_HIP_LIBRARY") or os.path.join(_HERE, "librt_hip.so")   # RT_HIP_LIBRARY: e.g. the audit build

RT_OK = 0
RT_ERR_QUEUE_OVERFLOW = -4
RT_FP64, RT_FP32 = 0, 1
RT_OUT_F64, RT_OUT_F32, RT_OUT_U8 = 0, 1, 2
RT_FLAG_ASYNC, RT_FLAG_TIMING, RT_FLAG_SEQUENTIAL, RT_FLAG_BRUTE_FORCE = 1, 2, 4, 8
RT_MAX_LEVELS = 32
RT_STATIC_MAX_PIXELS = 131072

EXPORTS = ["rt_last_error", "rt_device_count", "rt_scene_create", "rt_scene_destroy", "rt_scene_set_stream",
           "rt_sync", "rt_rows_in_part", "rt_render", "rt_render_static", "rt_render_multi", "rt_host_alloc", "rt_host_free", "rt_nearest_hit", "rt_shadow_transmission", "rt_camera_rays"]


class SceneDesc(C.Structure):
    _fields_ = [("n_surfaces", C.c_int32), ("surf_type", C.c_void_p), ("surf_data", C.c_void_p),
                ("surf_material", C.c_void_p), ("n_materials", C.c_int32), ("materials", C.c_void_p),
                ("n_lights", C.c_int32), ("lights", C.c_void_p), ("background", C.c_double * 3)]


class CameraDesc(C.Structure):
    _fields_ = [("pos", C.c_double * 3), ("forward", C.c_double * 3), ("right", C.c_double * 3),
                ("up", C.c_double * 3), ("screen_distance", C.c_double), ("screen_width", C.c_double),
                ("screen_height", C.c_double)]


class Frame(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("block_rows", C.c_int32), ("part", C.c_int32),
                ("n_parts", C.c_int32), ("n_shadow", C.c_int32), ("max_depth", C.c_int32), ("precision", C.c_int32),
                ("jitter_mode", C.c_int32), ("jitter_seed", C.c_uint32), ("out_format", C.c_int32),
                ("flags", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("rays", C.c_int64), ("hits", C.c_int64), ("shadow_rays", C.c_int64), ("segments", C.c_int64),
                ("batches", C.c_int32), ("levels", C.c_int32),
                ("level_rays", C.c_int64 * RT_MAX_LEVELS), ("level_hits", C.c_int64 * RT_MAX_LEVELS),
                ("trace_ms", C.c_float), ("shade_ms", C.c_float), ("other_ms", C.c_float),
                ("trace_launches", C.c_int32), ("shade_launches", C.c_int32), ("total_ms", C.c_float),
                ("retested", C.c_int64)]

    def as_dict(self):
        d = {k: getattr(self, k) for k in ("rays", "hits", "shadow_rays", "segments", "batches", "levels",
                                           "trace_ms", "shade_ms", "other_ms", "trace_launches",
                                           "shade_launches", "total_ms", "retested")}
        d["level_rays"] = list(self.level_rays[: self.levels])
        d["level_hits"] = list(self.level_hits[: self.levels])
        return d


class RtError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"librt_hip error {code}: {message}")
        self.code = code


_lib = None


def load():
    """Load librt_hip.so once; raises RuntimeError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           "or `make -C raytracer_amd/csrc` (there is no CPU fallback)")
    lib = C.CDLL(LIB_PATH)
    lib.rt_last_error.restype = C.c_char_p
    lib.rt_scene_destroy.restype = None
    for name in EXPORTS:
        getattr(lib, name)   # AttributeError here = header and library disagree
    lib.rt_scene_create.argtypes = [C.POINTER(SceneDesc), C.c_int, C.POINTER(C.c_void_p)]
    lib.rt_scene_destroy.argtypes = [C.c_void_p]
    lib.rt_scene_set_stream.argtypes = [C.c_void_p, C.c_void_p]
    lib.rt_sync.argtypes = [C.c_void_p]
    lib.rt_render.argtypes = [C.c_void_p, C.POINTER(CameraDesc), C.POINTER(Frame), C.c_void_p, C.c_int,
                              C.POINTER(Stats)]
    lib.rt_render_static.argtypes = [C.c_void_p, C.POINTER(CameraDesc), C.POINTER(Frame), C.c_void_p, C.POINTER(Stats)]
    lib.rt_render_multi.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(CameraDesc), C.POINTER(Frame), C.c_void_p, C.c_int,
                                    C.POINTER(Stats)]
    lib.rt_host_alloc.argtypes = [C.c_size_t, C.POINTER(C.c_void_p)]
    lib.rt_host_free.argtypes = [C.c_void_p]
    lib.rt_host_free.restype = None
    lib.rt_nearest_hit.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.rt_shadow_transmission.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    lib.rt_camera_rays.argtypes = [C.c_void_p, C.POINTER(CameraDesc), C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                   C.c_void_p, C.c_void_p]
    lib.rt_rows_in_part.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32)]
    _lib = lib
    return lib


def check(rc):
    if rc != RT_OK:
        raise RtError(rc, load().rt_last_error().decode("utf-8", "replace"))


def device_count() -> int:
    n = C.c_int(0)
    check(load().rt_device_count(C.byref(n)))
    return n.value
