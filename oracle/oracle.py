"""ctypes front end of the CPU oracle (oracle/rt_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
It reuses the product's host-side packer so both sides see byte-identical scene arrays.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from raytracer_amd.packing import pack_scene, pack_camera, pack_surfaces, rows_in_part, PackedScene

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "librt_oracle.so")

JIT_PIXEL_KEYED, JIT_NUMPY_STREAM = 0, 1


class _Scene(C.Structure):
    _fields_ = [("n_surfaces", C.c_int), ("surf_type", C.c_void_p), ("surf_data", C.c_void_p),
                ("surf_material", C.c_void_p), ("n_materials", C.c_int), ("materials", C.c_void_p),
                ("n_lights", C.c_int), ("lights", C.c_void_p), ("background", C.c_double * 3)]


class _Camera(C.Structure):
    _fields_ = [("v", C.c_double * 15)]


class Stats(C.Structure):
    _fields_ = [("rays", C.c_longlong), ("hits", C.c_longlong), ("shadow_rays", C.c_longlong),
                ("segments", C.c_longlong), ("batches", C.c_int)]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


def build(force=False):
    """Compile librt_oracle.so with gcc if it is missing or older than its source."""
    src = os.path.join(_HERE, "rt_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_render.restype = C.c_int
        _lib.orc_render_sequential.restype = C.c_int
    return _lib


def set_threads(n):
    lib().orc_set_threads(int(n))


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _scene_struct(ps: PackedScene):
    s = _Scene()
    s.n_surfaces = len(ps.surf_type); s.surf_type = _ptr(ps.surf_type)
    s.surf_data = _ptr(ps.surf_data); s.surf_material = _ptr(ps.surf_material)
    s.n_materials = len(ps.materials); s.materials = _ptr(ps.materials)
    s.n_lights = len(ps.lights); s.lights = _ptr(ps.lights)
    s.background[:] = list(ps.background)
    return s


def _rays(origins, dirs):
    o = np.ascontiguousarray(origins, dtype=np.float64).reshape(-1, 3)
    d = np.ascontiguousarray(dirs, dtype=np.float64).reshape(-1, 3)
    assert o.shape == d.shape
    return o, d


def render(camera, scene_settings, materials, surfaces, lights, width, height, *, jitter=("pixel", 1234),
           block_rows=1, part=0, n_parts=1, return_stats=False, sequential=False):
    """Oracle counterpart of the reference's render_vectorized (ray_tracer.py:437-604), or with
    sequential=True of its per-pixel recursive render (ray_tracer.py:398-434)."""
    from raytracer_amd.jitter import as_jitter
    jit = as_jitter(jitter)
    camera.setup(width / height)
    ps = pack_scene(materials, surfaces, lights, scene_settings.background_color)
    sc = _scene_struct(ps)
    cam = _Camera(); cam.v[:] = list(pack_camera(camera))
    nrows = len(rows_in_part(height, block_rows, part, n_parts))
    out = np.empty((nrows, width, 3), dtype=np.float64)
    st = Stats()
    fn = lib().orc_render_sequential if sequential else lib().orc_render
    rc = fn(C.byref(sc), C.byref(cam), int(width), int(height),
                          int(scene_settings.root_number_shadow_rays), int(scene_settings.max_recursions),
                          int(jit.mode), C.c_uint32(jit.seed & 0xFFFFFFFF), int(block_rows), int(part), int(n_parts),
                          _ptr(out), C.byref(st))
    if rc != 0:
        raise RuntimeError(f"orc_render failed ({rc})")
    return (out, st.as_dict()) if return_stats else out


def intersect(surface, origins, dirs):
    """Oracle counterpart of surface.intersect_batch (sphere.py:45, infinite_plane.py:26, cube.py:64)."""
    o, d = _rays(origins, dirs)
    st, sd, _ = pack_surfaces([surface])
    t = np.empty(len(o)); n = np.empty((len(o), 3))
    lib().orc_intersect(int(st[0]), _ptr(sd), _ptr(o), _ptr(d), C.c_longlong(len(o)), _ptr(t), _ptr(n))
    return t, n


def nearest_hit(surfaces, origins, dirs):
    """Oracle counterpart of find_nearest_intersection_batch (ray_tracer.py:112-127)."""
    o, d = _rays(origins, dirs)
    st, sd, sm = pack_surfaces(surfaces)
    ps = PackedScene(st, sd, sm, np.zeros((1, 11)), np.zeros((0, 9)), np.zeros(3))
    sc = _scene_struct(ps)
    t = np.empty(len(o)); idx = np.empty(len(o), dtype=np.int32); n = np.empty((len(o), 3))
    lib().orc_nearest_hit(C.byref(sc), _ptr(o), _ptr(d), C.c_longlong(len(o)), _ptr(t), _ptr(idx), _ptr(n))
    return t, idx, n


def shadow_transmission(materials, surfaces, origins, dirs, max_distances):
    """Oracle counterpart of compute_shadow_transmission_batch (ray_tracer.py:247-293)."""
    o, d = _rays(origins, dirs)
    md = np.ascontiguousarray(max_distances, dtype=np.float64)
    ps = pack_scene(materials, surfaces, [], np.zeros(3))
    sc = _scene_struct(ps)
    out = np.empty(len(o))
    lib().orc_shadow_transmission(C.byref(sc), _ptr(o), _ptr(d), _ptr(md), C.c_longlong(len(o)), _ptr(out))
    return out


def camera_rays(camera, width, height, y0=0, y1=None):
    """Oracle counterpart of Camera.generate_rays_for_rows (camera.py:75-104)."""
    y1 = height if y1 is None else y1
    cam = _Camera(); cam.v[:] = list(pack_camera(camera))
    n = (y1 - y0) * width
    o = np.empty((n, 3)); d = np.empty((n, 3))
    lib().orc_camera_rays(C.byref(cam), int(width), int(height), int(y0), int(y1), _ptr(o), _ptr(d))
    return o, d


def mt_random(seed, n):
    out = np.empty(n)
    lib().orc_mt_random(C.c_uint32(seed), C.c_longlong(n), _ptr(out))
    return out


def jitter_cell(seed, pixels, depth, light, cell):
    p = np.ascontiguousarray(pixels, dtype=np.uint32)
    ui = np.empty(len(p)); uj = np.empty(len(p))
    lib().orc_jitter_cell(C.c_uint32(seed), _ptr(p), C.c_longlong(len(p)), int(depth), int(light), int(cell),
                          _ptr(ui), _ptr(uj))
    return ui, uj
