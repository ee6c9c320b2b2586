"""Host side of the drop-in boundary: `render_hip` has the signature and return value of the
reference's `render_vectorized` (ray_tracer.py:437-604) and runs it on the GPU through the C-ABI
of librt_hip.so.  Also the GPU-backed counterparts of the reference helpers that unit tests pin:
`nearest_hit` (find_nearest_intersection_batch, ray_tracer.py:112-127), `shadow_transmission`
(compute_shadow_transmission_batch, :247-293) and `camera_rays` (camera.py:75-104).
"""
from __future__ import annotations

import ctypes as C
import time

import numpy as np

from . import _lib
from .jitter import as_jitter
from .packing import PackedScene, pack_camera, pack_scene, pack_surfaces, rows_in_part

_PRECISION = {"fp64": _lib.RT_FP64, "fp32": _lib.RT_FP32, "f64": _lib.RT_FP64, "f32": _lib.RT_FP32}
_OUT = {"f64": (_lib.RT_OUT_F64, np.float64), "f32": (_lib.RT_OUT_F32, np.float32), "u8": (_lib.RT_OUT_U8, np.uint8)}


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class _PinnedPool:
    """Page-locked host buffers for output frames (rt_host_alloc).  `empty(shape, dtype)` returns an ndarray backed by
    a pooled block; the block goes back to the pool when the array (and every view of it) has been garbage-collected, so
    a caller that drops the previous frame before asking for the next one reuses the same pinned memory, and a caller
    that keeps frames simply gets new blocks.  Device -> host copies into these arrays are asynchronous and run at PCIe
    speed; into ordinary (pageable) NumPy arrays the runtime stages them at a fraction of that."""

    def __init__(self):
        self._free = {}          # nbytes -> [address, ...]
        self.allocated = 0

    def empty(self, shape, dtype):
        import weakref
        lib = _lib.load()
        dtype = np.dtype(dtype)
        nbytes = max(int(np.prod(shape)) * dtype.itemsize, 1)
        size = 1 << (nbytes - 1).bit_length() if nbytes > (1 << 16) else (1 << 16)     # few size classes
        stack = self._free.setdefault(size, [])
        if stack:
            addr = stack.pop()
        else:
            p = C.c_void_p()
            _lib.check(lib.rt_host_alloc(size, C.byref(p)))
            addr = p.value
            self.allocated += size
        buf = (C.c_ubyte * size).from_address(addr)
        weakref.finalize(buf, stack.append, addr)        # runs when the last array built on `buf` is gone
        return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    def trim(self):
        lib = _lib.load()
        for stack in self._free.values():
            while stack:
                lib.rt_host_free(C.c_void_p(stack.pop()))


pinned = _PinnedPool()


def pinned_empty(shape, dtype=np.float32):
    """An uninitialised ndarray in page-locked host memory (see _PinnedPool); pass it as `out=` to GpuScene.render."""
    return pinned.empty(shape, dtype)


class GpuScene:
    """Device copy of a packed scene (rt_scene handle).  Reusable across frames; owns a HIP stream
    and the render workspace.  Use as a context manager or call close()."""

    def __init__(self, packed: PackedScene, device: int = 0):
        self._lib = _lib.load()
        self.packed = packed
        self.device = int(device)
        d = _lib.SceneDesc()
        d.n_surfaces = len(packed.surf_type); d.surf_type = _ptr(packed.surf_type)
        d.surf_data = _ptr(packed.surf_data); d.surf_material = _ptr(packed.surf_material)
        d.n_materials = len(packed.materials); d.materials = _ptr(packed.materials)
        d.n_lights = len(packed.lights); d.lights = _ptr(packed.lights)
        d.background[:] = list(packed.background)
        h = C.c_void_p()
        _lib.check(self._lib.rt_scene_create(C.byref(d), self.device, C.byref(h)))
        self._h = h
        self._keepalive = []   # rt_stats of async frames: the library writes them at the next sync

    @classmethod
    def from_objects(cls, materials, surfaces, lights, background_color, device=0):
        return cls(pack_scene(materials, surfaces, lights, background_color), device)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.rt_scene_destroy(self._h)   # synchronises the stream first
            self._h = None
            self._keepalive = []

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def set_stream(self, hip_stream_ptr):
        _lib.check(self._lib.rt_scene_set_stream(self._h, C.c_void_p(hip_stream_ptr or 0)))

    def sync(self):
        try:
            _lib.check(self._lib.rt_sync(self._h))
        finally:
            self._keepalive.clear()

    # -- the hot path -------------------------------------------------------------------------
    def render(self, camera, width, height, n_shadow, max_depth, *, precision="fp64", jitter=None,
               block_rows=1, part=0, n_parts=1, out_format=None, out_ptr=None, flags=0, stats=None, static=False,
               out=None):
        """Render a row set.  With `out_ptr` (a device pointer, e.g. torch_tensor.data_ptr()) the
        result stays on the GPU; otherwise a NumPy array (rows, width, 3) is returned.
        static=True (needs out_ptr) goes through rt_render_static, the entry point that allocates nothing.
        `out`: a C-contiguous host array (rows, width, 3) of the output format to fill instead of a new pinned one."""
        jit = as_jitter(jitter)
        if camera.forward is None:
            camera.setup(width / height)
        cam = _lib.CameraDesc()
        v = pack_camera(camera)
        cam.pos[:] = v[0:3]; cam.forward[:] = v[3:6]; cam.right[:] = v[6:9]; cam.up[:] = v[9:12]
        cam.screen_distance, cam.screen_width, cam.screen_height = v[12], v[13], v[14]
        prec = _PRECISION[precision]
        if out_format is None:
            out_format = "f64" if prec == _lib.RT_FP64 else "f32"
        fmt, dtype = _OUT[out_format]
        fr = _lib.Frame(int(width), int(height), int(block_rows), int(part), int(n_parts), int(n_shadow),
                        int(max_depth), prec, int(jit.mode), jit.seed & 0xFFFFFFFF, fmt, int(flags))
        st = stats if stats is not None else _lib.Stats()
        if static:
            if out_ptr is None:
                raise ValueError("static=True renders into device memory: pass out_ptr")
            _lib.check(self._lib.rt_render_static(self._h, C.byref(cam), C.byref(fr), C.c_void_p(out_ptr),
                                                  None if (flags & _lib.RT_FLAG_ASYNC) else C.byref(st)))
            self.last_stats = st.as_dict()
            return None
        if out_ptr is not None:
            if flags & _lib.RT_FLAG_ASYNC:
                self._keepalive.append(st)
            _lib.check(self._lib.rt_render(self._h, C.byref(cam), C.byref(fr), C.c_void_p(out_ptr), 1, C.byref(st)))
            if not (flags & _lib.RT_FLAG_ASYNC):
                self.last_stats = st.as_dict()
            return None
        rows = len(rows_in_part(height, block_rows, part, n_parts))
        if out is None:
            out = pinned.empty((rows, width, 3), dtype)
        elif out.shape != (rows, width, 3) or out.dtype != dtype or not out.flags.c_contiguous:
            raise ValueError(f"out must be a C-contiguous {np.dtype(dtype).name} array of shape {(rows, width, 3)}")
        _lib.check(self._lib.rt_render(self._h, C.byref(cam), C.byref(fr), _ptr(out), 0, C.byref(st)))
        self.last_stats = st.as_dict()
        return out

    # -- helpers pinned by unit tests ---------------------------------------------------------
    def nearest_hit(self, origins, dirs):
        o, d = _rays(origins, dirs)
        t = np.empty(len(o)); idx = np.empty(len(o), dtype=np.int32); n = np.empty((len(o), 3))
        _lib.check(self._lib.rt_nearest_hit(self._h, _ptr(o), _ptr(d), len(o), _ptr(t), _ptr(idx), _ptr(n)))
        return t, idx, n

    def shadow_transmission(self, origins, dirs, max_distances):
        o, d = _rays(origins, dirs)
        md = np.ascontiguousarray(max_distances, dtype=np.float64)
        if md.shape != (len(o),):
            raise ValueError("max_distances must have one entry per ray")
        out = np.empty(len(o))
        _lib.check(self._lib.rt_shadow_transmission(self._h, _ptr(o), _ptr(d), _ptr(md), len(o), _ptr(out)))
        return out


def _rays(origins, dirs):
    o = np.ascontiguousarray(origins, dtype=np.float64).reshape(-1, 3)
    d = np.ascontiguousarray(dirs, dtype=np.float64).reshape(-1, 3)
    if o.shape != d.shape:
        raise ValueError(f"origins {o.shape} and directions {d.shape} differ")
    return o, d


_scene_cache = {}       # (device, digest of the packed arrays) -> GpuScene; a couple of entries, oldest dropped first


def _cached_scene(materials, surfaces, lights, background_color, device):
    """The GpuScene of these objects, kept across render_hip calls: uploading the scene and growing the render
    workspace (GBs of queues at 4K) costs far more than a frame.  Keyed by the CONTENT of the packed arrays, so edited
    objects get a new device copy."""
    import hashlib
    packed = pack_scene(materials, surfaces, lights, background_color)
    h = hashlib.blake2b(digest_size=16)
    for a in (packed.surf_type, packed.surf_data, packed.surf_material, packed.materials, packed.lights,
              np.asarray(packed.background, dtype=np.float64)):
        h.update(np.ascontiguousarray(a).tobytes()); h.update(b"|")
    import os
    # (the A/B switches rt_scene_create reads from the environment are part of what the handle is)
    env = tuple(os.environ.get(k) for k in ("RT_ACCEL", "RT_BVH_LEAF", "RT_BVH_FANOUT", "RT_BVH_ORDER", "RT_SHADE", "RT_SORT_HITS"))
    key = (int(device), h.digest(), env)
    gs = _scene_cache.get(key)
    if gs is None or getattr(gs, "_h", None) is None:
        while len(_scene_cache) >= 2:
            _scene_cache.pop(next(iter(_scene_cache))).close()
        gs = _scene_cache[key] = GpuScene(packed, device)
    return gs


def clear_scene_cache():
    while _scene_cache:
        _scene_cache.pop(next(iter(_scene_cache))).close()


def render_hip(camera, scene_settings, materials, surfaces, lights, width, height, *,
               precision="fp64", jitter=None, device=0, stats=None, verbose=True, sequential=False, out_dtype=np.float64):
    """Drop-in for the reference's render_vectorized: returns (height, width, 3) float64 unclipped
    linear radiance and, like the reference, calls camera.setup(width / height) first.

    Keyword-only extras: precision "fp64" (1e-9 relative parity) or "fp32" (fast mode, <= 1/255 per
    8-bit channel); jitter = int seed | ("pixel", seed) | ("numpy", seed); stats = dict to fill;
    sequential=True renders with the semantics of the reference's per-pixel `render`
    (ray_tracer.py:398-434; differences listed at RT_FLAG_SEQUENTIAL in include/rt_hip.h);
    out_dtype = np.float64 (the reference's return type), np.float32 (half the bytes over PCIe; the natural
    format of precision="fp32") or np.uint8 (the reference's save_image quantisation, ray_tracer.py:606-608).
    The returned array lives in pooled page-locked memory (see _PinnedPool); the device copy of the scene and
    its workspace are kept for the next call with the same scene (clear_scene_cache() drops them).
    """
    t0 = time.time()
    camera.setup(width / height)
    n_shadow = int(scene_settings.root_number_shadow_rays)
    max_depth = int(scene_settings.max_recursions)
    if verbose:
        print(f"Max depth: {max_depth}, Shadow rays: {n_shadow}x{n_shadow}={n_shadow * n_shadow} per light")
        print(f"HIP rendering {width}x{height} = {width * height} rays ({precision})...")
    fmt = {np.dtype(np.float64): "f64", np.dtype(np.float32): "f32", np.dtype(np.uint8): "u8"}[np.dtype(out_dtype)]
    gs = _cached_scene(materials, surfaces, lights, scene_settings.background_color, device)
    img = gs.render(camera, width, height, n_shadow, max_depth, precision=precision, jitter=jitter, out_format=fmt,
                    flags=_lib.RT_FLAG_SEQUENTIAL if sequential else 0)
    st = gs.last_stats
    if stats is not None:
        stats.update(st)
    if verbose:
        print(f"HIP rendering complete in {time.time() - t0:.3f}s ({st['rays']} total rays, "
              f"{st['shadow_rays']} shadow rays)")
    return img


render_vectorized = render_hip   # same name as the reference entry point


def render_multi(scenes, camera, width, height, n_shadow, max_depth, *, precision="fp32", jitter=None, block_rows=8,
                 out_format=None, out_ptr=None, out=None, flags=0):
    """ONE frame rendered by several GpuScene handles (the same scene, one per GPU) through rt_render_multi: handle i
    renders the interleaved row blocks i, i + n, ... and copies them straight into the frame.  Returns (image, stats);
    with `out_ptr` (device memory on scenes[0]'s GPU) the image is None."""
    jit = as_jitter(jitter)
    if camera.forward is None:
        camera.setup(width / height)
    cam = _lib.CameraDesc()
    v = pack_camera(camera)
    cam.pos[:] = v[0:3]; cam.forward[:] = v[3:6]; cam.right[:] = v[6:9]; cam.up[:] = v[9:12]
    cam.screen_distance, cam.screen_width, cam.screen_height = v[12], v[13], v[14]
    prec = _PRECISION[precision]
    if out_format is None:
        out_format = "f64" if prec == _lib.RT_FP64 else "f32"
    fmt, dtype = _OUT[out_format]
    fr = _lib.Frame(int(width), int(height), int(block_rows), 0, 1, int(n_shadow), int(max_depth), prec, int(jit.mode),
                    jit.seed & 0xFFFFFFFF, fmt, int(flags))
    handles = (C.c_void_p * len(scenes))(*[gs._h for gs in scenes])
    st = _lib.Stats()
    lib = _lib.load()
    if out_ptr is not None:
        _lib.check(lib.rt_render_multi(handles, len(scenes), C.byref(cam), C.byref(fr), C.c_void_p(out_ptr), 1, C.byref(st)))
        return None, st.as_dict()
    if out is None:
        out = pinned.empty((height, width, 3), dtype)
    elif out.shape != (height, width, 3) or out.dtype != dtype or not out.flags.c_contiguous:
        raise ValueError(f"out must be a C-contiguous {np.dtype(dtype).name} array of shape {(height, width, 3)}")
    _lib.check(lib.rt_render_multi(handles, len(scenes), C.byref(cam), C.byref(fr), _ptr(out), 0, C.byref(st)))
    return out, st.as_dict()


def render(camera, scene_settings, materials, surfaces, lights, width, height, **kw):
    """Counterpart of the reference's sequential `render` (ray_tracer.py:398-434) on the GPU."""
    return render_hip(camera, scene_settings, materials, surfaces, lights, width, height, sequential=True, **kw)


def nearest_hit(surfaces, origins, dirs, device=0):
    st, sd, sm = pack_surfaces(surfaces)
    ps = PackedScene(st, sd, np.zeros(len(surfaces), dtype=np.int32), np.zeros((1, 11)), np.zeros((0, 9)), np.zeros(3))
    with GpuScene(ps, device) as gs:
        return gs.nearest_hit(origins, dirs)


def shadow_transmission(materials, surfaces, origins, dirs, max_distances, device=0):
    with GpuScene.from_objects(materials, surfaces, [], np.zeros(3), device) as gs:
        return gs.shadow_transmission(origins, dirs, max_distances)


def camera_rays(camera, width, height, y0=0, y1=None, device=0):
    y1 = height if y1 is None else y1
    lib = _lib.load()
    ps = PackedScene(np.zeros(0, np.int32), np.zeros((0, 8)), np.zeros(0, np.int32), np.zeros((1, 11)),
                     np.zeros((0, 9)), np.zeros(3))
    cam = _lib.CameraDesc()
    v = pack_camera(camera)
    cam.pos[:] = v[0:3]; cam.forward[:] = v[3:6]; cam.right[:] = v[6:9]; cam.up[:] = v[9:12]
    cam.screen_distance, cam.screen_width, cam.screen_height = v[12], v[13], v[14]
    n = (y1 - y0) * width
    o = np.empty((n, 3)); d = np.empty((n, 3))
    with GpuScene(ps, device) as gs:
        _lib.check(lib.rt_camera_rays(gs._h, C.byref(cam), int(width), int(height), int(y0), int(y1), _ptr(o), _ptr(d)))
    return o, d
