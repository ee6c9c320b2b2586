"""Host-side scene objects and scene-file parser.

These are the objects the renderer's drop-in boundary reads.  They keep the
attribute names of the reference's classes so that code written against the reference
(`camera.position`, `light.shadow_intensity`, `sphere.material_index`, ...) keeps working:

    Camera          reference camera.py:5-28
    Light           reference light.py:4-10
    Material        reference material.py:4-10
    SceneSettings   reference scene_settings.py:4-8
    Sphere          reference surfaces/sphere.py:5-8
    InfinitePlane   reference surfaces/infinite_plane.py:5-9   (normal is unit length after the ctor)
    Cube            reference surfaces/cube.py:5-12            (min_bound / max_bound derived)
    parse_scene_file / separate_objects   reference ray_tracer.py:39-91

The surfaces' ``intersect_batch(origins, dirs) -> (t, normals)`` method is kept as well, but it is
served by the HIP library (one-primitive scene through ``rt_nearest_hit``), not by NumPy: there is
no CPU compute path in this package.
"""
from __future__ import annotations

import numpy as np

__all__ = [
    "Camera", "Light", "Material", "SceneSettings", "Sphere", "InfinitePlane", "Cube",
    "parse_scene_file", "parse_scene_text", "separate_objects", "load_scene",
]


def _vec3(values) -> np.ndarray:
    v = np.array(values, dtype=np.float64)
    if v.shape != (3,):
        raise ValueError(f"expected 3 components, got shape {v.shape}")
    return v


def _unit(v: np.ndarray) -> np.ndarray:
    # 1-D np.linalg.norm == sqrt(dot(v, v)); the reference normalises with it everywhere on the host.
    return v / np.linalg.norm(v)


class Camera:
    """Pinhole camera; `setup()` derives the orthonormal basis (reference camera.py:17-28)."""

    def __init__(self, position, look_at, up_vector, screen_distance, screen_width):
        self.position = _vec3(position)
        self.look_at = _vec3(look_at)
        self.up_vector = _vec3(up_vector)
        self.screen_distance = screen_distance
        self.screen_width = screen_width
        self.forward = self.right = self.up = None
        self.screen_height = None

    def setup(self, aspect_ratio):
        self.forward = _unit(self.look_at - self.position)
        self.right = _unit(np.cross(self.forward, self.up_vector))
        self.up = _unit(np.cross(self.right, self.forward))
        self.screen_height = self.screen_width / aspect_ratio

    # Ray generation is part of the hot path; it runs on the GPU.
    def generate_all_rays(self, image_width, image_height):
        """(N,3) origins and unit directions, row-major (reference camera.py:46-73)."""
        return self.generate_rays_for_rows(image_width, image_height, 0, image_height)

    def generate_rays_for_rows(self, image_width, image_height, y_start, y_end):
        """Rays of rows [y_start, y_end) using the *global* height (reference camera.py:75-104)."""
        from . import render as _render
        if self.forward is None:
            self.setup(image_width / image_height)
        return _render.camera_rays(self, image_width, image_height, y_start, y_end)


class Light:
    def __init__(self, position, color, specular_intensity, shadow_intensity, radius):
        self.position = _vec3(position)
        self.color = _vec3(color)
        self.specular_intensity = specular_intensity
        self.shadow_intensity = shadow_intensity
        self.radius = radius


class Material:
    def __init__(self, diffuse_color, specular_color, reflection_color, shininess, transparency):
        self.diffuse_color = _vec3(diffuse_color)
        self.specular_color = _vec3(specular_color)
        self.reflection_color = _vec3(reflection_color)
        self.shininess = shininess
        self.transparency = transparency


class SceneSettings:
    def __init__(self, background_color, root_number_shadow_rays, max_recursions):
        self.background_color = _vec3(background_color)
        self.root_number_shadow_rays = int(root_number_shadow_rays)
        self.max_recursions = int(max_recursions)


class _Surface:
    """Common part of the three primitives: the batch intersection served by the GPU."""

    material_index: int

    def intersect_batch(self, ray_origins, ray_directions):
        """t (N,) with +inf for a miss, normals (N,3) zero for a miss — same contract as the
        reference's per-surface method (sphere.py:45, infinite_plane.py:26, cube.py:64)."""
        from . import render as _render
        t, _idx, normals = _render.nearest_hit([self], ray_origins, ray_directions)
        return t, normals


class Sphere(_Surface):
    def __init__(self, position, radius, material_index):
        self.position = _vec3(position)
        self.radius = radius
        self.material_index = material_index


class InfinitePlane(_Surface):
    def __init__(self, normal, offset, material_index):
        self.normal = _unit(_vec3(normal))
        self.offset = offset
        self.material_index = material_index


class Cube(_Surface):
    def __init__(self, position, scale, material_index):
        self.position = _vec3(position)
        self.scale = scale
        self.material_index = material_index
        half = scale / 2.0
        self.min_bound = self.position - half
        self.max_bound = self.position + half


# record type -> (number of parameters, constructor from the float list)
_RECORDS = {
    "cam": (11, lambda p: Camera(p[0:3], p[3:6], p[6:9], p[9], p[10])),
    "set": (5, lambda p: SceneSettings(p[0:3], p[3], p[4])),
    "mtl": (11, lambda p: Material(p[0:3], p[3:6], p[6:9], p[9], p[10])),
    "sph": (5, lambda p: Sphere(p[0:3], p[3], int(p[4]))),
    "pln": (5, lambda p: InfinitePlane(p[0:3], p[3], int(p[4]))),
    "box": (5, lambda p: Cube(p[0:3], p[3], int(p[4]))),
    "lgt": (9, lambda p: Light(p[0:3], p[3:6], p[6], p[7], p[8])),
}


def parse_scene_text(text: str):
    """Parse scene-file text (format: reference README.md:32-52, ray_tracer.py:44-73).

    Blank lines and lines starting with '#' are skipped; a later `cam`/`set` replaces an earlier
    one; an unknown record type raises ValueError like the reference does.
    """
    camera = settings = None
    objects = []
    for raw in text.splitlines():
        line = raw.strip()
        if not line or line[0] == "#":
            continue
        kind, *fields = line.split()
        params = [float(f) for f in fields]
        if kind not in _RECORDS:
            raise ValueError("Unknown object type: {}".format(kind))
        nparam, make = _RECORDS[kind]
        if len(params) < nparam:
            raise IndexError(f"'{kind}' record needs {nparam} numbers, got {len(params)}")
        obj = make(params)
        if kind == "cam":
            camera = obj
        elif kind == "set":
            settings = obj
        else:
            objects.append(obj)
    return camera, settings, objects


def parse_scene_file(file_path):
    with open(file_path, "r") as f:
        return parse_scene_text(f.read())


def separate_objects(objects):
    """Split the parsed object list into (materials, surfaces, lights), file order kept per list.

    Surface list order defines nearest-hit tie-breaking (reference ray_tracer.py:119-125).
    """
    materials = [o for o in objects if isinstance(o, Material)]
    surfaces = [o for o in objects if isinstance(o, _Surface)]
    lights = [o for o in objects if isinstance(o, Light)]
    return materials, surfaces, lights


def load_scene(file_path, shadow_rays=None, max_recursions=None):
    """Convenience used by the benchmarks: parse + separate + optional `set` overrides
    (the BASELINE configs override the file's shadow-ray root and recursion cap)."""
    camera, settings, objects = parse_scene_file(file_path)
    if shadow_rays is not None:
        settings.root_number_shadow_rays = int(shadow_rays)
    if max_recursions is not None:
        settings.max_recursions = int(max_recursions)
    materials, surfaces, lights = separate_objects(objects)
    return camera, settings, materials, surfaces, lights
