"""Pack host scene objects into the flat arrays the C-ABI takes (include/rt_hip.h, rt_scene_desc).

Layout (all float64 / int32, C-contiguous, list order preserved — surface order defines nearest-hit
tie-breaking, reference ray_tracer.py:119-125):

    surf_type[n]        0 sphere, 1 plane, 2 box
    surf_data[n][8]     sphere: cx cy cz r . . . . | plane: nx ny nz offset . . . . | box: min xyz, max xyz . .
    surf_material[n]    0-based material index (reference: `material_index - 1`, ray_tracer.py:518)
    materials[m][11]    diffuse rgb, specular rgb, reflection rgb, shininess, transparency
    lights[l][9]        position xyz, color rgb, specular_intensity, shadow_intensity, radius
    background[3]
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from .scene import Sphere, InfinitePlane, Cube

SURF_SPHERE, SURF_PLANE, SURF_BOX = 0, 1, 2


@dataclass
class PackedScene:
    surf_type: np.ndarray
    surf_data: np.ndarray
    surf_material: np.ndarray
    materials: np.ndarray
    lights: np.ndarray
    background: np.ndarray

    @property
    def counts(self):
        t = self.surf_type
        return int((t == SURF_SPHERE).sum()), int((t == SURF_PLANE).sum()), int((t == SURF_BOX).sum())


def pack_surfaces(surfaces, n_materials=None):
    n = len(surfaces)
    surf_type = np.zeros(n, dtype=np.int32)
    surf_data = np.zeros((n, 8), dtype=np.float64)
    surf_material = np.zeros(n, dtype=np.int32)
    for k, s in enumerate(surfaces):
        # Duck-typed on the attribute names of the reference's classes (sphere.py:6-8,
        # infinite_plane.py:6-9, cube.py:6-12), so the reference's own objects pack unchanged.
        if isinstance(s, Sphere) or (hasattr(s, "radius") and hasattr(s, "position")):
            surf_type[k] = SURF_SPHERE
            surf_data[k, 0:3] = s.position
            surf_data[k, 3] = s.radius
        elif isinstance(s, InfinitePlane) or (hasattr(s, "normal") and hasattr(s, "offset")):
            surf_type[k] = SURF_PLANE
            surf_data[k, 0:3] = s.normal
            surf_data[k, 3] = s.offset
        elif isinstance(s, Cube) or (hasattr(s, "min_bound") and hasattr(s, "max_bound")):
            surf_type[k] = SURF_BOX
            surf_data[k, 0:3] = s.min_bound
            surf_data[k, 3:6] = s.max_bound
        else:
            raise TypeError(f"surface {k}: unsupported type {type(s).__name__}")
        idx = int(s.material_index) - 1
        if n_materials is not None:
            # The reference indexes NumPy arrays with `material_index - 1` (ray_tracer.py:276,518):
            # a negative index wraps, anything else out of range raises IndexError.
            if idx < 0:
                idx += n_materials
            if not 0 <= idx < n_materials:
                raise IndexError(f"surface {k}: material index {s.material_index} out of range "
                                 f"for {n_materials} materials")
        surf_material[k] = idx
    return surf_type, surf_data, surf_material


def pack_scene(materials, surfaces, lights, background_color) -> PackedScene:
    mats = np.zeros((len(materials), 11), dtype=np.float64)
    for k, m in enumerate(materials):
        mats[k, 0:3] = m.diffuse_color
        mats[k, 3:6] = m.specular_color
        mats[k, 6:9] = m.reflection_color
        mats[k, 9] = m.shininess
        mats[k, 10] = m.transparency
    lgts = np.zeros((len(lights), 9), dtype=np.float64)
    for k, l in enumerate(lights):
        lgts[k, 0:3] = l.position
        lgts[k, 3:6] = l.color
        lgts[k, 6] = l.specular_intensity
        lgts[k, 7] = l.shadow_intensity
        lgts[k, 8] = l.radius
    st, sd, sm = pack_surfaces(surfaces, n_materials=len(materials) if len(surfaces) else None)
    bg = np.array(background_color, dtype=np.float64).reshape(3)
    return PackedScene(st, sd, sm, mats, lgts, bg)


def pack_camera(camera) -> np.ndarray:
    """[pos3, forward3, right3, up3, screen_distance, screen_width, screen_height] (after setup())."""
    if camera.forward is None:
        raise ValueError("camera.setup(aspect_ratio) must run before packing")
    out = np.zeros(15, dtype=np.float64)
    out[0:3] = camera.position
    out[3:6] = camera.forward
    out[6:9] = camera.right
    out[9:12] = camera.up
    out[12] = camera.screen_distance
    out[13] = camera.screen_width
    out[14] = camera.screen_height
    return out


def rows_in_part(height, block_rows=1, part=0, n_parts=1) -> np.ndarray:
    """Global row indices of a row set: rows y with ((y // block_rows) % n_parts) == part."""
    y = np.arange(height)
    if n_parts <= 1:
        return y
    return y[(y // block_rows) % n_parts == part]
