"""Scene-text writer and the synthetic many-primitive scene (BASELINE.json configs[3]).

The synthetic scene is emitted in the reference's text format (README.md:32-52 of the reference) so
that the reference / the oracle can parse exactly the same input.  Numbers are written with repr()
so they round-trip bit-exactly.
"""
from __future__ import annotations

import numpy as np

from .scene import Camera, SceneSettings, Material, Light, Sphere, InfinitePlane, Cube


def _f(*vals):
    return " ".join(repr(float(v)) for v in vals)


def scene_to_text(camera, settings, materials, surfaces, lights, header=None) -> str:
    """Serialise host objects back to scene-file text (materials first, then surfaces in list
    order, then lights — relative order inside each list is what the renderer depends on)."""
    out = []
    if header:
        out += ["# " + h for h in header.splitlines()]
    out.append("cam " + _f(*camera.position, *camera.look_at, *camera.up_vector,
                           camera.screen_distance, camera.screen_width))
    out.append("set " + _f(*settings.background_color) +
               f" {int(settings.root_number_shadow_rays)} {int(settings.max_recursions)}")
    for m in materials:
        out.append("mtl " + _f(*m.diffuse_color, *m.specular_color, *m.reflection_color,
                               m.shininess, m.transparency))
    for s in surfaces:
        if isinstance(s, Sphere):
            out.append("sph " + _f(*s.position, s.radius) + f" {int(s.material_index)}")
        elif isinstance(s, InfinitePlane):
            out.append("pln " + _f(*s.normal, s.offset) + f" {int(s.material_index)}")
        elif isinstance(s, Cube):
            out.append("box " + _f(*s.position, s.scale) + f" {int(s.material_index)}")
        else:
            raise TypeError(type(s).__name__)
    for l in lights:
        out.append("lgt " + _f(*l.position, *l.color, l.specular_intensity, l.shadow_intensity, l.radius))
    return "\n".join(out) + "\n"


def synthetic_scene_text(n_spheres=4096, n_boxes=512, seed=20260213, shadow_rays=3, max_recursions=3) -> str:
    """The C4 scene: one floor plane, `n_spheres` spheres and `n_boxes` boxes interleaved in
    generation order (cross-type tie order is exercised), 16 materials, 4 area lights."""
    rng = np.random.default_rng(seed)
    materials = []
    for _ in range(16):
        diffuse = rng.uniform(0.05, 0.95, 3)
        specular = rng.uniform(0.2, 1.0, 3)
        reflection = 0.25 * (rng.random(3) < 0.5)
        shininess = float(rng.choice([10.0, 30.0, 60.0, 100.0]))
        transparency = 0.5 if rng.random() < 0.25 else 0.0
        materials.append(Material(diffuse, specular, reflection, shininess, transparency))
    kinds = np.array([0] * n_spheres + [1] * n_boxes)
    rng.shuffle(kinds)
    surfaces = [InfinitePlane([0.0, 1.0, 0.0], -1.0, 1)]
    for kind in kinds:
        centre = [rng.uniform(-48, 48), rng.uniform(-0.5, 12), rng.uniform(-48, 48)]
        mat = int(rng.integers(1, 17))
        if kind == 0:
            surfaces.append(Sphere(centre, float(rng.uniform(0.15, 0.6)), mat))
        else:
            surfaces.append(Cube(centre, float(rng.uniform(0.4, 1.2)), mat))
    lights = [Light([sx * 30.0, 40.0, sz * 30.0], [0.5, 0.5, 0.5], 1.0, 0.8, 2.0)
              for sx in (-1, 1) for sz in (-1, 1)]
    camera = Camera([0, 30, -70], [0, 0, 0], [0, 1, 0], 1.4, 1.6)
    settings = SceneSettings([0.5, 0.7, 1.0], shadow_rays, max_recursions)
    header = (f"synthetic many-primitive scene: {n_spheres} spheres + {n_boxes} boxes, "
              f"numpy default_rng({seed})")
    return scene_to_text(camera, settings, materials, surfaces, lights, header)
