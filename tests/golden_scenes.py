"""Scenes used by tests/gen_golden.py (built there from the reference's classes) rebuilt from this
package's host objects, plus the named primitives of fixture g4."""
from raytracer_amd import Sphere, InfinitePlane, Cube, Material


def g4_primitives():
    return {
        "sphere_unit": Sphere([0, 0, 0], 1.0, 1),
        "sphere_off": Sphere([0.3, -0.2, 0.5], 0.75, 1),
        "plane_y": InfinitePlane([0, 1, 0], -1.0, 1),
        "plane_tilt": InfinitePlane([1, 2, -0.5], 0.25, 1),
        "box_unit": Cube([0, 0, 0], 2.0, 1),
        "box_off": Cube([0.3, -0.2, 0.5], 1.3, 1),
    }


def g4_tie_surfaces(order):
    objs = {
        "pln": InfinitePlane([0, 1, 0], -1.0, 1),
        "box": Cube([0, -0.5, 3], 1.0, 2),
        "sph": Sphere([2, 0, 0], 1.0, 3),
        "sph2": Sphere([2, 0, 0], 1.0, 4),
        "box2": Cube([-2, 0, 0], 1.0, 5),
    }
    return [objs[str(k)] for k in order]


def shadow_walk_scene():
    mats = [Material([1, 1, 1], [1, 1, 1], [0, 0, 0], 10, 0.0),
            Material([1, 1, 1], [1, 1, 1], [0, 0, 0], 10, 0.5),
            Material([1, 1, 1], [1, 1, 1], [0, 0, 0], 10, 0.9),
            Material([1, 1, 1], [1, 1, 1], [0, 0, 0], 10, 1e-7)]
    surfs = [Sphere([0, 0, 0], 1.0, 2), Sphere([3, 0, 0], 1.0, 3), Cube([6, 0, 0], 1.5, 2),
             Sphere([9, 0, 0], 0.8, 1), Sphere([0, 4, 0], 1.0, 4), InfinitePlane([0, 1, 0], -3.0, 3)]
    for k in range(12):
        surfs.append(Cube([0, 8, 3 + 1.0 * k], 0.5, 3))
    return mats, surfs
