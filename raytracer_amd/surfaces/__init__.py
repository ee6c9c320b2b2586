"""Reference-compatible package name for the three primitives (see raytracer_amd.scene)."""
from ..scene import Sphere, InfinitePlane, Cube  # noqa: F401
