"""Reference-compatible module name; the class lives in raytracer_amd.scene."""
from ..scene import InfinitePlane  # noqa: F401
