"""Reference-compatible module name; the class lives in raytracer_amd.scene."""
from ..scene import Sphere  # noqa: F401
