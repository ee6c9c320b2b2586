"""Reference-compatible module name; the class lives in raytracer_amd.scene."""
from .scene import Material  # noqa: F401
