"""Reference-compatible module name; the class lives in raytracer_amd.scene."""
from .scene import SceneSettings  # noqa: F401
