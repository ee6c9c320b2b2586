"""raytracer_amd — MI355X-native drop-in for the vectorized renderer of AviadDahan/raytracer.

Host objects (scene parser, Camera/Light/Material/SceneSettings, surfaces) mirror the reference;
the hot path (`render_hip`) runs in hand-written HIP kernels behind the C-ABI of librt_hip.so.
"""
from .scene import (Camera, Light, Material, SceneSettings, Sphere, InfinitePlane, Cube,  # noqa: F401
                    parse_scene_file, parse_scene_text, separate_objects, load_scene)
from .jitter import PixelKeyedJitter, NumpyStreamJitter  # noqa: F401
from .scene_cache import load_packed, pack_scene_text  # noqa: F401

__version__ = "0.1.0"


def __getattr__(name):
    # The renderer needs the HIP library; import it lazily so that parsing/packing work without it.
    if name in ("render_hip", "render_vectorized", "nearest_hit", "shadow_transmission", "camera_rays"):
        from . import render as _render
        return getattr(_render, name)
    raise AttributeError(name)
