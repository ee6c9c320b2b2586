"""Fast scene loading: text -> packed arrays without per-object Python work, plus an on-disk cache
of the packed arrays keyed by the file's content hash (the step in front of the hot
path, reference ray_tracer.py:39-91 + the packing that replaces :454-460).

`load_packed(path)` returns `(camera, settings, PackedScene)` with arrays byte-identical to
`pack_scene(*separate_objects(parse_scene_file(path)))` — the object path stays the definition, this
is the bulk route for large generated scenes (BASELINE config 4: 4,609 surfaces).  Records are
validated with line numbers in the messages; the exception types are the reference's (ValueError for
an unknown record or a bad number, IndexError for a short record or a material index out of range).
"""
from __future__ import annotations

import hashlib
import math
import os
import tempfile

import numpy as np

from .packing import SURF_BOX, SURF_PLANE, SURF_SPHERE, PackedScene
from .scene import Camera, SceneSettings

CACHE_VERSION = 2
_NPARAM = {"cam": 11, "set": 5, "mtl": 11, "sph": 5, "pln": 5, "box": 5, "lgt": 9}


def _records(text):
    """Yield (line_no, kind, params ndarray) for every record line."""
    for no, raw in enumerate(text.splitlines(), 1):
        line = raw.strip()
        if not line or line[0] == "#":
            continue
        kind, *fields = line.split()
        if kind not in _NPARAM:
            raise ValueError(f"Unknown object type: {kind} (line {no})")
        try:
            params = [float(f) for f in fields]
        except ValueError as e:
            raise ValueError(f"line {no}: {e}") from None
        if len(params) < _NPARAM[kind]:
            raise IndexError(f"line {no}: '{kind}' record needs {_NPARAM[kind]} numbers, got {len(params)}")
        yield no, kind, params


def pack_scene_text(text: str, strict: bool = True):
    """Parse scene text straight into (camera, settings, PackedScene).

    strict=True additionally rejects values the renderer cannot give a meaning to (non-finite
    numbers, radius/edge <= 0, zero plane normal, fractional material index, negative light radius);
    the reference would silently render garbage or divide by zero for those."""
    camera = settings = None
    mats, lights, st, sd, sm, where = [], [], [], [], [], []
    for no, kind, p in _records(text):
        if strict and not all(map(math.isfinite, p[: _NPARAM[kind]])):
            raise ValueError(f"line {no}: non-finite number in '{kind}' record")
        if kind == "cam":
            camera = Camera(p[0:3], p[3:6], p[6:9], p[9], p[10])
        elif kind == "set":
            settings = SceneSettings(p[0:3], p[3], p[4])
        elif kind == "mtl":
            mats.append(p[:11])
        elif kind == "lgt":
            if strict and p[8] < 0:
                raise ValueError(f"line {no}: light radius must be >= 0")
            lights.append(p[:9])
        else:
            if strict and p[4] != int(p[4]):
                raise ValueError(f"line {no}: material index {p[4]} is not an integer")
            row = [0.0] * 8
            if kind == "sph":
                if strict and not p[3] > 0:
                    raise ValueError(f"line {no}: sphere radius must be > 0")
                row[0:4] = p[0:4]; st.append(SURF_SPHERE)
            elif kind == "pln":
                n = np.array(p[0:3], dtype=np.float64)
                norm = np.linalg.norm(n)                     # InfinitePlane ctor, infinite_plane.py:6-7
                if strict and not norm > 0:
                    raise ValueError(f"line {no}: plane normal must be non-zero")
                row[0:3] = (n / norm).tolist(); row[3] = p[3]; st.append(SURF_PLANE)
            else:
                if strict and not p[3] > 0:
                    raise ValueError(f"line {no}: box edge must be > 0")
                half = p[3] / 2.0                                                  # Cube ctor, cube.py:10-12
                row[0:3] = [p[0] - half, p[1] - half, p[2] - half]
                row[3:6] = [p[0] + half, p[1] + half, p[2] + half]; st.append(SURF_BOX)
            sd.append(row); sm.append(int(p[4])); where.append(no)
    n_mat = len(mats)
    surf_material = np.zeros(len(sm), dtype=np.int32)
    for k, (mi, no) in enumerate(zip(sm, where)):
        idx = mi - 1                                         # ray_tracer.py:518 (negative indices wrap like NumPy)
        if idx < 0:
            idx += n_mat
        if not 0 <= idx < n_mat:
            raise IndexError(f"line {no}: material index {mi} out of range for {n_mat} materials")
        surf_material[k] = idx
    if strict and (camera is None or settings is None):
        raise ValueError("scene needs a 'cam' and a 'set' record")
    packed = PackedScene(np.asarray(st, dtype=np.int32).reshape(-1), np.asarray(sd, dtype=np.float64).reshape(-1, 8),
                         surf_material, np.asarray(mats, dtype=np.float64).reshape(-1, 11),
                         np.asarray(lights, dtype=np.float64).reshape(-1, 9),
                         np.array(settings.background_color if settings else [0, 0, 0], dtype=np.float64))
    return camera, settings, packed


def _cache_dir(cache_dir=None):
    d = cache_dir or os.environ.get("RT_SCENE_CACHE") or os.path.join(tempfile.gettempdir(), "raytracer_amd_scene_cache")
    os.makedirs(d, exist_ok=True)
    return d


def load_packed(file_path, cache_dir=None, use_cache=True, strict=True):
    """(camera, settings, PackedScene) of a scene file; packed arrays are cached on disk under the
    sha256 of the file's bytes (so edits invalidate, renames do not).  Cache files are plain .npz
    (no pickles) written atomically."""
    with open(file_path, "rb") as f:
        raw = f.read()
    key = hashlib.sha256(b"rt-scene-v%d-%d\0" % (CACHE_VERSION, int(strict)) + raw).hexdigest()
    path = os.path.join(_cache_dir(cache_dir), key + ".npz") if use_cache else None
    if path and os.path.exists(path):
        try:
            with np.load(path, allow_pickle=False) as z:
                cam = z["camera"]; stg = z["settings"]
                packed = PackedScene(z["surf_type"], z["surf_data"], z["surf_material"], z["materials"], z["lights"], z["background"])
            return (Camera(cam[0:3], cam[3:6], cam[6:9], cam[9], cam[10]), SceneSettings(stg[0:3], stg[3], stg[4]), packed)
        except Exception:
            pass                                             # unreadable cache entry: rebuild it
    camera, settings, packed = pack_scene_text(raw.decode("utf-8"), strict=strict)
    if path:
        cam = np.concatenate([camera.position, camera.look_at, camera.up_vector, [camera.screen_distance, camera.screen_width]])
        stg = np.concatenate([settings.background_color, [settings.root_number_shadow_rays, settings.max_recursions]])
        fd, tmp = tempfile.mkstemp(dir=os.path.dirname(path), suffix=".tmp.npz")
        os.close(fd)
        np.savez(tmp, camera=cam, settings=stg, surf_type=packed.surf_type, surf_data=packed.surf_data,
                 surf_material=packed.surf_material, materials=packed.materials, lights=packed.lights,
                 background=packed.background)
        os.replace(tmp, path)
    return camera, settings, packed
