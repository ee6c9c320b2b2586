"""Scene loading in front of the hot path: the bulk text -> packed-array route
and its disk cache must give arrays byte-identical to the object route, and reject broken files
with the reference's exception types."""
import os
import time

import numpy as np
import pytest

from conftest import scene_path


def _same(a, b):
    for f in ("surf_type", "surf_data", "surf_material", "materials", "lights", "background"):
        x, y = getattr(a, f), getattr(b, f)
        assert x.dtype == y.dtype and x.shape == y.shape and x.tobytes() == y.tobytes(), f


@pytest.mark.parametrize("name", ["pool.txt", "original.txt", "boxes.txt", "synth_small.txt"])
def test_bulk_route_equals_object_route(tmp_path, name):
    from raytracer_amd import parse_scene_file, separate_objects
    from raytracer_amd.packing import pack_scene
    from raytracer_amd.scene_cache import load_packed
    cam, st, objs = parse_scene_file(scene_path(name))
    mats, surfs, lights = separate_objects(objs)
    ref = pack_scene(mats, surfs, lights, st.background_color)
    for _ in range(2):                      # second call is served from the cache
        c2, s2, packed = load_packed(scene_path(name), cache_dir=str(tmp_path))
        _same(packed, ref)
        assert np.array_equal(c2.position, cam.position) and c2.screen_width == cam.screen_width
        assert (s2.root_number_shadow_rays, s2.max_recursions) == (st.root_number_shadow_rays, st.max_recursions)
    assert len(os.listdir(tmp_path)) == 1


def test_large_synthetic_scene_and_cache_invalidation(tmp_path):
    from raytracer_amd import parse_scene_text, separate_objects
    from raytracer_amd.packing import pack_scene
    from raytracer_amd.scene_cache import load_packed
    from raytracer_amd.synth import synthetic_scene_text
    text = synthetic_scene_text(4096, 512)
    f = tmp_path / "big.txt"
    f.write_text(text)
    cam, st, objs = parse_scene_text(text)
    ref = pack_scene(*separate_objects(objs), st.background_color)
    t0 = time.perf_counter(); _, _, a = load_packed(str(f), cache_dir=str(tmp_path / "c")); t1 = time.perf_counter()
    _, _, b = load_packed(str(f), cache_dir=str(tmp_path / "c")); t2 = time.perf_counter()
    _same(a, ref); _same(b, ref)
    assert (t2 - t1) < (t1 - t0)            # the cached load skips the parse
    f.write_text(text + "sph 0 0 0 1 1\n")  # edited file -> new key
    _, _, c = load_packed(str(f), cache_dir=str(tmp_path / "c"))
    assert len(c.surf_type) == len(ref.surf_type) + 1
    assert len(os.listdir(tmp_path / "c")) == 2


@pytest.mark.parametrize("text,exc", [
    ("cam 0 0 5 0 0 0 0 1 0 1 1\nset 0 0 0 1 1\nfoo 1 2 3\n", ValueError),                    # unknown record (ray_tracer.py:72)
    ("cam 0 0 5 0 0 0 0 1 0 1 1\nset 0 0 0 1 1\nsph 0 0 x 1 1\n", ValueError),                # bad number
    ("cam 0 0 5 0 0 0 0 1 0 1 1\nset 0 0 0 1 1\nsph 0 0 0 1\n", IndexError),                  # short record
    ("cam 0 0 5 0 0 0 0 1 0 1 1\nset 0 0 0 1 1\nsph 0 0 0 1 2\n", IndexError),                # material out of range (:518)
    ("cam 0 0 5 0 0 0 0 1 0 1 1\nset 0 0 0 1 1\nmtl 1 1 1 1 1 1 0 0 0 1 0\nsph 0 0 0 -1 1\n", ValueError),
    ("cam 0 0 5 0 0 0 0 1 0 1 1\nset 0 0 0 1 1\nmtl 1 1 1 1 1 1 0 0 0 1 0\npln 0 0 0 1 1\n", ValueError),
    ("cam 0 0 5 0 0 0 0 1 0 1 1\nset 0 0 0 1 1\nmtl 1 1 1 1 1 1 0 0 0 1 0\nsph 0 0 nan 1 1\n", ValueError),
    ("set 0 0 0 1 1\n", ValueError),                                                         # no camera
])
def test_broken_scenes_are_rejected(text, exc):
    from raytracer_amd.scene_cache import pack_scene_text
    with pytest.raises(exc):
        pack_scene_text(text)


def test_error_messages_name_the_line():
    from raytracer_amd.scene_cache import pack_scene_text
    with pytest.raises(IndexError, match="line 4"):
        pack_scene_text("cam 0 0 5 0 0 0 0 1 0 1 1\n# c\nset 0 0 0 1 1\nsph 0 0 0 1\n")
