"""CPU-only tests: host logic (parser, packer, row sets, jitter, quantiser, synthetic scene) and
that librt_hip.so loads and exports every symbol include/rt_hip.h declares (no compute calls)."""
import os
import re

import numpy as np
import pytest

from conftest import ROOT, scene_path


def test_library_exports_every_declared_symbol():
    from raytracer_amd import _lib
    header = open(os.path.join(ROOT, "include", "rt_hip.h")).read()
    declared = set(re.findall(r"^(?:int|void|const char\*)\s+(rt_[a-z_0-9]+)\s*\(", header, flags=re.M))
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), name


def test_ctypes_struct_sizes_match_header_layout():
    import ctypes as C
    from raytracer_amd import _lib
    assert C.sizeof(_lib.CameraDesc) == 15 * 8
    assert C.sizeof(_lib.Frame) == 12 * 4
    assert C.sizeof(_lib.SceneDesc) == 8 * 8 + 24
    assert C.sizeof(_lib.Stats) == 4 * 8 + 2 * 4 + 2 * 32 * 8 + 3 * 4 + 2 * 4 + 4 + 8


def test_rows_in_part_matches_library():
    import ctypes as C
    from raytracer_amd import _lib
    from raytracer_amd.packing import rows_in_part
    lib = _lib.load()
    for H, br, n in [(54, 4, 3), (1080, 8, 8), (7, 16, 2), (2160, 16, 8), (5, 1, 8), (33, 5, 4)]:
        total = 0
        for part in range(n):
            r = C.c_int32(-1)
            assert lib.rt_rows_in_part(H, br, part, n, C.byref(r)) == 0
            assert r.value == len(rows_in_part(H, br, part, n))
            total += r.value
        assert total == H
    r = C.c_int32()
    assert lib.rt_rows_in_part(10, 0, 0, 1, C.byref(r)) != 0
    assert b"rt_rows_in_part" in lib.rt_last_error()


def test_parser_roundtrip_and_errors(tmp_path):
    from raytracer_amd import parse_scene_file, parse_scene_text, separate_objects, Sphere, InfinitePlane
    from raytracer_amd.synth import scene_to_text
    cam, st, objs = parse_scene_file(scene_path("pool.txt"))
    mats, surfs, lights = separate_objects(objs)
    assert (len(mats), len(surfs), len(lights)) == (7, 7, 5)
    assert isinstance(surfs[0], InfinitePlane) and isinstance(surfs[1], Sphere)
    assert st.root_number_shadow_rays == 5 and st.max_recursions == 10
    txt = scene_to_text(cam, st, mats, surfs, lights)
    cam2, st2, objs2 = parse_scene_text(txt)
    m2, s2, l2 = separate_objects(objs2)
    assert np.array_equal(cam2.position, cam.position) and cam2.screen_width == cam.screen_width
    assert all(np.array_equal(a.position, b.position) for a, b in zip(surfs[1:], s2[1:]))
    with pytest.raises(ValueError, match="Unknown object type"):
        parse_scene_text("foo 1 2 3\n")
    with pytest.raises(ValueError):
        parse_scene_text("sph 0 0 zero 1 1\n")
    later = parse_scene_text("set 0 0 0 1 1\nset 1 1 1 4 7\n# c\n\n")[1]
    assert later.root_number_shadow_rays == 4 and later.max_recursions == 7


def test_packer_layout_and_material_index_rules():
    from raytracer_amd import Sphere, InfinitePlane, Cube, Material, Light
    from raytracer_amd.packing import pack_scene
    mats = [Material([.1, .2, .3], [.4, .5, .6], [.7, .8, .9], 10, 0.25)] * 2
    surfs = [Cube([1, 2, 3], 2.0, 2), InfinitePlane([0, 2, 0], -1.0, 1), Sphere([4, 5, 6], 0.5, 0)]
    ps = pack_scene(mats, surfs, [Light([1, 2, 3], [.5, .5, .5], 1, 0.9, 2.0)], [0, 0.5, 1])
    assert list(ps.surf_type) == [2, 1, 0]
    assert np.allclose(ps.surf_data[0, :6], [0, 1, 2, 2, 3, 4])
    assert np.allclose(ps.surf_data[1, :4], [0, 1, 0, -1])           # plane normal is unit after the ctor
    assert list(ps.surf_material) == [1, 0, 1]                        # index 0 wraps to the last material
    assert ps.materials.shape == (2, 11) and ps.lights.shape == (1, 9)
    with pytest.raises(IndexError):
        pack_scene(mats, [Sphere([0, 0, 0], 1, 3)], [], [0, 0, 0])


def test_quantiser_is_save_image():
    from raytracer_amd.image import quantize_u8
    x = np.array([-0.5, 0.0, 0.5, 0.999, 1.0, 2.376, 1 / 255, 254.9999 / 255])
    assert list(quantize_u8(x)) == [0, 0, 127, 254, 255, 255, 1, 254]


def test_synthetic_scene_is_deterministic_and_parses():
    from raytracer_amd import parse_scene_text, separate_objects, Cube, Sphere
    from raytracer_amd.synth import synthetic_scene_text
    a = synthetic_scene_text(64, 8, seed=3)
    assert a == synthetic_scene_text(64, 8, seed=3)
    cam, st, objs = parse_scene_text(a)
    mats, surfs, lights = separate_objects(objs)
    assert len(mats) == 16 and len(lights) == 4 and len(surfs) == 1 + 64 + 8
    assert sum(isinstance(s, Cube) for s in surfs) == 8 and sum(isinstance(s, Sphere) for s in surfs) == 64


def test_flop_model_constants():
    from raytracer_amd.flops import algorithmic_flops
    # pool.txt: 6 spheres + 1 plane -> F_q = 7 + 23*6 + 12 = 157
    f = algorithmic_flops(n_sph=6, n_pln=1, n_box=0, n_lights=5, n_shadow=5, rays=1, hits=0, children=0, segments=0)
    assert f == 157 + 20
