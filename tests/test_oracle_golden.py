"""The CPU oracle (oracle/rt_oracle.c) must reproduce the REFERENCE's outputs bit for bit.
The fixtures were produced by running the untouched reference (tests/golden/make_golden.py)."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import blinn_cases, ppm_md5
from oracle import oracle_py as O
from raytracer_amd import _abi as A


def total_rays(c):
    return c["primary_rays"] + c["shadow_rays"] + c["reflection_rays"] + c["transmission_rays"]


def check_counts(c_or, c_ref, W, H, s):
    # reference wrap counters: ray_create = primary + shadow, tinypush = shadow + secondary
    primary = W * H * s * s
    assert c_or["primary_rays"] == primary
    assert c_or["primary_rays"] + c_or["shadow_rays"] == c_ref["ray_create"]
    assert c_or["shadow_rays"] + c_or["reflection_rays"] + c_or["transmission_rays"] == c_ref["ray_tinypush"]
    assert c_or["sphere_tests"] == c_ref["sphere_tests"]
    assert c_or["sphere_hits"] == c_ref["sphere_hits"]
    assert c_or["polygon_tests"] == c_ref["polygon_tests"]
    assert c_or["polygon_hits"] == c_ref["polygon_hits"]


@pytest.mark.parametrize("idx", [0, 1, 2, 3])
def test_full_frames_hash_and_counts(golden, make_scene, idx):
    meta, arrays = golden
    fr = meta["frames"][idx]
    W, H, s, d = fr["W"], fr["H"], fr["sqrtSpp"], fr["depth"]
    sc = make_scene("builtin")
    cam = O.prepare_camera(sc.c, W, H, s, d)
    argb, _, cnt = O.render(sc.c, cam, nthreads=8)
    assert ppm_md5(argb, W, H) == fr["md5"]
    check_counts(cnt, fr["counts"], W, H, s)
    if (W, H, s, d) == (640, 480, 1, 4):
        assert np.array_equal(argb, arrays["builtin_640x480_s1_d4_argb"])


def test_survey_hashes(golden):
    """The hashes the survey recorded independently."""
    meta, _ = golden
    want = {(640, 480, 1, 0): "49ad4e3dbf597a6e52bcf701268b3213", (640, 480, 1, 4): "21b6c0b8d0a1bd0332763158e24de37c",
            (640, 480, 1, 8): "6cedc4933eebc74dc350f400b9b1e442", (1920, 1080, 4, 8): "8ef06efe68cbcfc377f09711e6da2f77"}
    for fr in meta["frames"]:
        assert fr["md5"] == want[(fr["W"], fr["H"], fr["sqrtSpp"], fr["depth"])]


def test_node_counts_per_depth(make_scene):
    """shade nodes per depth at 640x480, sqrtSpp 1, depth 8."""
    sc = make_scene("builtin")
    cam = O.prepare_camera(sc.c, 640, 480, 1, 8)
    _, _, cnt = O.render(sc.c, cam, nthreads=8)
    assert cnt["nodes_at_depth"][:9] == [143294, 28133, 2944, 1147, 481, 301, 166, 115, 68]


@pytest.mark.parametrize("idx", range(7))
def test_small_frames_bit_exact(golden, make_scene, idx):
    meta, arrays = golden
    fr = meta["small_frames"][idx]
    sc = make_scene(fr["scene"])
    cam = O.prepare_camera(sc.c, fr["W"], fr["H"], fr["sqrtSpp"], fr["depth"])
    argb, rad, cnt = O.render(sc.c, cam, nthreads=8)
    assert np.array_equal(argb, arrays[fr["key"] + "_argb"])
    assert np.array_equal(rad.view(np.uint32), arrays[fr["key"] + "_rad"].view(np.uint32))
    check_counts(cnt, fr["counts"], fr["W"], fr["H"], fr["sqrtSpp"])


@pytest.mark.parametrize("idx", range(5))
def test_sparse_pixels_of_full_size_configs(golden, make_scene, idx):
    """get_pixel_color pins of BASELINE configs 1-5 at their real frame sizes."""
    meta, arrays = golden
    sp = meta["sparse"][idx]
    sc = make_scene(sp["scene"])
    cam = O.prepare_camera(sc.c, sp["W"], sp["H"], sp["sqrtSpp"], sp["depth"])
    xy, want = arrays[sp["key"] + "_xy"], arrays[sp["key"] + "_rad"]
    if sp["scene"] == "c5":
        xy, want = xy[:4], want[:4]  # 16k spheres x 64 spp x 8 lights: keep the CPU suite short
    got = np.stack([O.pixel(sc.c, cam, int(x), int(y)) for x, y in xy])
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("idx,step", [(0, 4), (1, 32)])
def test_more_pins_of_the_sphere_clouds_at_full_size(pins, make_scene, idx, step):
    """tests/golden/pins.npz (the reference's get_pixel_color on 32 768 pixels of config 4 and 8192 of config 5 at
    their real frame sizes): every 4th / 32nd of them through the oracle, bit for bit (the GPU suite takes them all)."""
    meta, arrays = pins
    sp = meta["pins"][idx]
    sc = make_scene(sp["scene"])
    cam = O.prepare_camera(sc.c, sp["W"], sp["H"], sp["sqrtSpp"], sp["depth"])
    xy, want = arrays[sp["key"] + "_xy"][::step], arrays[sp["key"] + "_rad"][::step]
    got = np.stack([O.pixel(sc.c, cam, int(x), int(y)) for x, y in xy])
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_survey_radiance_samples(golden):
    """SURVEY.md appendix C.3 (printed %.9g by the survey's own harness)."""
    _, arrays = golden
    rad = arrays["builtin_640x480_s1_d4_sparse_rad"]
    want = [(0.447058797, 0.627450109, 0.870588183), (0.505998731, 0.651651263, 0.858318627),
            (0.529218018, 0.684354544, 0.89345324), (0.500535667, 0.655672193, 0.864770949),
            (0.100000001, 0, 0), (0.436202645, 0, 0), (0.673790276, 0.828926802, 1),
            (0.530928075, 0.686064601, 0.895163298)]
    assert np.array_equal(rad[:8], np.array(want, dtype=np.float32))


def test_prepare_camera(golden, make_scene):
    meta, _ = golden
    sc = make_scene("builtin")
    for W, H, s, vd, hw, hh, sw in meta["prepare_scene"]:
        cam = O.prepare_camera(sc.c, W, H, s, 4)
        got = np.array([cam.viewDistance, cam.viewPlaneHalfWidth, cam.viewPlaneHalfHeight, cam.sppWidth], np.float32)
        assert np.array_equal(got, np.array([vd, hw, hh, sw], np.float32))


def test_intersection_kats(golden, make_scene):
    meta, _ = golden
    for k in meta["kats"]["sphere"]:
        sph = A.rt_sphere(A.rt_vec4(*k["center"], 1.0), k["radius"])
        hit, pt, d = O.intersect_sphere(k["origin"], k["dir"], sph)
        assert hit == k["hit"]
        if hit:
            assert d == np.float32(k["d"]) and np.array_equal(pt, np.array(k["pt"], np.float32))
    keep = make_scene("builtin")
    floor = keep.c.objects[2].poly_obj
    for k in meta["kats"]["polygon"]:
        hit, _, _ = O.intersect_polygon(k["origin"], k["dir"], floor)
        assert hit == k["hit"]
    # the survey's own known answers (appendix A.4)
    o, dz = (0, 0, 0), (0, 0, -1)
    hit, pt, d = O.intersect_sphere(o, dz, A.rt_sphere(A.rt_vec4(0, 0, -5, 1), 1))
    assert hit == 1 and d == 4 and tuple(pt) == (0, 0, -4)
    hit, pt, d = O.intersect_sphere(o, dz, A.rt_sphere(A.rt_vec4(0, 0, 5, 1), 1))
    assert hit == 1 and d == 0  # sphere behind the ray: "hit at distance 0" quirk (ray.c:198,230-233)
    hit, pt, d = O.intersect_sphere(o, dz, A.rt_sphere(A.rt_vec4(1, 0, -5, 1), 1))
    assert hit == 1 and d == 5  # tangent
    hit, _, _ = O.intersect_sphere(o, dz, A.rt_sphere(A.rt_vec4(1.5, 0, -5, 1), 1))
    assert hit == 0


def test_checkerboard_kats(golden, make_scene):
    meta, _ = golden
    keep = make_scene("builtin")
    floor = keep.c.objects[2]
    for k in meta["kats"]["color"]:
        assert np.array_equal(O.object_color(floor, k["pt"]), np.array(k["rgb"], np.float32))


def test_slab_and_threads_do_not_change_pixels(make_scene):
    sc = make_scene("builtin")
    cam = O.prepare_camera(sc.c, 96, 72, 2, 4)
    a1, r1, c1 = O.render(sc.c, cam, nthreads=1)
    a8, r8, c8 = O.render(sc.c, cam, nthreads=8)
    assert np.array_equal(a1, a8) and np.array_equal(r1, r8) and c1 == c8
    a_s, r_s, _ = O.render(sc.c, cam, row0=24, nrows=16, nthreads=2)
    assert np.array_equal(a_s[24:40], a1[24:40]) and not a_s[:24].any() and not a_s[40:].any()


def test_tone_step_equals_the_reference_bit_for_bit(golden):
    """rt_oracle_tone against apply_tone / reinhard_tone of the compiled reference (raytrace.c:792-871),
    fixtures made by tests/golden/make_golden_tone.py."""
    import json
    _, arrays = golden
    here = os.path.dirname(os.path.abspath(__file__))
    want = np.load(os.path.join(here, "golden", "tone.npz"))
    cases = json.load(open(os.path.join(here, "golden", "tone.json")))["cases"]
    assert len(cases) == 10
    for c in cases:
        out, sums = O.tone(arrays[c["input"]], c["op"], c["ldMax"], c["lMax"], c["keyPix"])
        assert np.array_equal(out.view(np.uint32), want[c["key"]].view(np.uint32)), c["key"]
        assert np.isfinite(sums).all()


def test_second_shading_routine_equals_the_reference_bit_for_bit():
    """get_shade_color_phongblinn (raytrace.c:574-685): the restatement against frames the compiled
    reference rendered with that routine (tests/golden/make_golden_blinn.py)."""
    n = 0
    O.set_shading(1)
    try:
        for c, sc, argb_ref, rad_ref in blinn_cases():
            cam = O.prepare_camera(sc.c, c["W"], c["H"], c["sqrtSpp"], 8)
            argb, rad, cnt = O.render(sc.c, cam, nthreads=8)
            assert np.array_equal(argb, argb_ref), c["key"]
            assert np.array_equal(rad.view(np.uint32), rad_ref.view(np.uint32)), c["key"]
            assert cnt["shade_nodes"] == c["shade_nodes"]
            assert cnt["primary_rays"] + cnt["shadow_rays"] == c["counts"]["ray_create"]
            assert cnt["reflection_rays"] == cnt["transmission_rays"] == 0
            assert cnt["sphere_tests"] == c["counts"]["sphere_tests"]
            assert cnt["polygon_tests"] == c["counts"]["polygon_tests"]
            n += 1
    finally:
        O.set_shading(0)
    assert n == 6
