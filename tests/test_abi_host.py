"""CPU-side checks: the C-ABI library loads and exports every symbol include/rt_hip.h declares,
struct layouts match the reference's, and the host helpers (scene construction, camera, P6 writer)
reproduce the reference's values.  No compute calls: there is no GPU here."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

import raytracer_amd as rt
from raytracer_amd import _abi as A
from raytracer_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "rt_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z_0-9]+|raytrace|write_image)\s*\(", src)) - {"rt_hip_ctx"})


def test_every_declared_symbol_is_exported_and_bound():
    lib = _lib.load()
    names = declared_functions()
    assert "raytrace" in names and "rt_hip_render" in names
    for n in names:
        assert hasattr(lib, n), f"librt_hip.so does not export {n}"
    assert sorted(_lib.EXPORTS) == names
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(l.split()[-1] for l in out.splitlines() if " T " in l)
    assert set(names) <= exported


def test_layouts_match_reference_x86_64():
    # SURVEY.md appendix B.1 (measured with sizeof/offsetof on the compiled reference)
    assert C.sizeof(A.rt_vec4) == 16 and C.sizeof(A.rt_color) == 12 and C.sizeof(A.rt_plane) == 16
    assert C.sizeof(A.rt_sphere) == 20 and C.sizeof(A.rt_material) == 52
    assert C.sizeof(A.rt_polygon) == 32 and A.rt_polygon.nVerticies.offset == 16 and A.rt_polygon.vertex.offset == 24
    assert C.sizeof(A.rt_pointlight) == 32 and A.rt_pointlight.color.offset == 16 and A.rt_pointlight.range.offset == 28
    assert C.sizeof(A.rt_object3d) == 104 and A.rt_object3d.debugName.offset == 8
    assert A.rt_object3d.material.offset == 16 and A.rt_object3d.g.offset == 72
    assert C.sizeof(A.rt_scene_t) == 184
    for name, off in [("eyePos", 0), ("lookAt", 16), ("upVec", 32), ("U", 48), ("V", 64), ("N", 80),
                      ("frameBufferWidth", 96), ("frameBufferHeight", 100), ("sqrtSpp", 104), ("viewDistance", 108),
                      ("viewPlaneHalfWidth", 112), ("viewPlaneHalfHeight", 116), ("sppWidth", 120),
                      ("bgColor", 124), ("ambientLightColor", 136), ("ldMax", 148), ("lMax", 152),
                      ("nLights", 156), ("nObjects", 160), ("lights", 168), ("objects", 176)]:
        assert getattr(A.rt_scene_t, name).offset == off, name


def test_header_compiles_as_c_and_cpp(tmp_path):
    src = '#include "rt_hip.h"\nint main(void){return (int)sizeof(rt_hip_stats) == 304 ? 0 : 1;}\n'
    for comp, ext in (("gcc", "c"), ("g++", "cpp")):
        f = tmp_path / f"t.{ext}"
        f.write_text(src)
        exe = tmp_path / f"t_{ext}"
        subprocess.check_call([comp, "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(f), "-o", str(exe)])
        assert subprocess.call([str(exe)]) == 0
    assert C.sizeof(A.rt_hip_stats) == 304


def test_builtin_scene_equals_the_reference_parse_scene(golden):
    meta, _ = golden
    want = meta["builtin_scene"]
    keep = rt.Scene.builtin()
    s = keep.c
    f32 = lambda v: np.array(v, dtype=np.float32)
    assert s.nObjects == want["nObjects"] and s.nLights == want["nLights"]
    assert np.array_equal(f32([s.bgColor.r, s.bgColor.g, s.bgColor.b]), f32(want["bg"]))
    assert np.array_equal(f32([s.ambientLightColor.r, s.ambientLightColor.g, s.ambientLightColor.b]), f32(want["ambient"]))
    for name in ("eye", "U", "V", "N"):
        v = getattr(s, "eyePos" if name == "eye" else name)
        assert np.array_equal(f32([v.x, v.y, v.z]).view(np.uint32), f32(want[name]).view(np.uint32)), name
    assert np.array_equal(f32([s.lookAt.x, s.lookAt.y, s.lookAt.z, s.lookAt.w]), f32(want["lookAt"]))
    L = s.lights[0]
    assert np.array_equal(f32([L.position.x, L.position.y, L.position.z, L.color.r, L.color.g, L.color.b, L.range]),
                          f32(want["light0"]))
    for i, od in enumerate(want["objects"]):
        o = s.objects[i]
        m = o.material
        assert o.geometryType == od["type"]
        got = [m.colors[0].r, m.colors[0].g, m.colors[0].b, m.colors[1].r, m.colors[1].g, m.colors[1].b,
               m.phong_ka, m.phong_kd, m.phong_ks, m.phong_ke, m.kr, m.kt, m.n]
        assert np.array_equal(f32(got), f32(od["material"]))
        if "sphere" in od:
            c = o.sphr_obj.center
            assert np.array_equal(f32([c.x, c.y, c.z, o.sphr_obj.radius]), f32(od["sphere"]))
        else:
            p = o.poly_obj
            assert np.array_equal(f32([p.plane.A, p.plane.B, p.plane.C, p.plane.F]).view(np.uint32),
                                  f32(od["plane"]).view(np.uint32))  # bit-exact incl. the -0.0 in C
            assert p.nVerticies == len(od["vertices"])
            for v, wv in enumerate(od["vertices"]):
                assert np.array_equal(f32([p.vertex[v].x, p.vertex[v].y, p.vertex[v].z, p.vertex[v].w]), f32(wv))


def test_camera_prepare_matches_prepare_scene(golden):
    meta, _ = golden
    sc = rt.Scene.builtin()
    for W, H, s, vd, hw, hh, sw in meta["prepare_scene"]:
        cam = rt.prepare_camera(sc, W, H, s, 4)
        got = np.array([cam.viewDistance, cam.viewPlaneHalfWidth, cam.viewPlaneHalfHeight, cam.sppWidth], np.float32)
        assert np.array_equal(got, np.array([vd, hw, hh, sw], np.float32))
        assert (cam.frameBufferWidth, cam.frameBufferHeight, cam.sqrtSpp, cam.depth) == (W, H, s, 4)


def test_poly_plane_kats_and_inward_walls(golden, make_scene):
    meta, _ = golden
    lib = _lib.load()
    for k in meta["kats"]["poly_plane"]:
        verts = (A.rt_vec4 * 4)(*[A.rt_vec4(v[0], v[1], v[2], 1.0) for v in k["vertices"]])
        poly = A.rt_polygon()
        poly.nVerticies = 4
        poly.vertex = C.cast(verts, C.POINTER(A.rt_vec4))
        pl = A.rt_plane()
        lib.rt_poly_plane(C.byref(pl), C.byref(poly))
        assert np.array_equal(np.array([pl.A, pl.B, pl.C, pl.F], np.float32).view(np.uint32),
                              np.array(k["plane"], np.float32).view(np.uint32))
    # the synthetic room's walls must face inwards with n.v0 <= 0 (geometry.c:29-32 forces F = |n.v0|)
    keep = make_scene("c4")
    sc = keep.c
    want = {1024: (0, 1, 0), 1025: (0, 0, 1), 1026: (1, 0, 0), 1027: (-1, 0, 0)}
    for i, n in want.items():
        p = sc.objects[i].poly_obj
        assert sc.objects[i].geometryType == A.RT_GEOMETRY_POLYGON
        assert (p.plane.A, p.plane.B, p.plane.C) == n
        v0 = p.vertex[0]
        assert p.plane.A * v0.x + p.plane.B * v0.y + p.plane.C * v0.z + p.plane.F == 0


def test_synthetic_scene_is_deterministic_and_in_spec():
    keep = [rt.Scene.synthetic(0xC0FFEE04, 1024, 4, 1, 0), rt.Scene.synthetic(0xC0FFEE04, 1024, 4, 1, 0),
            rt.Scene.synthetic(0xC0FFEE05, 1024, 4, 1, 0)]
    a, b, c = (k.c for k in keep)
    same = diff = 0
    for i in range(1024):
        sa, sb, sc_ = a.objects[i].sphr_obj, b.objects[i].sphr_obj, c.objects[i].sphr_obj
        assert (sa.center.x, sa.center.y, sa.center.z, sa.radius) == (sb.center.x, sb.center.y, sb.center.z, sb.radius)
        diff += (sa.center.x != sc_.center.x)
        assert -12 <= sa.center.x <= 12 and 0.3 <= sa.center.y <= 8 and -40 <= sa.center.z <= -4
        assert 0.15 <= sa.radius <= 0.6
        m = a.objects[i].material
        assert (m.kr, m.kt) == [(0, 0), (0.75, 0), (np.float32(0.01), np.float32(0.85)),
                                (np.float32(0.3), np.float32(0.4))][i % 4]
    assert diff > 1000
    with pytest.raises(ValueError):
        rt.Scene.synthetic(1, 0, 0, 1, 0)
    with pytest.raises(ValueError):
        rt.Scene.synthetic(1, 10, 5, 1, 0)


def test_write_image_is_raw_p6(tmp_path):
    W, H = 5, 3
    buf = (0xFF000000 | (np.arange(W * H, dtype=np.uint32) * 0x010203)).astype(np.uint32)
    f = tmp_path / "o.ppm"
    assert rt.write_image(f, buf, W, H) == 1
    data = f.read_bytes()
    head = b"P6\n5 3\n255\n"
    assert data.startswith(head) and len(data) == len(head) + 3 * W * H
    rgb = np.frombuffer(data[len(head):], np.uint8).reshape(-1, 3)
    assert np.array_equal(rgb[:, 0], (buf >> 16) & 255) and np.array_equal(rgb[:, 2], buf & 255)
    assert rt.write_image(tmp_path / "nodir" / "o.ppm", buf, W, H) == 0


def test_parse_scene_ignores_contents_and_reports_missing_file(tmp_path):
    f = tmp_path / "scene"
    f.write_text("camera { position 1 2 3 }\n# comment line\nsphere { radius 99 }\n")
    s = rt.parse_scene(f)
    assert s.c.nObjects == 3 and s.c.objects[0].sphr_obj.radius == np.float32(1.15)
    with pytest.raises(OSError):
        rt.parse_scene(tmp_path / "missing")


def test_product_package_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under raytracer_amd/ may reference it."""
    pkg = os.path.join(ROOT, "raytracer_amd")
    for dirpath, _, files in os.walk(pkg):
        if "build" in dirpath:
            continue
        for fn in files:
            if fn.endswith((".py", ".c", ".h", ".hip", ".cpp")) or fn == "Makefile":
                text = open(os.path.join(dirpath, fn), errors="replace").read()
                assert "oracle" not in text.lower(), f"{fn} mentions the oracle"
    out = subprocess.run(["ldd", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in out and "libref" not in out


def _scene_bytes(sc):
    """Everything of an rt_scene_t that is data (pointers and names left out)."""
    c = sc.c
    parts = [bytes(c)[:C.sizeof(A.rt_scene_t) - 16]]         # up to the two array pointers
    for i in range(c.nObjects):
        o = c.objects[i]
        parts.append(bytes(o.material))
        parts.append(bytes([o.geometryType & 0xFF]))
        if o.geometryType == 1:
            parts.append(bytes(o.sphr_obj))
        else:
            p = o.poly_obj
            parts.append(bytes(p.plane))
            parts.append(bytes(C.string_at(p.vertex, p.nVerticies * 16)))
    for i in range(c.nLights):
        parts.append(bytes(c.lights[i]))
    return b"".join(parts)


def test_scene_file_loader_reproduces_the_builtin_scene():
    """rt_load_scene really reads the grammar of the reference's `scene` file.
    tests/golden/whitted.scene states the hard-coded scene in that grammar; loading it must give the
    very bytes rt_scene_builtin() (= the reference's parse_scene, checked above) builds.  The reference
    itself never interprets the file, so beyond this equality the loader is 'parity unpinned'."""
    got = rt.load_scene(os.path.join(ROOT, "tests", "golden", "whitted.scene"))
    want = rt.Scene.builtin()
    assert (got.c.nObjects, got.c.nLights) == (3, 1)
    assert _scene_bytes(got) == _scene_bytes(want)
    assert got.c.objects[0].debugName == b"Sphere_1" and got.c.objects[2].debugName == b"Floor"
    got.free()


def test_scene_file_loader_grammar_and_errors(tmp_path, capfd):
    def load(text):
        f = tmp_path / "s.scene"
        f.write_text(text)
        return rt.load_scene(str(f))

    # the shape of the reference's own file: no material keys, no world block, '#' comments, 1.0f
    sc = load("desc { 2 1 }\ncamera { position 0 4.5 0 lookat 0 0 0 upvector 0 1 0 projection 45.0 1.0 1000.0 }\n"
              "pointlight { position 1 8 1 diffuse 1.0 1.0 1.0 }\n"
              "sphere { name s1 position 0 -5 -6 radius 1.15 diffuse 1.0 0.0 0.0 } # red\n"
              "polygon { name p point 7 0 0 point 7 0 -100 point -15 0 -100 point -15 0 0 diffuse .5 .5 1.0f }\n")
    c = sc.c
    assert (c.nObjects, c.nLights) == (2, 1)
    # camera basis as scene.c:236-258: N = normalize(lookat - position) = (0,-1,0); V = up; U = N x V = 0
    assert (c.N.x, c.N.y, c.N.z) == (0.0, -1.0, 0.0) and (c.V.x, c.V.y, c.V.z) == (0.0, 1.0, 0.0)
    assert (c.U.x, c.U.y, c.U.z) == (0.0, 0.0, 0.0)
    assert c.objects[0].geometryType == 1 and abs(c.objects[0].sphr_obj.radius - 1.15) < 1e-7
    assert c.objects[1].poly_obj.nVerticies == 4 and c.objects[1].poly_obj.plane.B == 1.0
    m = c.objects[0].material                                   # defaults = the builtin floor's material
    assert (m.phong_ka, m.kr, m.kt, m.n) == (np.float32(0.1), 0.0, 0.0, 1.0)
    assert c.lights[0].range == 150.0
    sc.free()
    for bad, why in [("sphere { radius }", "number expected"), ("sphere { colour 1 2 3 }", "unknown key"),
                     ("torus { }", "unknown block"), ("polygon { point 0 0 0 point 1 0 0 }", "at least three"),
                     ("desc { 1 0 }", "announces"), ("sphere { radius 1", "file ends"), ("sphere radius", "'{' expected")]:
        with pytest.raises(ValueError):
            load(bad)
        assert why in capfd.readouterr().out
    with pytest.raises(ValueError):
        rt.load_scene(str(tmp_path / "missing"))


def test_command_line_host_keeps_the_reference_messages(tmp_path):
    """raytracer_amd/rt (csrc/rt_main.c): main.c:127-131 and :153-157 - neither path touches a GPU."""
    import subprocess
    exe = os.path.join(os.path.dirname(rt.__file__), "rt")
    assert os.path.exists(exe), "run __graft_entry__.build()"
    p = subprocess.run([exe, "out.ppm", "scene"], capture_output=True, text=True)
    assert p.returncode == 1
    assert p.stdout == "raytrace outputFile sceneFile imgWidth imgHeight samplesPerPixel^2 depth\n"
    p = subprocess.run([exe, str(tmp_path / "o.ppm"), str(tmp_path / "missing"), "8", "8", "1", "1"],
                       capture_output=True, text=True)
    assert p.returncode == 1
    assert "Running Ray Tracer...\n" in p.stdout and p.stdout.endswith("Error parsing scene file.  Exiting...\n")
    p = subprocess.run([exe, "o.ppm", "scene", "8", "8", "1", "1", "--frobnicate"], capture_output=True, text=True)
    assert p.returncode == 1 and "unknown option" in p.stdout


def test_scene_loader_under_sanitizers_on_mutated_files():
    """rt_load_scene compiled with gcc -fsanitize=address,undefined on 300 mutated scene files
    (tools/fuzz_loader.py): no finding, whatever the file says."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_loader.py"), "300", "7"], capture_output=True,
                       text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "300 files, 0 batches with findings" in p.stdout


def test_launch_plan_order_is_a_stable_sort_slowest_first():
    """rt_launch_plan_order (the host half of k_primary's launch plan): every block once, by its slowest wave,
    slowest first, ties in raster order; times beyond 16 bits are compared by their high bits."""
    import numpy as np
    from raytracer_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(7)
    for nbx, nty, hi in [(1, 1, 10), (7, 5, 50), (60, 135, 12000), (3, 4, 1 << 30), (120, 270, 3)]:
        ticks = rng.integers(0, hi, size=(nty * nbx, 4), dtype=np.uint32)
        ticks[rng.integers(0, nty * nbx, size=max(1, nty * nbx // 3))] = 0  # sky tiles: many ties
        plan = np.zeros(nty * nbx, dtype=np.uint32)
        lib.rt_launch_plan_order(ticks.ctypes.data, nbx, nty, plan.ctypes.data)
        bx, ty = plan & 0x7FFF, plan >> 15
        assert (bx < nbx).all() and (ty < nty).all()
        blk = ty.astype(np.int64) * nbx + bx
        assert np.array_equal(np.sort(blk), np.arange(nty * nbx))  # a permutation of the raster
        cost = ticks.max(axis=1).astype(np.int64)
        shift = 0
        while (int(cost.max()) >> shift) >= 1 << 16:
            shift += 1
        key = cost >> shift
        want = np.lexsort((np.arange(nty * nbx), -key))  # by key descending, then raster order
        assert np.array_equal(blk, want)
