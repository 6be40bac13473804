"""The tolerance build of the trace kernels (rt_hip_ctx_set_numerics(RT_NUMERICS_FAST), RT_HIP_NUMERICS=fast):
fused multiply-adds, 1-ulp rcp / rsq / sqrt, float sphere roots - checked at the north star's bar, NOT at
bit identity: 8-bit RGB within +-1 LSB on >= 99.9 % of the pixels, per-pixel radiance relative error <= 1e-4
on >= 99.9 % of the pixels (a silhouette or polygon-edge sample whose verdict sits within rounding of its
threshold flips, as between two CPU builds of the reference: SURVEY.md appendix C.1), ray counts within
0.01 % of the reference's.  Everything goes through the C-ABI; the oracle and the reference's fixtures
are the checker.

One class of frames is outside any tolerance build's reach, and the tests say so where they meet it: with
the eye at x = 0 and an ODD frame width the sample columns are integers in sub-sample units, and whole
columns of samples land EXACTLY on the checkerboard's colour boundaries (odd world x: object3d.c:97-114;
floor hits at x = 4.5 * column / row).  Which colour such a sample gets is decided by the last bit of the
reference's correctly rounded divide and square root; about half of them flip under 1-ulp reciprocals
(1.2 % of the pixels of the 67x45 fixture).  Even widths - every BASELINE config - put the columns at half
integers, where 4.5 * column / row is never an integer.

The sphere-cloud scenes (synthetic, configs 4/5) are ill-conditioned by construction: small mirror and glass
spheres amplify a direction's last bit by ~2 * distance / radius per bounce, so ANY change of rounding moves a
few per cent of the pixels by more than 1e-4 - the reference against itself compiled with -O3 -march=native:
2.2 %, 5 % and 12.5 % of the pixels of three such frames (profiles/r02_ref_flag_sensitivity.txt, made by
tools/ref_flag_sensitivity.py).  Only the exact build meets the north star's bar on those (bit for bit:
tests/test_gpu_parity.py); the tolerance build is held to that yardstick there (SCENE_FRAC), and to the
north star's own bar on the reference's scene, configs 1-3."""
import numpy as np
import pytest

import raytracer_amd as rt
from conftest import parity_report, rgb_bytes
from oracle import oracle_py as O
from test_gpu_parity import LSB_FRAC, REL_FRAC, assert_parity, rays_of, render_dev, torch_cuda  # noqa: F401

pytestmark = pytest.mark.gpu

SCENE_FRAC = 0.5    # sphere clouds: fraction of pixels within 1e-4 (the reference's own -O2 / -O3 builds: 0.875 .. 0.98
                    # on three frames; dense clouds at 9 spp and depth 4+ go lower) - with it the tests ask for the same
                    # picture: 90 % of the pixels within 1 LSB, a median error of rounding size, a small mean difference


@pytest.fixture()
def fast_env(monkeypatch):
    monkeypatch.setenv("RT_HIP_NUMERICS", "fast")


def fast_ctx(sc, **kw):
    ctx = rt.Context(**kw)
    ctx.upload(sc)
    ctx.set_numerics(rt.NUMERICS_FAST)
    return ctx


@pytest.mark.parametrize("idx", range(7))
def test_fast_small_frames_against_reference_fixtures(golden, make_scene, fast_env, idx):
    meta, arrays = golden
    fr = meta["small_frames"][idx]
    sc = make_scene(fr["scene"])
    cam = rt.prepare_camera(sc, fr["W"], fr["H"], fr["sqrtSpp"], fr["depth"])
    argb, rad = rt.render(sc, cam)
    if fr["W"] % 2 or fr["scene"] != "builtin":  # see the module docstring: boundary columns, sphere clouds
        lsb_ok, rel_ok, med, _ = parity_report(argb, rad, arrays[fr["key"] + "_argb"], arrays[fr["key"] + "_rad"])
        assert lsb_ok >= (0.98 if fr["scene"] == "builtin" else 0.9), (fr["key"], lsb_ok)
        assert rel_ok >= (0.98 if fr["scene"] == "builtin" else SCENE_FRAC) and med <= 1e-6, (fr["key"], rel_ok, med)
        return
    assert_parity(argb, rad, arrays[fr["key"] + "_argb"], arrays[fr["key"] + "_rad"], fr["key"])


def test_fast_config1_against_the_reference_frame(golden, fast_env):
    """BASELINE config 1 through raytrace(), the reference's own signature, against the reference's frame."""
    _, arrays = golden
    sc = rt.Scene.builtin()
    W, H = 640, 480
    buf = np.zeros(W * H, dtype=np.uint32)
    rt.raytrace(buf, sc, 45.0, W / H, 1.0, 200.0, W, H, 1, 4)
    want = arrays["builtin_640x480_s1_d4_argb"].reshape(-1)
    assert (np.abs(rgb_bytes(buf) - rgb_bytes(want)).max(-1) <= 1).mean() >= LSB_FRAC
    argb, rad = rt.render(sc, rt.prepare_camera(sc, W, H, 1, 4))
    a_o, r_o, _ = O.render(sc.c, rt.prepare_camera(sc, W, H, 1, 4), nthreads=16)
    assert_parity(argb, rad, a_o, r_o, "C1 fast")


def test_fast_c2_full_frame_and_ray_counts(torch_cuda, golden, make_scene):
    """BASELINE config 2 at full size: the whole frame against the oracle (= the reference, bit for bit),
    ray counts against the reference's --wrap counters."""
    meta, _ = golden
    fr = [f for f in meta["frames"] if f["W"] == 1920][0]
    sc = make_scene("builtin")
    cam = rt.prepare_camera(sc, 1920, 1080, 4, 8)
    ctx = fast_ctx(sc)
    argb, rad, st = render_dev(torch_cuda, ctx, cam)
    a_o, r_o, c_o = O.render(sc.c, cam, nthreads=16)
    assert_parity(argb, rad, a_o, r_o, "C2 fast")
    ref_rays = fr["counts"]["ray_create"] + fr["counts"]["ray_tinypush"] - (fr["counts"]["ray_create"] - 1920 * 1080 * 16)
    assert abs(rays_of(st) - ref_rays) <= 1e-4 * ref_rays
    assert abs(st.shadow_rays - c_o["shadow_rays"]) <= 1e-4 * c_o["shadow_rays"]
    assert abs(st.reflection_rays - c_o["reflection_rays"]) <= 1e-4 * c_o["reflection_rays"]
    assert abs(st.transmission_rays - c_o["transmission_rays"]) <= 1e-4 * c_o["transmission_rays"]
    # the exact build on the same context gives the reference's pixels again, the switch back and forth is clean
    ctx.set_numerics(rt.NUMERICS_EXACT)
    argb_e, _, st_e = render_dev(torch_cuda, ctx, cam)
    assert (argb_e != a_o).sum() == 0 and rays_of(st_e) == ref_rays
    ctx.set_numerics(rt.NUMERICS_FAST)
    argb2, rad2, _ = render_dev(torch_cuda, ctx, cam)
    assert np.array_equal(argb2, argb) and np.array_equal(rad2.view(np.uint32), rad.view(np.uint32))  # deterministic
    with pytest.raises(rt.RtHipError):
        ctx.set_numerics(5)
    ctx.close()


def test_fast_sparse_pins_of_full_size_configs(torch_cuda, golden, make_scene):
    """The reference's get_pixel_color pins of configs 1-3 (builtin scene), config 4 (1024 spheres, 4K) and
    config 5 (16 384 spheres, 8 lights, depth 16, 8K, 64 spp), rendered at the real frame sizes."""
    meta, arrays = golden
    ok = {"builtin": 0, "c4": 0, "c5": 0}
    total = dict(ok)
    for sp in meta["sparse"]:
        sc = make_scene(sp["scene"])
        cam = rt.prepare_camera(sc, sp["W"], sp["H"], sp["sqrtSpp"], sp["depth"])
        ctx = fast_ctx(sc, queue_cap=1 << 23)
        limit = {"builtin": 128, "c4": 16, "c5": 4}[sp["scene"]]
        xy, want = arrays[sp["key"] + "_xy"][:limit], arrays[sp["key"] + "_rad"][:limit]
        rows = {}
        for (x, y), w in zip(xy, want):
            if int(y) not in rows:
                rows[int(y)] = render_dev(torch_cuda, ctx, cam, int(y), 1)[1][int(y)]
            rel = np.abs(rows[int(y)][x] - w) / np.maximum(np.abs(w), 1e-3)
            ok[sp["scene"]] += bool((rel <= 1e-4).all())
            total[sp["scene"]] += 1
        ctx.close()
    assert total == {"builtin": 384, "c4": 16, "c5": 4}
    assert ok["builtin"] >= 383, ok              # the reference's scene: the north star's bar
    assert ok["c4"] >= 10 and ok["c5"] >= 1, ok  # sphere clouds (c5: 64 samples of depth-16 trees per pixel): see the module docstring


def test_fast_random_scenes_within_tolerance(torch_cuda):
    """Random scenes of all three kernel classes, moved eyes, ragged frames (the generator of the exact build's
    bit-identity test).  The reference's scene: the north star's bar over the pixels of all its cases together
    (frames are small), no case off in more than 2 % of its pixels.  Sphere clouds: SCENE_FRAC per case, a
    median error of rounding size, and the same picture (mean absolute difference)."""
    rng = np.random.default_rng(20081103)
    n_pix = n_lsb = n_rel = 0
    for case in range(100):
        cls = int(rng.integers(0, 4))
        if cls == 0:
            sc, desc = rt.Scene.builtin(), "builtin"
        else:
            n_sph = int([rng.integers(0, 40), rng.integers(40, 600), rng.integers(2049, 3200)][cls - 1])
            n_pol = int(rng.integers(0, 5)) or int(n_sph == 0)
            n_l = int(rng.choice([1, 1, 2, 3, 8, 33, 40])) if cls < 3 else int(rng.integers(1, 4))
            args = (int(rng.integers(1, 2**31)), n_sph, n_pol, n_l, int(rng.integers(0, 2)))
            sc, desc = rt.Scene.synthetic(*args), f"synthetic{args}"
        if rng.random() < 0.5:
            sc.c.eyePos.x += float(rng.normal(0, 2))
            sc.c.eyePos.y += float(rng.normal(0, 1))
            sc.c.eyePos.z += float(rng.normal(0, 6))
        big = cls == 3
        W, H = 2 * int(rng.integers(4, 30 if big else 80)), int(rng.integers(8, 40 if big else 120))  # even widths
        s, d = int(rng.integers(1, 4)), int(rng.integers(0, 11))
        cam = rt.prepare_camera(sc, W, H, s, d)
        ctx = fast_ctx(sc, queue_cap=1 << 21)
        argb, rad, _ = render_dev(torch_cuda, ctx, cam)
        ctx.close()
        a_o, r_o, _ = O.render(sc.c, cam, nthreads=16)
        lsb = (np.abs(rgb_bytes(argb) - rgb_bytes(a_o)).max(-1) <= 1)
        both_nan = np.isnan(rad) & np.isnan(r_o)  # (an eye moved onto a surface: 0/0 in the reference too)
        relerr = np.where(both_nan, 0.0, np.abs(rad - r_o) / np.maximum(np.abs(r_o), 1e-3)).max(-1)
        rel = relerr <= 1e-4
        if cls != 0:
            assert lsb.mean() >= 0.9 and rel.mean() >= SCENE_FRAC and np.median(relerr) <= 1e-4 and \
                np.where(both_nan, 0.0, np.abs(rad - r_o)).mean() <= 5e-3, f"case {case}: {W}x{H} s{s} d{d} {desc}: {lsb.mean()} {rel.mean()}"
            continue
        assert lsb.mean() >= 0.98 and rel.mean() >= 0.98, f"case {case}: {W}x{H} s{s} d{d} {desc}: {lsb.mean()} {rel.mean()}"
        n_pix += W * H
        n_lsb += int(lsb.sum())
        n_rel += int(rel.sum())
    assert n_pix > 50000 and n_lsb >= LSB_FRAC * n_pix and n_rel >= REL_FRAC * n_pix, (n_pix, n_lsb, n_rel)


def test_fast_polygon_scene_files_within_tolerance(torch_cuda, tmp_path, fast_env):
    """Random scene files with polygons in arbitrary planes (tests/scene_fuzz.py) through rt_load_scene and
    the host-pointer entry point."""
    from scene_fuzz import random_scene_text
    rng = np.random.default_rng(19800101)
    n_pix = n_lsb = n_rel = 0
    for case in range(60):
        text, kinds = random_scene_text(rng)
        path = tmp_path / f"s{case}.scene"
        path.write_text(text)
        sc = rt.load_scene(str(path))
        W, H = 2 * int(rng.integers(10, 70)), int(rng.integers(20, 100))
        s, d = int(rng.integers(1, 4)), int(rng.integers(0, 7))
        cam = rt.prepare_camera(sc, W, H, s, d)
        argb, rad = rt.render(sc, cam)
        a_o, r_o, _ = O.render(sc.c, cam, nthreads=16)
        lsb = (np.abs(rgb_bytes(argb) - rgb_bytes(a_o)).max(-1) <= 1)
        rel = (np.abs(rad - r_o) / np.maximum(np.abs(r_o), 1e-3)).max(-1) <= 1e-4
        assert lsb.mean() >= 0.98 and rel.mean() >= 0.98, f"case {case} {W}x{H} s{s} d{d} {kinds}: {lsb.mean()} {rel.mean()}\n{text}"
        n_pix += W * H
        n_lsb += int(lsb.sum())
        n_rel += int(rel.sum())
    assert n_lsb >= LSB_FRAC * n_pix and n_rel >= REL_FRAC * n_pix, (n_pix, n_lsb, n_rel)


def test_fast_mode_rejects_unknown_names_in_the_environment(monkeypatch):
    sc = rt.Scene.builtin()
    cam = rt.prepare_camera(sc, 32, 16, 1, 2)
    monkeypatch.setenv("RT_HIP_NUMERICS", "sloppy")
    with pytest.raises(rt.RtHipError):
        rt.render(sc, cam)
    monkeypatch.setenv("RT_HIP_NUMERICS", "exact")
    rt.render(sc, cam)
