"""development: random scene FILES with arbitrary polygons (tests/scene_fuzz.py) through rt_load_scene; GPU
against the oracle with libm's powf rounding taken out: every pixel bit-identical.  Exercises the
three-stage polygon test (edge-screen data for any plane, unscreenable polygons).
FUZZ_CLOUD=1: 130-600 small spheres on top (a grid, the job pipeline).
usage: python tools/fuzz_polygons.py [cases] [seed]"""
import os, sys, tempfile, time; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, raytracer_amd as rt
from oracle import oracle_py as O
from scene_fuzz import random_scene_text

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 50
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
O.set_rounded_pow(True)
tmp = tempfile.mkdtemp()
bad, cover = 0, []
t0 = time.time()
for case in range(cases):
    text, kinds = random_scene_text(rng, cloud=int(rng.choice([130, 250, 600])) if os.environ.get("FUZZ_CLOUD") else 0)
    path = os.path.join(tmp, f"s{case}.scene")
    open(path, "w").write(text)
    sc = rt.load_scene(path)
    W, H = int(rng.integers(20, 140)), int(rng.integers(20, 100))
    s, d = int(rng.integers(1, 4)), int(rng.integers(0, 7))
    cam = rt.prepare_camera(sc, W, H, s, d)
    argb, rad = rt.render(sc, cam)
    a_o, r_o, _ = O.render(sc.c, cam, nthreads=16)
    differ = (rad.view(np.uint32) != r_o.view(np.uint32)).any(-1)
    ok = not differ.any() and np.array_equal(argb, a_o)
    bad += not ok
    cover.append(float((np.abs(r_o - np.array([0.1, 0.1, 0.3], np.float32)).max(-1) > 1e-6).mean()))
    print(("ok " if ok else "BAD"), case, f"{W}x{H} s{s} d{d}", kinds, "" if ok else f"{differ.sum()} pixels differ, file {path}", flush=True)
print(f"{cases} cases, {bad} bad, {time.time() - t0:.1f} s; mean fraction of pixels that show an object {np.mean(cover):.3f}")
sys.exit(1 if bad else 0)
