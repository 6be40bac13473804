"""development: the instrumented render's counters (rt_hip_render_stats) against the oracle's on random scenes of all
three kernel classes: rays by kind and shade nodes always; sphere / polygon tests and hits where the scene is small
enough to be searched by the reference's own loop (grids test fewer spheres, that being their point); pixels too.
usage: python tools/fuzz_counts.py [cases] [seed]"""
import sys, time; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch, raytracer_amd as rt
from oracle import oracle_py as O
from test_gpu_parity import render_dev
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
O.set_rounded_pow(True)
bad = 0
t0 = time.time()
for c in range(cases):
    kind = int(rng.integers(0, 2)); cls = int(rng.integers(0, 4))
    if cls == 0:
        sc, desc, n_sph = rt.Scene.builtin(), "builtin", 2
    else:
        n_sph = int([rng.integers(0, 40), rng.integers(40, 600), rng.integers(2049, 3200)][cls - 1])
        n_pol, n_l = int(rng.integers(0, 5)), int(rng.choice([1, 1, 2, 3, 8]))
        if n_sph + n_pol == 0:
            n_pol = 1
        seed = int(rng.integers(1, 2**31))
        sc, desc = rt.Scene.synthetic(seed, n_sph, n_pol, n_l, kind), f"syn seed={seed} sph={n_sph} poly={n_pol} lights={n_l} kind={kind}"
    W, H = int(rng.integers(1, 60)), int(rng.integers(1, 40))
    s, d = int(rng.integers(1, 4)), int(rng.integers(0, 9))
    cam = rt.prepare_camera(sc, W, H, s, d)
    ctx = rt.Context(); ctx.upload(sc)
    argb, rad, st = render_dev(torch, ctx, cam, stats=True)
    ctx.close()
    a_o, r_o, co = O.render(sc.c, cam, nthreads=16)
    keys = ["primary_rays", "shadow_rays", "reflection_rays", "transmission_rays", "shade_nodes"]
    if n_sph <= 128:
        keys += ["sphere_tests", "sphere_hits", "polygon_tests", "polygon_hits"]
    wrong = [(k, getattr(st, k), co[k]) for k in keys if getattr(st, k) != co[k]]
    pix = np.array_equal(rad.view(np.uint32), r_o.view(np.uint32)) and np.array_equal(argb, a_o)
    ok = not wrong and pix
    bad += not ok
    print("ok " if ok else "BAD", c, desc, f"{W}x{H} s{s} d{d}", "pixels same" if pix else "PIXELS DIFFER", wrong, flush=True)
print(f"{cases} cases, {bad} bad, {time.time() - t0:.1f} s")
sys.exit(1 if bad else 0)
