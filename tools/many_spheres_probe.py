"""development: scenes of very many spheres (100 000 to 2.2 M - beyond the 2 M a cell chunk's 21-bit indices reach: the fused kernels' lists; config 5 has 16 384) on a small frame against the oracle:
every pixel bit-identical (libm's powf rounding taken out).  usage: python tools/many_spheres_probe.py"""
import sys, time; sys.path.insert(0, '.')
import numpy as np, raytracer_amd as rt
from oracle import oracle_py as O
O.set_rounded_pow(True)
bad = 0
for n_sph, n_l, kind, W, H, s, d in [(100000, 2, 1, 48, 32, 1, 6), (400000, 8, 1, 32, 20, 1, 4), (150000, 1, 0, 40, 24, 2, 5), (2200000, 2, 1, 16, 10, 1, 3)]:
    t0 = time.time()
    sc = rt.Scene.synthetic(1234 + n_sph, n_sph, 2, n_l, kind)
    cam = rt.prepare_camera(sc, W, H, s, d)
    argb, rad = rt.render(sc, cam)
    t1 = time.time()
    a_o, r_o, _ = O.render(sc.c, cam, nthreads=16)
    differ = (rad.view(np.uint32) != r_o.view(np.uint32)).any(-1)
    ok = not differ.any() and np.array_equal(argb, a_o)
    bad += not ok
    print("ok " if ok else "BAD", n_sph, "spheres", n_l, "lights", W, H, s, d, "differ", int(differ.sum()), "GPU call %.1f s, oracle %.1f s" % (t1 - t0, time.time() - t1), flush=True)
sys.exit(1 if bad else 0)
