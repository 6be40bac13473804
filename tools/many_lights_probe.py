"""development: scenes of 65 to 1000 lights (scene files; rt_scene_synthetic stops at 64) with 5, 60, 300 and 2500 spheres against the oracle:\nevery pixel bit-identical (libm's powf rounding taken out).  Beyond 32 lights the big scenes leave the job pipeline for the fused kernels."""
import sys, time, tempfile, os; sys.path.insert(0, '.')
import numpy as np, raytracer_amd as rt
from oracle import oracle_py as O
O.set_rounded_pow(True)
rng = np.random.default_rng(3)
bad = 0
tmp = tempfile.mkdtemp()
for n_sph, n_l in [(5, 65), (5, 100), (5, 257), (300, 65), (300, 130), (2500, 65), (2500, 200), (60, 1000)]:
    parts = ["desc\n{\n\t%d\n\t%d\n}\n" % (n_sph + 1, n_l),
             "camera\n{\nposition 0 4.5 2\nlookat 0 4.5 -8\nupvector 0 1 0\nprojection 45.0 1.0 200.0\n}\n",
             "world\n{\nambient 1.0 1.0 1.0\nbackground .3 .4 .6\n}\n"]
    for l in range(n_l):
        parts.append("pointlight\n{\nposition %r %r %r\ndiffuse %g %g %g\nrange 150\n}\n"
                     % (float(rng.uniform(-10, 10)), float(rng.uniform(2, 14)), float(rng.uniform(-16, 4)), *(rng.uniform(0.3, 1, 3) / n_l * 3)))
    for k in range(n_sph):
        x, y, z = rng.uniform(-3, 3), 4.5 + rng.uniform(-3, 3), -8.0 + rng.uniform(-3, 3)
        kr, kt = [(0.0, 0.0), (0.5, 0.0), (0.05, 0.8), (0.3, 0.5)][int(rng.integers(0, 4))]
        parts.append("sphere\n{\nname S%d\nposition %r %r %r\nradius %r\ndiffuse %g %g %g\nspecular 1 1 1\n"
                     "ka 0.1 kd 0.4 ks 0.3 ke %d kr %g kt %g n %g\n}\n"
                     % (k, x, y, z, float(rng.uniform(0.05, 0.5)), *rng.uniform(0.2, 1, 3), 20, kr, kt, 1.25))
    parts.append("polygon\n{\nname P\npoint 20 0 12\npoint 20 0 -28\npoint -20 0 -28\npoint -20 0 12\ndiffuse 0 1 0\nka 0.1 kd 0.6 ks 0.2 ke 2 kr 0.4 kt 0 n 1\n}\n")
    path = os.path.join(tmp, "l.scene"); open(path, "w").write("".join(parts))
    sc = rt.load_scene(path)
    cam = rt.prepare_camera(sc, 40, 28, 2, 4)
    try:
        argb, rad = rt.render(sc, cam)
    except rt.RtHipError as e:
        print("refused", n_sph, n_l, str(e)[:120], flush=True); continue
    a_o, r_o, _ = O.render(sc.c, cam, nthreads=16)
    differ = (rad.view(np.uint32) != r_o.view(np.uint32)).any(-1)
    ok = not differ.any() and np.array_equal(argb, a_o)
    bad += not ok
    print("ok " if ok else "BAD", n_sph, "spheres", n_l, "lights differ", int(differ.sum()), flush=True)
sys.exit(1 if bad else 0)
