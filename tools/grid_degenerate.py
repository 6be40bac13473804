import sys; sys.path.insert(0,'.')
import numpy as np, raytracer_amd as rt, ctypes as C
from oracle import oracle_py as O
# a degenerate big scene: 300 large overlapping spheres clustered in a small region + a few far away
sc = rt.Scene.synthetic(123, 300, 2, 2, 0)
rng = np.random.default_rng(1)
for i in range(sc.c.nObjects):
    o = sc.c.objects[i]
    if o.geometryType == 1:
        o.sphr_obj.center.x = float(rng.normal(0, 0.3)); o.sphr_obj.center.y = float(2 + rng.normal(0, 0.3)); o.sphr_obj.center.z = float(-10 + rng.normal(0, 0.3))
        o.sphr_obj.radius = float(rng.uniform(0.5, 3.0))
O.set_rounded_pow(True)
cam = rt.prepare_camera(sc, 96, 64, 1, 4)
argb, rad = rt.render(sc, cam)
a_o, r_o, _ = O.render(sc.c, cam, nthreads=16)
print("degenerate cluster: pixels that differ", int((rad.view(np.uint32) != r_o.view(np.uint32)).any(-1).sum()))
