"""Ad-hoc GPU check: HIP path vs oracle on the builtin scene (development aid)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import raytracer_amd as rt
from oracle import oracle_py as O

def cmp(scene, W, H, s, d, label):
    cam = rt.prepare_camera(scene, W, H, s, d)
    t = time.time(); argb, rad = rt.render(scene, cam); tg = time.time() - t
    t = time.time(); a_o, r_o, c_o = O.render(scene.c, cam, nthreads=16); to = time.time() - t
    neq = (argb != a_o).sum()
    bytes_g = np.stack([(argb >> 16) & 255, (argb >> 8) & 255, argb & 255], -1).astype(int)
    bytes_o = np.stack([(a_o >> 16) & 255, (a_o >> 8) & 255, a_o & 255], -1).astype(int)
    within1 = (np.abs(bytes_g - bytes_o).max(-1) <= 1).mean()
    rel = np.abs(rad - r_o) / np.maximum(np.abs(r_o), 1e-3)
    exact = (rad == r_o).all(-1).mean()
    print(f"{label}: {W}x{H} s{s} d{d}: argb!= {neq}  within1LSB {within1:.6f}  rad exact {exact:.6f} "
          f"max rel {rel.max():.3e}  frac rel>1e-4 {(rel.max(-1) > 1e-4).mean():.2e}  gpu {tg:.3f}s cpu16 {to:.3f}s", flush=True)

sc = rt.Scene.builtin()
cmp(sc, 64, 48, 1, 0, "builtin")
cmp(sc, 64, 48, 1, 4, "builtin")
cmp(sc, 640, 480, 1, 4, "builtin")
cmp(sc, 640, 480, 2, 8, "builtin")
cmp(sc, 333, 217, 3, 5, "builtin")
s2 = rt.Scene.synthetic(0xC0FFEE04, 64, 4, 1, 0)
cmp(s2, 160, 120, 2, 4, "syn64")
s3 = rt.Scene.synthetic(0xC0FFEE05, 300, 4, 8, 1)
cmp(s3, 160, 120, 1, 6, "syn300x8L")
s4 = rt.Scene.synthetic(0xC0FFEE05, 3000, 4, 2, 1)
cmp(s4, 96, 64, 1, 3, "syn3000 tiled")
