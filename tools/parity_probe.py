"""development: how close is config 2 to the oracle? exact-radiance fraction, differing 8-bit pixels"""
import sys; sys.path.insert(0, '.')
import numpy as np, raytracer_amd as rt
from oracle import oracle_py as O
W, H, s, d = (int(a) for a in sys.argv[1:5]) if len(sys.argv) > 4 else (1920, 1080, 4, 8)
sc = rt.Scene.builtin(); cam = rt.prepare_camera(sc, W, H, s, d)
argb, rad = rt.render(sc, cam)
import os
O.set_rounded_pow(os.environ.get("ROUNDED_POW") == "1")
a_o, r_o, _ = O.render(sc.c, cam, nthreads=16)
exact = (rad.view(np.uint32) == r_o.view(np.uint32)).all(-1)
diff8 = (argb != a_o)
rel = np.abs(rad - r_o) / np.maximum(np.abs(r_o), 1e-3)
print(f"{W}x{H} s{s} d{d}: radiance bit-exact pixels {exact.mean():.6f} ({(~exact).sum()} differ), 8-bit pixels differing {diff8.sum()}, max rel {rel.max():.3e}")
ulp = np.abs(rad.view(np.int32).astype(np.int64) - r_o.view(np.int32).astype(np.int64))
print("ulp distance histogram of differing channels:", np.bincount(np.minimum(ulp[ulp > 0], 8)))
ys, xs = np.nonzero(~exact)
for y, x in list(zip(ys, xs))[:12]:
    print(f"pixel x={x} y={y}: got {rad[y, x]} ({rad[y, x].view(np.uint32)}), want {r_o[y, x]} ({r_o[y, x].view(np.uint32)})")
