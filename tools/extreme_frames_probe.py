"""development: frames of extreme shapes (very wide, very tall with a band of rows near the end, one pixel, many samples a
pixel) against the oracle on the reference's scene and on config 4's: every pixel bit-identical (libm's powf rounding taken out)."""
import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch, raytracer_amd as rt
from oracle import oracle_py as O
from test_gpu_parity import render_dev
O.set_rounded_pow(True)
bad = 0
for name, sc in [("builtin", rt.Scene.builtin()), ("c4", rt.Scene.synthetic(seed=0xC0FFEE04, n_spheres=1024, n_polys=4, n_lights=1, kind=0))]:
    for (W, H, s, d, r0, n) in [(65536, 9, 1, 3, 0, 9), (100000, 3, 1, 2, 1, 2), (7, 200000, 1, 4, 199990, 10), (3, 50000, 2, 4, 24997, 11),
                                (1, 1, 1, 8, 0, 1), (2, 2, 16, 5, 0, 2), (1, 9, 31, 2, 3, 5), (40000, 16, 2, 1, 8, 8)]:
        cam = rt.prepare_camera(sc, W, H, s, d)
        ctx = rt.Context(); ctx.upload(sc)
        try:
            argb, rad, st = render_dev(torch, ctx, cam, r0, n)
        except rt.RtHipError as e:
            print("ERR", name, W, H, s, d, r0, n, str(e)[:100], flush=True); bad += 1; ctx.close(); continue
        ctx.close()
        a_o, r_o, _ = O.render(sc.c, cam, nthreads=16) if W * H * s * s <= 3_000_000 else (None, None, None)
        if a_o is None:   # too big for a whole oracle frame: the rows asked for, pixel by pixel, sampled
            xs = np.unique(np.linspace(0, W - 1, 300).astype(int)); ok = True
            for y in range(r0, r0 + n):
                for x in xs:
                    want = O.pixel(sc.c, cam, int(x), int(y))
                    ok &= bool((rad[y, x].view(np.uint32) == want.view(np.uint32)).all())
        else:
            ok = np.array_equal(rad[r0:r0 + n].view(np.uint32), r_o[r0:r0 + n].view(np.uint32)) and np.array_equal(argb[r0:r0 + n], a_o[r0:r0 + n])
        bad += not ok
        print("ok " if ok else "BAD", name, W, H, s, d, "rows", r0, n, flush=True)
print(bad, "bad")
sys.exit(1 if bad else 0)
