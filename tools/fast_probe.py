"""development: the tolerance build against the exact build (= the reference, bit for bit) on full frames:
parity fractions, ray counts, k_primary time.  python tools/fast_probe.py [W H s d [scene]]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import raytracer_amd as rt
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import parity_report

W, H, s, d = [int(x) for x in (sys.argv[1:5] or [1920, 1080, 4, 8])]
name = sys.argv[5] if len(sys.argv) > 5 else "builtin"
sc = rt.Scene.builtin() if name == "builtin" else rt.Scene.synthetic(*[int(x) for x in name.split(",")])
cam = rt.prepare_camera(sc, W, H, s, d)
out = {}
for mode in (rt.NUMERICS_EXACT, rt.NUMERICS_FAST):
    ctx = rt.Context(queue_cap=1 << 25)
    ctx.upload(sc)
    ctx.set_numerics(mode)
    argb = torch.zeros((H, W), dtype=torch.int32, device="cuda")
    rad = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
    for _ in range(3):
        ctx.render_rows(cam, argb.data_ptr(), rad.data_ptr()); st = ctx.finish()
    ctx.set_timing(True)
    k0, n = 0.0, 0
    t0 = time.perf_counter()
    for _ in range(20):
        ctx.render_rows(cam, argb.data_ptr(), rad.data_ptr()); st = ctx.finish()
        k0 += st.k0_ms; n += st.k0_launches
    dt = (time.perf_counter() - t0) / 20
    rays = st.primary_rays + st.shadow_rays + st.reflection_rays + st.transmission_rays
    out[mode] = (argb.cpu().numpy().view(np.uint32), rad.cpu().numpy(), rays)
    print(f"mode {mode}: k_primary {k0 / max(n, 1):.4f} ms x{n // 20}, frame {dt * 1e3:.3f} ms, rays {rays}, "
          f"shade {st.shade_nodes} refl {st.reflection_rays} trans {st.transmission_rays}")
    ctx.close()
a0, r0, n0 = out[0]
a1, r1, n1 = out[1]
lsb, rel, med, exact = parity_report(a1, r1, a0, r0)
relerr = (np.abs(r1 - r0) / np.maximum(np.abs(r0), 1e-3)).max(-1)
print(f"fast vs exact: within 1 LSB {lsb:.6f}, rel<=1e-4 {rel:.6f}, median rel {med:.3g}, bit-equal {exact:.4f}, "
      f"pixels off {int((relerr > 1e-4).sum())}, rays {n1 - n0:+d} ({abs(n1 - n0) / n0:.2e})")
print("rel error percentiles 50/90/99/99.9:", np.percentile(relerr, [50, 90, 99, 99.9]))
