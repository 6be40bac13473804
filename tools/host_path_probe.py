"""development: per-frame time of the host-buffer entry points (PCIe-inclusive), config 2"""
import sys, time; sys.path.insert(0, '.')
import numpy as np, raytracer_amd as rt
sc = rt.Scene.builtin(); W, H = 1920, 1080
buf = np.zeros(W * H, dtype=np.uint32)
for name, fn in (("raytrace()", lambda: rt.raytrace(buf, sc, 45.0, W / H, 1.0, 200.0, W, H, 4, 8)),):
    fn(); fn()
    t0 = time.perf_counter(); n = 20
    for _ in range(n): fn()
    dt = (time.perf_counter() - t0) / n
    print(f"{name}: {dt*1e3:.3f} ms per frame = {57693197/dt/1e6:.0f} Mrays/s (host frame buffer, blocking, D2H included)")
