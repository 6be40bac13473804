"""development: the host-pointer path with the frame split over several contexts on the SAME GPU (device list
[0,0,...]): the tails of some parts overlap the primary kernels of others."""
import sys, time; sys.path.insert(0, '.')
import numpy as np, raytracer_amd as rt
sc = rt.Scene.builtin(); W, H = 1920, 1080
cam = rt.prepare_camera(sc, W, H, 4, 8)
for devs in (None, [0], [0, 0], [0, 0, 0], [0, 0, 0, 0]):
    for _ in range(3):
        rt.render(sc, cam, want_radiance=False, devices=devs)
    t0 = time.perf_counter(); n = 20
    for _ in range(n):
        rt.render(sc, cam, want_radiance=False, devices=devs)
    dt = (time.perf_counter() - t0) / n
    print(f"devices {devs}: {dt*1e3:.3f} ms per frame (host buffers, D2H included)", flush=True)
