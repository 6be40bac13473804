"""development: time a full frame of BASELINE config 4 (or a scaled-down config 5) and print the batching /
overflow statistics.  usage: python tools/config_probe.py c4|c5 [W H sqrtSpp depth]"""
import sys, time; sys.path.insert(0, '.')
import torch, raytracer_amd as rt
which = sys.argv[1] if len(sys.argv) > 1 else "c4"
if which == "c4":
    sc = rt.Scene.synthetic(0xC0FFEE04, 1024, 4, 1, 0); W, H, s, d = 3840, 2160, 4, 8
else:
    sc = rt.Scene.synthetic(0xC0FFEE05, 16384, 4, 8, 1); W, H, s, d = 1920, 1080, 2, 16
if len(sys.argv) > 5:
    W, H, s, d = map(int, sys.argv[2:6])
cam = rt.prepare_camera(sc, W, H, s, d)
ctx = rt.Context(); ctx.upload(sc)
fb = torch.zeros((H, W), dtype=torch.int32, device='cuda')
for it in range(2):
    t0 = time.perf_counter()
    ctx.render_rows(cam, fb.data_ptr(), 0, 0, H, 0); st = ctx.finish().as_dict()
    dt = time.perf_counter() - t0
    rays = st["primary_rays"] + st["shadow_rays"] + st["reflection_rays"] + st["transmission_rays"]
    print(f"{which} {W}x{H} s{s} d{d}: {dt*1e3:.1f} ms, {rays/dt/1e6:.0f} Mrays/s, rays {rays}, batches {st['batches']}, "
          f"overflow retries {st['overflow_retries']}, nodes {st['tree_nodes']}, queued hits {st['queued_rays']}", flush=True)
