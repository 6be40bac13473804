import sys, time; sys.path.insert(0, '.')
import torch, raytracer_amd as rt
sc = rt.Scene.synthetic(0xC0FFEE05, 16384, 0, 8, 1); W, H, s, d = 1920, 1080, 2, 16
cam = rt.prepare_camera(sc, W, H, s, d)
ctx = rt.Context(); ctx.upload(sc)
fb = torch.zeros((H, W), dtype=torch.int32, device='cuda')
for it in range(3):
    t0 = time.perf_counter()
    ctx.render_rows(cam, fb.data_ptr(), 0, 0, H, 0); st = ctx.finish().as_dict()
    dt = time.perf_counter() - t0
    print(f"c5 no polys {W}x{H} s{s} d{d}: {dt*1e3:.1f} ms", flush=True)
print("sky fraction", float((fb == fb[0,0]).float().mean()))
