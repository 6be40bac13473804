import sys; sys.path.insert(0, '.')
import numpy as np, torch, raytracer_amd as rt
from oracle import oracle_py as O
sc = rt.Scene.synthetic(0xC0FFEE05, 16384, 0, 8, 1)
W, H, s, d = 1920, 1080, 2, 16
cam = rt.prepare_camera(sc, W, H, s, d)
ctx = rt.Context(); ctx.upload(sc)
fb = torch.zeros((H, W), dtype=torch.int32, device='cuda'); rad = torch.zeros((H, W, 3), dtype=torch.float32, device='cuda')
O.set_rounded_pow(True)
for (x, y) in [(1571, 114), (1190, 79), (308, 258)]:
    y0 = y // 8 * 8
    ctx.render_rows(cam, fb.data_ptr(), rad.data_ptr(), y0, 8, 0); ctx.finish()
    fast = rad[y, x].cpu().numpy().copy()
    ctx.render_stats(cam, fb.data_ptr(), rad.data_ptr(), y0, 8, 0)
    full = rad[y, x].cpu().numpy().copy()
    print((x, y), "early-exit walk", fast, "whole-line walk", full, "oracle", O.pixel(sc.c, cam, x, y), flush=True)
