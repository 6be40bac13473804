"""development: where the tolerance build differs from the exact one. python tools/fast_diff.py W H s d"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import raytracer_amd as rt
W, H, s, d = [int(x) for x in sys.argv[1:5]]
sc = rt.Scene.builtin() if len(sys.argv) < 6 else rt.Scene.synthetic(*[int(v) for v in sys.argv[5].split(",")])
cam = rt.prepare_camera(sc, W, H, s, d)
res = []
for mode in (0, 1):
    ctx = rt.Context(queue_cap=1 << 22); ctx.upload(sc); ctx.set_numerics(mode)
    argb = torch.zeros((H, W), dtype=torch.int32, device="cuda"); rad = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
    ctx.render_rows(cam, argb.data_ptr(), rad.data_ptr()); ctx.finish()
    res.append(rad.cpu().numpy()); ctx.close()
rel = (np.abs(res[1] - res[0]) / np.maximum(np.abs(res[0]), 1e-3)).max(-1)
bad = np.argwhere(rel > 1e-4)
print(len(bad), "pixels off of", W * H, "percentiles 50/90/99", np.percentile(rel, [50, 90, 99]))
for y, x in bad[:60]:
    print(x, y, res[0][y, x], res[1][y, x], (res[1][y, x] - res[0][y, x]) * s * s)
if len(bad) > 60:
    dl = np.abs(res[1] - res[0])[rel > 1e-4].max(-1) * s * s
    print("sample-equivalents histogram:", np.histogram(dl, bins=[0, 1e-3, 1e-2, 0.05, 0.2, 0.5, 1, 2, 16])[0])
    print("rows histogram:", np.histogram(bad[:, 0], bins=12, range=(0, H))[0])
    print("cols histogram:", np.histogram(bad[:, 1], bins=12, range=(0, W))[0])
    small = bad[(dl < 0.05)]
    for y, x in small[:25]:
        print("small", x, y, res[0][y, x], res[1][y, x])
