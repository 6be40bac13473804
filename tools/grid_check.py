"""development: random rays through the grid of a big scene against the loop over every sphere
(rt_hip_selftest_grid); prints mismatching rays.  usage: python tools/grid_check.py [nrays] [seed]"""
import sys; sys.path.insert(0, '.')
import numpy as np, torch, raytracer_amd as rt
from raytracer_amd import _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
sc = rt.Scene.synthetic(0xC0FFEE05, 16384, 0, 8, 1)
ctx = rt.Context(); ctx.upload(sc)
bad_total = 0
for filtered in (0, 1):
    # origins: inside the cloud, outside it, on sphere surfaces; directions: random, axis-parallel, towards spheres
    o = np.stack([rng.uniform(-16, 16, n), rng.uniform(-3, 12, n), rng.uniform(-70, 6, n)], 1).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    ax = rng.random(n) < 0.05
    d[ax] = np.eye(3, dtype=np.float32)[rng.integers(0, 3, ax.sum())] * rng.choice([-1, 1], ax.sum())[:, None].astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    scale = rng.choice([1.0, 1.0, 1.0, 0.9, 1.08, 1.0 + 3e-7], n)[:, None]   # the reference's refraction makes non-unit rays
    d = d * scale
    mag = rng.choice([9999999.9, 99999.9, 30.0, 5.0], n).astype(np.float32)
    rays = torch.from_numpy(np.concatenate([o, d.astype(np.float32), mag[:, None]], 1).astype(np.float32)).cuda().contiguous()
    out = torch.zeros((n, 4), dtype=torch.int32, device='cuda')
    _lib.check(_lib.load().rt_hip_selftest_grid(ctx._h, rays.data_ptr(), n, filtered, out.data_ptr()), "selftest_grid")
    r = out.cpu().numpy().view(np.uint32)
    bad = (r[:, 0] != r[:, 2]) | (r[:, 1] != r[:, 3])
    hits = (r[:, 2] != 0xFFFFFFFF).mean()
    print(f"filtered={filtered}: {bad.sum()} of {n} rays differ; {hits:.3f} of the rays hit something", flush=True)
    for i in np.nonzero(bad)[0][:6]:
        print("  ray", rays[i].cpu().numpy().tolist(), "grid", r[i, 0], r[i, 1:2].view(np.float32), "loop", r[i, 2], r[i, 3:4].view(np.float32))
    bad_total += int(bad.sum())
sys.exit(1 if bad_total else 0)
