"""development: one render call of more than 2^30 primary samples through the job pipeline (its sample indices have 30
bits: the call is split into batches) - config 4's scene at 16384 x 8192, 3 x 3 samples (1.2 G samples), depth 3 -
against the oracle on 200 random pixels, bit for bit."""
import sys, time; sys.path.insert(0, '.')
import numpy as np, torch, raytracer_amd as rt
from oracle import oracle_py as O
O.set_rounded_pow(True)
sc = rt.Scene.synthetic(seed=0xC0FFEE04, n_spheres=1024, n_polys=4, n_lights=1, kind=0)
W, H, s, d = 16384, 8192, 3, 3
cam = rt.prepare_camera(sc, W, H, s, d)
ctx = rt.Context(); ctx.upload(sc)
argb = torch.zeros((H, W), dtype=torch.int32, device="cuda")
rad = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
t0 = time.time()
ctx.render_rows(cam, argb.data_ptr(), rad.data_ptr(), 0, H, torch.cuda.current_stream().cuda_stream)
st = ctx.finish()
print("rendered in %.2f s: %d batches, %d retries, primary rays %d (= %d)" % (time.time() - t0, st.batches, st.overflow_retries, st.primary_rays, W * H * s * s), flush=True)
rng = np.random.default_rng(1)
xy = np.stack([rng.integers(0, W, 200), rng.integers(0, H, 200)], -1)
got = rad[torch.from_numpy(xy[:, 1]).cuda(), torch.from_numpy(xy[:, 0]).cuda()].cpu().numpy()
want = np.stack([O.pixel(sc.c, cam, int(x), int(y)) for x, y in xy])
bad = int((got.view(np.uint32) != want.view(np.uint32)).any(-1).sum())
print("pixels that differ:", bad, "of", len(xy))
ctx.close()
sys.exit(1 if bad or st.primary_rays != W * H * s * s else 0)
