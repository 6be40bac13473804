"""development: RT_HIP_LIB=<a -DRT_TILE_TRACE build> python tools/tile_trace.py
start / end of every k_primary wave of config 2 (wall clock, 10 ns), by class of tile: who runs when."""
import sys, os, ctypes as C; sys.path.insert(0, '.')
import numpy as np, torch, raytracer_amd as rt
sc = rt.Scene.builtin(); cam = rt.prepare_camera(sc, 1920, 1080, 4, 8)
ctx = rt.Context(); ctx.upload(sc); ctx.set_numerics(rt.NUMERICS_FAST)
fb = torch.zeros((1080, 1920), dtype=torch.int32, device='cuda')
for _ in range(4):
    ctx.render_rows(cam, fb.data_ptr(), 0, 0, 1080, 0); ctx.finish()
torch.cuda.synchronize()
lib = C.CDLL(os.environ["RT_HIP_LIB"])
nw = int(os.environ.get('RT_TRACE_WAVES', 60 * 135 * 4))
a = np.zeros((nw, 4), dtype=np.uint64)
assert lib.rt_dev_tile_trace_read(a.ctypes.data_as(C.c_void_p), C.c_size_t(a.nbytes)) == 0
t0 = a[:, 0].astype(np.int64); t1 = a[:, 1].astype(np.int64); cls = a[:, 2].astype(np.int64)
hw = a[:, 3] & np.uint64(0xffffffff); xcc = (a[:, 3] >> np.uint64(32)) & np.uint64(0xf)
ok = a[:, 0] != 0
print('waves never recorded:', int((~ok).sum()), 'first', np.nonzero(~ok)[0][:8])
t0 = t0[ok]; t1 = t1[ok]; cls = cls[ok]; xcc = xcc[ok]; nw = int(ok.sum())
base = t0.min(); t0 -= base; t1 -= base
print(f"kernel span {t1.max() / 100:.1f} us, waves {nw}")
names = ["sky", "lit polygon", "polygon shadowed", "no spheres", "spheres", "second copy, left"]
for k, n in enumerate(names):
    m = cls == k
    if m.any():
        d = (t1 - t0)[m] / 100
        print(f"{n:18s} waves {m.sum():6d} start {t0[m].min() / 100:6.1f}..{t0[m].max() / 100:6.1f} us  last end {t1[m].max() / 100:6.1f}  "
              f"dur mean {d.mean():6.1f} p50 {np.median(d):6.1f} p95 {np.percentile(d, 95):6.1f} max {d.max():6.1f}")
print("t(us)  resident by class            started  ended")
step = 500
for b in range(0, int(t1.max()) + step, step):
    live = [(int(((t0 <= b) & (t1 > b) & (cls == k)).sum())) for k in range(6)]
    print(f"{b / 100:5.0f}  {live}  sum {sum(live):5d}   {int(((t0 >= b) & (t0 < b + step)).sum()):6d} {int(((t1 >= b) & (t1 < b + step)).sum()):6d}")
# the last 20 waves to end
o = np.argsort(t1)[-12:]
for w in o:
    print(f"late wave {w:6d} block ({(w // 4) % 60:2d},{(w // 4) // 60:3d}) cls {cls[w]} start {t0[w] / 100:6.1f} end {t1[w] / 100:6.1f} xcc {int(xcc[w])}")
# per-XCC end time
for x in range(8):
    m = xcc == x
    print(f"xcc {x}: waves {m.sum():6d} last end {t1[m].max() / 100:6.1f} us, wave-us {((t1 - t0)[m]).sum() / 100:9.0f}")
if nw != 60 * 135 * 4: sys.exit(0)
# map of the blocks' longest wave, one character per block (tens of us; '.' under 10)
dur = ((t1 - t0) / 100).reshape(135, 60, 4).max(axis=2)
st = (t0 / 100).reshape(135, 60, 4).min(axis=2)
print("block map: longest wave of the block, tens of us")
for y in range(135):
    if dur[y].max() >= 10:
        print(f"{y:3d} " + "".join('.' if d < 10 else (chr(ord('0') + int(d // 10)) if d < 100 else chr(ord('A') + int((d - 100) // 10))) for d in dur[y]) + f"  start {st[y].min():5.1f}")
