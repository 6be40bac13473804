"""development: k_primary time of one build.  python tools/k0_time.py [fast|exact] [check]
`check` also renders with the exact build and prints the parity of the tolerance build against it."""
import sys, os, time, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import raytracer_amd as rt
mode = rt.NUMERICS_EXACT if "exact" in sys.argv[1:] else rt.NUMERICS_FAST
W, H, s, d = [int(v) for v in os.environ.get("K0_FRAME", "1920,1080,4,8").split(",")]
ROW0 = int(os.environ.get("K0_ROW0", "0"))
NROWS = int(os.environ.get("K0_NROWS", "0")) or None
sc = rt.Scene.builtin()
cam = rt.prepare_camera(sc, W, H, s, d)


def run(mode, frames):
    ctx = rt.Context(queue_cap=1 << 25)
    ctx.upload(sc)
    ctx.set_numerics(mode)
    argb = torch.zeros((H, W), dtype=torch.int32, device="cuda")
    rad = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
    for _ in range(30):
        ctx.render_rows(cam, argb.data_ptr(), rad.data_ptr(), ROW0, NROWS or H - ROW0); st = ctx.finish()
    ctx.set_timing(True)
    k0, n = 0.0, 0
    t0 = time.perf_counter()
    for _ in range(frames):
        ctx.render_rows(cam, argb.data_ptr(), rad.data_ptr(), ROW0, NROWS or H - ROW0); st = ctx.finish()
        k0 += st.k0_ms; n += st.k0_launches
    dt = (time.perf_counter() - t0) / frames
    rays = st.primary_rays + st.shadow_rays + st.reflection_rays + st.transmission_rays
    a, r = argb.cpu().numpy().view(np.uint32), rad.cpu().numpy()
    ctx.close()
    return k0 / max(n, 1), dt * 1e3, rays, a, r


k0, dt, rays, a1, r1 = run(mode, 100)
print(f"mode {mode}: k_primary {k0:.4f} ms, frame {dt:.4f} ms, rays {rays}, crc {zlib.crc32(a1.tobytes()):08x}")
if "check" in sys.argv[1:]:
    _, _, rays0, a0, r0 = run(rt.NUMERICS_EXACT, 3)
    b = lambda a: np.stack([(a >> 16) & 255, (a >> 8) & 255, a & 255], -1).astype(int)
    lsb = (np.abs(b(a1) - b(a0)).max(-1) <= 1).mean()
    rel = (np.abs(r1 - r0) / np.maximum(np.abs(r0), 1e-3)).max(-1)
    print(f"vs exact: within 1 LSB {lsb:.6f}, rel<=1e-4 {(rel <= 1e-4).mean():.6f} ({int((rel > 1e-4).sum())} off), rays {rays - rays0:+d}")
