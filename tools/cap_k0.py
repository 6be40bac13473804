"""development: k_primary of config 2 alone against the context's queue capacity (shard strides: powers of two or not)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as rt
sc = rt.Scene.builtin(); W, H = 1920, 1080
cam = rt.prepare_camera(sc, W, H, 4, 8)
fb = torch.zeros((H, W), dtype=torch.int32, device="cuda")
st_ = torch.cuda.Stream()
for cap in (0, 4 << 20, (4 << 20) + (64 << 10), 5 << 20, (5 << 20) + 64 * 7 * 37, 6 << 20, 8 << 20, (8 << 20) + (64 << 8), 16 << 20):
    c = rt.Context(queue_cap=cap); c.upload(sc); c.set_numerics(rt.NUMERICS_FAST)
    for _ in range(8):
        c.render_rows(cam, fb.data_ptr(), 0, 0, H, st_.cuda_stream); st = c.finish()
    res = []
    for rep in range(2):
        c.set_timing(2); ms = n = 0
        for _ in range(30):
            c.render_rows(cam, fb.data_ptr(), 0, 0, H, st_.cuda_stream); st = c.finish(); ms += st.k0_ms; n += st.k0_launches
        c.set_timing(0); res.append(ms / max(n, 1) * 1e3)
    print(f"cap {cap:>10d}: k_primary {res[0]:.1f} / {res[1]:.1f} us  work {st.work_mib} MiB batches {st.batches}", flush=True)
    c.close()
