"""development: k_primary of the reference's scene, the library's own kernels against the run-time compiled ones, alternating"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as rt
sc = rt.Scene.builtin(); W, H = 1920, 1080
cam = rt.prepare_camera(sc, W, H, 4, 8)
fb = torch.zeros((H, W), dtype=torch.int32, device="cuda")
st_ = torch.cuda.Stream()
ctxs = {}
for name, mode, spec in (("built-in", rt.JIT_OFF, True), ("run-time", rt.JIT_NOW_ANY, True), ("run-time, specialization off", rt.JIT_NOW_ANY, False), ("generic", rt.JIT_OFF, False)):
    c = rt.Context(); c.upload(sc); c.set_numerics(rt.NUMERICS_FAST); c.set_specialization(spec); c.set_jit(mode); ctxs[name] = c
    for _ in range(6):
        c.render_rows(cam, fb.data_ptr(), 0, 0, H, st_.cuda_stream); c.finish()
for rep in range(2):
    for name, c in ctxs.items():
        c.set_timing(2); ms = n = 0
        for _ in range(30):
            c.render_rows(cam, fb.data_ptr(), 0, 0, H, st_.cuda_stream); st = c.finish(); ms += st.k0_ms; n += st.k0_launches
        c.set_timing(0)
        print(f"{name}: k_primary {ms / n * 1e3:.1f} us  (planned launches {st.planned_launches})", flush=True)
