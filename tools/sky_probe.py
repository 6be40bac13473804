import sys; sys.path.insert(0,'.')
import torch, raytracer_amd as rt
sc = rt.Scene.synthetic(5, 1, 0, 1, 0)
W,H,s,d = 1920,1080,4,8
cam = rt.prepare_camera(sc, W, H, s, d)
ctx = rt.Context(); ctx.upload(sc); ctx.set_timing(True)
fb = torch.zeros((H,W), dtype=torch.int32, device='cuda')
for i in range(5):
    ctx.render_rows(cam, fb.data_ptr()); st = ctx.finish()
    print("k0 ms", st.k0_ms, "nodes", st.shade_nodes, "sky frac", float((fb==fb[0,0]).float().mean()))
