import sys; sys.path.insert(0,'.')
import torch, raytracer_amd as rt
sc=rt.Scene.builtin(); cam=rt.prepare_camera(sc,1920,1080,4,8)
ctx=rt.Context(); ctx.upload(sc)
fb=torch.zeros((1080,1920),dtype=torch.int32,device='cuda')
st=ctx.render_stats(cam,fb.data_ptr(),0,0,1080,0).as_dict()
for k in ("polygon_tests","polygon_full","polygon_hits","polygon_screen_unsure","polygon_screen_mismatch","k0_polygon_full","sphere_tests","sphere_hits","shade_nodes","k0_shade_nodes"): print(k, st[k])
