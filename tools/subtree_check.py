"""development: the second render of a frame (k_subtree path) against the first (one launch per level) on random scenes"""
import sys; sys.path.insert(0,'.')
import numpy as np, torch, raytracer_amd as rt
rng=np.random.default_rng(7)
used=0
for case in range(30):
    n_sph=int(rng.integers(1,600)); sc=rt.Scene.synthetic(int(rng.integers(1,2**31)), n_sph, int(rng.integers(1,4)), int(rng.integers(1,4)), int(rng.integers(0,2)))
    W,H,s,d=96,64,2,int(rng.integers(3,11))
    cam=rt.prepare_camera(sc,W,H,s,d); ctx=rt.Context(); ctx.upload(sc)
    res=[]
    for _ in range(2):
        a=torch.zeros((H,W),dtype=torch.int32,device='cuda'); r=torch.zeros((H,W,3),dtype=torch.float32,device='cuda')
        ctx.render_rows(cam,a.data_ptr(),r.data_ptr()); st=ctx.finish(); res.append((a,r,st.queued_rays,st.shade_nodes))
    assert torch.equal(res[0][0],res[1][0]) and torch.equal(res[0][1].view(torch.int32),res[1][1].view(torch.int32)) and res[0][3]==res[1][3]
    used += res[1][2] < res[0][2]
    ctx.close()
print("cases where the second render queued fewer hits:", used, "of 30")
