import sys; sys.path.insert(0,'.')
import numpy as np, raytracer_amd as rt
from oracle import oracle_py as O
sc=rt.Scene.builtin()
for d in (0,4):
    cam=rt.prepare_camera(sc,640,480,1,d)
    a,r=rt.render(sc,cam); ao,ro,_=O.render(sc.c,cam,nthreads=16)
    ne=(r!=ro).any(-1)
    ys,xs=np.nonzero(ne)
    print("depth",d,"inexact",ne.sum(),"of",ne.size)
    # classify by colour: floor pixels have b==0
    floor=(ro[...,2]==0)|((ro[...,1]==0))
    bg=np.isclose(ro[...,2],0.8705882)
    print(" inexact on floor",(ne&floor).sum(),"floor px",floor.sum()," on bg",(ne&bg).sum(), " other",(ne&~floor&~bg).sum(), "other px", (~floor&~bg).sum())
    k=np.argsort(-(np.abs(r-ro).max(-1)).ravel())[:5]
    for i in k:
        y,x=divmod(int(i),640); print("  ",x,y,r[y,x],ro[y,x])
