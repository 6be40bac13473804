"""development: stage times of warm blocking raytrace() calls (RT_HIP_TIMES=1 on stderr).  usage: RT_HIP_NUMERICS=exact|fast python tools/warm_times.py"""
import os, sys, time
os.environ["RT_HIP_TIMES"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytracer_amd as rt
sc = rt.Scene.builtin()
W, H = 1920, 1080
buf = np.zeros(W * H, dtype=np.uint32)
ts = []
for i in range(25):
    t = time.perf_counter()
    rt.raytrace(buf, sc, 45.0, W / H, 1.0, 200.0, W, H, 4, 8)
    ts.append(time.perf_counter() - t)
print("warm calls ms:", " ".join("%.3f" % (x * 1e3) for x in ts[5:]))
