import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
import raytracer_amd as rt
from test_gpu_launch_plan import render_dev
fr = []
for seed in (976732691, 12345, 777, 31337, 4242, 99):
    sc = rt.Scene.synthetic(seed, 3091, 4, 2, 0)
    cam = rt.prepare_camera(sc, 27, 27, 4, 9)
    res = {}
    for name, fused in (("fused", True), ("jobs", False)):
        ctx = rt.Context(queue_cap=1 << 21); ctx.upload(sc); ctx.set_numerics(rt.NUMERICS_FAST)
        if fused:
            ctx.set_launch_plan(False); ctx.set_pipeline(rt.PIPELINE_FUSED)
        a, r, st = render_dev(torch, ctx, cam); res[name] = r; ctx.close()
    ex = rt.Context(queue_cap=1 << 21); ex.upload(sc); ex.set_numerics(rt.NUMERICS_EXACT)
    _, rx, _ = render_dev(torch, ex, cam); ex.close()
    def frac(a, b, tol): return (np.abs(a - b) <= tol * np.maximum(np.abs(b), 1e-3)).all(axis=2).mean()
    print(seed, "jobs vs fused (fast) 1e-4: %.3f  1e-2: %.3f | jobs-fast vs exact 1e-4: %.3f | fused-fast vs exact 1e-4: %.3f" % (
        frac(res["jobs"], res["fused"], 1e-4), frac(res["jobs"], res["fused"], 1e-2), frac(res["jobs"], rx, 1e-4), frac(res["fused"], rx, 1e-4)))
