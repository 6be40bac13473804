"""development: a row slab that starts off the 8-row raster, through the job pipeline and through the fused kernels.
usage: python tools/slab_probe.py <fuzz_big seed> <case>"""
import sys, os, tempfile; sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch, runpy
import raytracer_amd as rt
from test_gpu_parity import render_dev
seed, case = int(sys.argv[1]), int(sys.argv[2])
# the scene of that case: tools/fuzz_big.py writes it when asked to render from `case` on for one case
os.environ["FUZZ_DUMP"] = "1"
sys.argv = ["fuzz_big.py", str(case + 1), str(seed), str(case)]
try:
    runpy.run_path("tools/fuzz_big.py", run_name="__main__")
except SystemExit:
    pass
path = os.environ["FUZZ_LAST_SCENE"]
sc = rt.load_scene(path)
W, H, s = [int(x) for x in os.environ["FUZZ_LAST_FRAME"].split()][:3]
for d in (0, 1, 2, 7):
    cam = rt.prepare_camera(sc, W, H, s, d)
    out = {}
    for name in ("jobs", "fused"):
        ctx = rt.Context(queue_cap=1 << 22); ctx.upload(sc)
        if name == "fused":
            ctx.set_pipeline(rt.PIPELINE_FUSED)
        for (r0, n) in ((0, H), (6, 5), (8, 5), (6, 8)):
            a, r, st = render_dev(torch, ctx, cam, r0, n)
            out[(name, r0, n)] = r[r0:r0 + n].copy()
        ctx.close()
    for (r0, n) in ((6, 5), (8, 5), (6, 8)):
        full = out[("fused", 0, H)][r0:r0 + n]
        for name in ("jobs", "fused"):
            diff = (out[(name, r0, n)].view(np.uint32) != full.view(np.uint32)).any(-1)
            ys, xs = np.nonzero(diff)
            print("depth", d, "slab", (r0, n), name, "differs from the whole frame in", int(diff.sum()), "pixels", list(zip((ys + r0).tolist(), xs.tolist()))[:8])
    print("depth", d, "jobs whole frame == fused whole frame:", np.array_equal(out[("jobs", 0, H)].view(np.uint32), out[("fused", 0, H)].view(np.uint32)))
