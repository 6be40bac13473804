"""development: which random cloud overflows a context of a million queue entries on one tile row (for the test of
rt_hip_render's retry in tests/test_gpu_parity.py).  usage: python tools/bushy_probe.py"""
import sys, os; sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch, tempfile
import raytracer_amd as rt
from test_gpu_parity import render_dev, bushy_scene_text
for seed in range(60):
    p = os.path.join(tempfile.mkdtemp(), "s.scene"); open(p, "w").write(bushy_scene_text(seed))
    sc = rt.load_scene(p)
    cam = rt.prepare_camera(sc, 33, 16, 3, 8)
    ctx = rt.Context(queue_cap=1 << 20); ctx.upload(sc)
    try:
        a, r, st = render_dev(torch, ctx, cam)
        print(seed, "ok: nodes", st.shade_nodes, "batches", st.batches, "retries", st.overflow_retries, flush=True)
    except rt.RtHipError as e:
        print(seed, "ERROR", str(e)[:90], flush=True)
    ctx.close()
