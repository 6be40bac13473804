#!/usr/bin/env python3
"""development: a case of tests/test_gpu_launch_plan.py::test_random_scenes_with_every_block_a_pair rendered several
times on one context - are the frames of the job pipeline deterministic, and where do they differ?
usage: python tools/jobs_probe.py <case> [out.npy]   (RT_HIP_NO_JOBS=1: the fused kernels, for comparison)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import raytracer_amd as rt
from test_gpu_parity import render_dev
want = int(sys.argv[1])
rng = np.random.default_rng(20261111)
for case in range(want + 1):
    cls = int(rng.integers(0, 4))
    if cls == 0:
        sc, desc = rt.Scene.builtin(), "builtin"
    else:
        n_sph = int([rng.integers(0, 40), rng.integers(40, 600), rng.integers(2049, 3200)][cls - 1])
        n_pol = int(rng.integers(0, 5)) or int(n_sph == 0)
        n_l = int(rng.choice([1, 1, 2, 3, 8, 33])) if cls < 3 else int(rng.integers(1, 4))
        args = (int(rng.integers(1, 2**31)), n_sph, n_pol, n_l, int(rng.integers(0, 2)))
        sc, desc = rt.Scene.synthetic(*args), f"synthetic{args}"
    if rng.random() < 0.1:
        sc.c.nLights = 0
    if rng.random() < 0.5:
        sc.c.eyePos.x += float(rng.normal(0, 2))
        sc.c.eyePos.y += float(rng.normal(0, 1))
        sc.c.eyePos.z += float(rng.normal(0, 6))
    big = cls == 3
    W, H = int(rng.integers(1, 60 if big else 150)), int(rng.integers(1, 40 if big else 100))
    s, d = int(rng.choice([1, 2, 2, 3, 4, 4])), int(rng.integers(0, 11))
    numerics = rt.NUMERICS_FAST if rng.random() < 0.3 else rt.NUMERICS_EXACT
print("case", want, desc, W, H, s, d, "numerics", numerics, "eye", sc.c.eyePos.x, sc.c.eyePos.y, sc.c.eyePos.z)
d = int(os.environ.get("PROBE_DEPTH", d))
cam = rt.prepare_camera(sc, W, H, s, d)
ctx = rt.Context(queue_cap=1 << 21)
ctx.upload(sc)
ctx.set_numerics(numerics)
frames = []
for k in range(6):
    a, r, st = render_dev(torch, ctx, cam)
    frames.append(r.copy())
    print(k, "rays", st.primary_rays, st.shade_nodes, st.reflection_rays, st.transmission_rays, "retries", st.overflow_retries, "nodes", st.tree_nodes, "queued", st.queued_rays)
for k in range(1, 6):
    diff = np.argwhere((frames[k].view(np.uint32) != frames[0].view(np.uint32)).any(axis=2))
    print("frame", k, "differs from frame 0 in", len(diff), "pixels", diff[:5].tolist(),
          [(frames[0][y, x].tolist(), frames[k][y, x].tolist()) for y, x in diff[:2]])
if len(sys.argv) > 2:
    np.save(sys.argv[2], frames[0])
