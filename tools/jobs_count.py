#!/usr/bin/env python3
"""development: the job kernels' counters of a -DRT_JOBS_COUNT build (RT_HIP_LIB=...): per kind of ray, iterations of the
waves' loops, walking lanes per iteration, jobs per refill, rays that took the loop over every sphere.
usage: RT_HIP_LIB=raytracer_amd/librt_hip_b.so python tools/jobs_count.py c5s|c4"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as rt
from bench import CONFIGS
cfg = sys.argv[1]
scene_args, W, H, s, d, _, _ = CONFIGS[cfg]
sc = rt.Scene.synthetic(*scene_args)
cam = rt.prepare_camera(sc, W, H, s, d)
ctx = rt.Context()
ctx.upload(sc)
fb = torch.zeros((H, W), dtype=torch.int32, device="cuda")
for _ in range(2):
    ctx.render_rows(cam, fb.data_ptr(), 0, 0, H, 0)
    st = ctx.finish()
for name, f in (("shadow", ("sphere_tests", "sphere_hits", "polygon_tests", "polygon_rej_parallel", "polygon_rej_behind")),
                ("child", ("polygon_full", "polygon_hits", "lights_lit", "nodes_sphere", "nodes_polygon"))):
    it, act, refill, jobs, loop = (getattr(st, k) for k in f)
    print(f"{cfg} {name}: wave iterations {it:.3e}, walking lanes per iteration {act / max(it, 1):.1f}, refills {refill:.3e} "
          f"({jobs / max(refill, 1):.1f} jobs each), jobs {jobs:.3e} ({it * 64 / max(jobs, 1):.1f} lane-iterations a job), loop rays {loop:.3e}")
for name, f in (("shadow", ("k0_sphere_tests", "k0_sphere_hits", "k0_polygon_tests", "k0_polygon_rej_parallel")),
                ("child", ("k0_polygon_rej_behind", "k0_polygon_full", "k0_lights_lit", "k0_nodes_sphere"))):
    blocks, lanes, listed, tests = (getattr(st, k) for k in f)
    print(f"{cfg} {name}: sphere tests {tests:.3e} (lane level), distance blocks run {blocks:.3e} with {lanes / max(blocks, 1):.1f} lanes each, "
          f"lane-iterations in a listed cell {listed:.3e}")
if os.environ.get("JOBS_TIMES"):  # a -DRT_JOBS_COUNT -DRT_JOBS_TIMES build: the second group of slots holds ticks per phase
    for name, f in (("shadow", ("k0_sphere_tests", "k0_sphere_hits", "k0_polygon_tests", "k0_polygon_rej_parallel")),
                    ("child", ("k0_polygon_rej_behind", "k0_polygon_full", "k0_lights_lit", "k0_nodes_sphere"))):
        fl, so, bu, stp = (getattr(st, k) for k in f)
        tot = max(fl + so + bu + stp, 1)
        print(f"{cfg} {name}: wave time by phase: results + refill {fl / tot:.2f}, sorting out jobs {so / tot:.2f}, building rays {bu / tot:.2f}, walking {stp / tot:.2f}")
print("rays", st.primary_rays, st.shadow_rays, st.reflection_rays, st.transmission_rays, "shade nodes", st.shade_nodes)
