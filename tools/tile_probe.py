"""development: RT_HIP_LIB=<a -DRT_TILE_CLOCKS build> python tools/tile_probe.py
waves and wave cycles of k_primary per class of tile (depth 0 render: only k_primary adds to the slots)."""
import sys; sys.path.insert(0, '.')
import torch, raytracer_amd as rt
sc = rt.Scene.builtin(); cam = rt.prepare_camera(sc, 1920, 1080, 4, 8)
ctx = rt.Context(); ctx.upload(sc); ctx.set_numerics(rt.NUMERICS_FAST)
fb = torch.zeros((1080, 1920), dtype=torch.int32, device='cuda')
for _ in range(3):
    ctx.render_rows(cam, fb.data_ptr(), 0, 0, 1080, 0); st = ctx.finish().as_dict()
names = ["sky (culled)", "one lit polygon, no spawn", "one polygon (shadowed)", "no spheres, other", "spheres in sight"]
fields = ["sphere_tests", "sphere_hits", "polygon_tests", "polygon_rej_parallel", "polygon_rej_behind", "polygon_full",
          "polygon_hits", "lights_lit", "nodes_sphere", "nodes_polygon"]
tot = sum(st[fields[2 * k + 1]] for k in range(5))
for k, n in enumerate(names):
    w, c = st[fields[2 * k]], st[fields[2 * k + 1]]
    print(f"{n:30s} waves {w:7d}  wave-cycles {c:12d} ({100 * c / max(tot, 1):5.1f} %)  cycles/wave {c / max(w, 1):9.0f}")
