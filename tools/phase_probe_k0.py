"""development: RT_HIP_LIB=<a -DRT_PHASE_CLOCKS build> python tools/phase_probe_k0.py [row0 nrows]
wave cycles k_primary spends per phase of a sample (depth 0 render: no bounce kernels add to the slots)."""
import sys; sys.path.insert(0, '.')
import torch, raytracer_amd as rt
row0, nrows = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (0, 1080)
sc = rt.Scene.builtin(); cam = rt.prepare_camera(sc, 1920, 1080, 4, 1)
ctx = rt.Context(); ctx.upload(sc); ctx.set_numerics(rt.NUMERICS_FAST)
fb = torch.zeros((1080, 1920), dtype=torch.int32, device='cuda')
for _ in range(3):
    ctx.render_rows(cam, fb.data_ptr(), 0, row0, nrows, 0); st = ctx.finish().as_dict()
names = ["0 primary visibility", "1 -", "2 pass A shadow", "3 N, V, pass B", "4 bg combine / to reserve", "5 reserve (atomic)",
         "6 node + queue stores", "7 sample tail (roots, sums)", "8 material, local colour, child rays built", "9 child searches"]
fields = ["sphere_tests", "sphere_hits", "polygon_tests", "polygon_rej_parallel", "polygon_rej_behind", "polygon_full",
          "polygon_hits", "lights_lit", "nodes_sphere", "nodes_polygon"]
tot = sum(st[f] for f in fields)
print("rows", row0, nrows, "total wave-cycles", tot, " (level-1 kernel adds to the same slots: depth 1)")
for n, f in zip(names, fields):
    print(f"{n:45s} {st[f]:14d}  {100*st[f]/max(tot,1):5.1f} %")
