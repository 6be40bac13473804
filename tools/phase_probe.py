"""development: RT_HIP_LIB=tools/abl/librt_PHASE.so python tools/phase_probe.py [depth]
prints the wave cycles k_bounce spends per phase (a -DRT_PHASE_CLOCKS build reports them in the
statistic fields the plain kernels leave unused)."""
import sys; sys.path.insert(0, '.')
import torch, raytracer_amd as rt
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 1
sc = rt.Scene.builtin(); cam = rt.prepare_camera(sc, 1920, 1080, 4, depth)
ctx = rt.Context(); ctx.upload(sc); ctx.set_numerics(rt.NUMERICS_FAST)
fb = torch.zeros((1080, 1920), dtype=torch.int32, device='cuda')
for _ in range(2):
    ctx.render_rows(cam, fb.data_ptr(), 0, 0, 1080, 0); st = ctx.finish().as_dict()
names = ["0 load entry", "1 -", "2 pass A shadow", "3 N, V, pass B", "4 bg combine / to reserve", "5 reserve (atomic)",
         "6 node + queue stores", "7 leaf store / loop tail (+ material, colour of hits that spawn nothing)",
         "8 material, local colour, child rays built", "9 child searches"]
fields = ["sphere_tests", "sphere_hits", "polygon_tests", "polygon_rej_parallel", "polygon_rej_behind", "polygon_full",
          "polygon_hits", "lights_lit", "nodes_sphere", "nodes_polygon"]
rays = st["reflection_rays"] + st["transmission_rays"]
iters = rays / 64.0
tot = sum(st[f] for f in fields)
print("queued rays", rays, "wave-iterations ~", int(iters))
for n, f in zip(names, fields):
    print(f"{n:75s} {st[f]/max(iters,1):10.0f} cycles/wave-iteration  {100*st[f]/max(tot,1):5.1f} %")
