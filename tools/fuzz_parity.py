"""development: random scenes / cameras / frame shapes, GPU against the oracle with libm's powf rounding
taken out (rt_oracle_set_rounded_pow): every pixel's radiance must be bit-identical.
usage: python tools/fuzz_parity.py [cases] [seed]"""
import os, sys, time; sys.path.insert(0, '.')
import numpy as np, raytracer_amd as rt
from oracle import oracle_py as O

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
O.set_rounded_pow(True)
bad = 0
t0 = time.time()
for c in range(cases):
    kind = int(rng.integers(0, 2))
    cls = int(rng.integers(0, 4))
    if cls == 0:
        sc, desc = rt.Scene.builtin(), "builtin"
    else:
        n_sph = int([rng.integers(0, 40), rng.integers(40, 600), rng.integers(2049, 3200)][cls - 1])
        n_pol, n_l = int(rng.integers(0, 5)), int(rng.choice([1, 1, 2, 3, 8, 33, 40]))
        if cls == 3:
            n_l = min(n_l, 3)
        if n_sph + n_pol == 0:
            n_pol = 1
        seed = int(rng.integers(1, 2**31))
        sc, desc = rt.Scene.synthetic(seed, n_sph, n_pol, n_l, kind), f"syn seed={seed} sph={n_sph} poly={n_pol} lights={n_l} kind={kind}"
    if rng.random() < 0.1:
        sc.c.nLights = 0; desc += " (lights switched off)"
    if rng.random() < 0.5:   # move the eye: rays that start inside or behind objects
        sc.c.eyePos.x += float(rng.normal(0, 2)); sc.c.eyePos.y += float(rng.normal(0, 1)); sc.c.eyePos.z += float(rng.normal(0, 6))
        desc += f" eye=({sc.c.eyePos.x:.3f},{sc.c.eyePos.y:.3f},{sc.c.eyePos.z:.3f})"
    if os.environ.get("FUZZ_WILD"):  # what no modeller would write: indices of 0 / -1 / 1e-30 / 1e30, gains outside [0, 1], exponents of 0
        for i in range(sc.c.nObjects):   # and 1, negative colours, spheres of radius 0 or less, spheres twice, a sphere around the eye
            o = sc.c.objects[i]; m = o.material
            if rng.random() < 0.3:
                m.n = float(rng.choice([0.0, -1.0, 1e-30, 1e30, 1e-12, 3e-19, 0.5, 3.0]))
            if rng.random() < 0.2:
                m.kr = float(rng.choice([-0.5, 1.5, 0.3])); m.kt = float(rng.choice([-0.5, 2.0, 0.6, 0.0]))
            if rng.random() < 0.2:
                m.phong_ke = float(rng.choice([0.0, 1.0, 3.0, 200.0]))
            if rng.random() < 0.1:
                m.colors[0].r = -0.5; m.colors[1].g = 3.0
            if o.geometryType == 1:
                u = rng.random()
                if u < 0.05:
                    o.sphr_obj.radius = 0.0
                elif u < 0.1:
                    o.sphr_obj.radius = -o.sphr_obj.radius
                elif u < 0.15 and i > 0 and sc.c.objects[i - 1].geometryType == 1:
                    q = sc.c.objects[i - 1].sphr_obj
                    o.sphr_obj.center.x, o.sphr_obj.center.y, o.sphr_obj.center.z, o.sphr_obj.radius = q.center.x, q.center.y, q.center.z, q.radius
                elif u < 0.17:
                    o.sphr_obj.center.x, o.sphr_obj.center.y, o.sphr_obj.center.z = sc.c.eyePos.x, sc.c.eyePos.y, sc.c.eyePos.z
        desc += " (wild)"
    big = cls == 3
    W, H = int(rng.integers(1, 40 if big else 110)), int(rng.integers(1, 30 if big else 80))
    s, d = int(rng.integers(1, 4)), int(rng.integers(0, 11))
    cam = rt.prepare_camera(sc, W, H, s, d)
    if os.environ.get("FUZZ_BLINN"):  # the reference's unused second shading routine (rt_hip_ctx_set_shading)
        import torch
        sys.path.insert(0, "tests")
        from test_gpu_parity import render_dev
        ctx = rt.Context(); ctx.upload(sc); ctx.set_shading(rt.SHADE_PHONGBLINN)
        argb, rad, _ = render_dev(torch, ctx, cam)
        ctx.close()
        O.set_shading(1)
    else:
        argb, rad = rt.render(sc, cam)
    a_o, r_o, _ = O.render(sc.c, cam, nthreads=16)
    differ = ((rad.view(np.uint32) != r_o.view(np.uint32)) & ~(np.isnan(rad) & np.isnan(r_o))).any(-1)
    tag = "ok " if not differ.any() and np.array_equal(argb, a_o) else "BAD"
    if tag == "BAD":
        bad += 1
        ys, xs = np.nonzero(differ)
        rel = np.abs(rad - r_o) / np.maximum(np.abs(r_o), 1e-3)
        desc += f"  -> {differ.sum()} pixels differ, max rel {rel.max():.3e}, first (x={xs[0]}, y={ys[0]}) got {rad[ys[0], xs[0]]} want {r_o[ys[0], xs[0]]}"
    print(f"{tag} {c:3d} {W}x{H} s{s} d{d} {desc}", flush=True)
print(f"{cases} cases, {bad} bad, {time.time() - t0:.1f} s")
sys.exit(1 if bad else 0)
