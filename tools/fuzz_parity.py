"""development: random scenes / cameras / frame shapes, GPU against the oracle with libm's powf rounding
taken out (rt_oracle_set_rounded_pow): every pixel's radiance must be bit-identical.
usage: python tools/fuzz_parity.py [cases] [seed]"""
import sys, time; sys.path.insert(0, '.')
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
    big = cls == 3
    W, H = int(rng.integers(1, 40 if big else 110)), int(rng.integers(1, 30 if big else 80))
    s, d = int(rng.integers(1, 4)), int(rng.integers(0, 11))
    cam = rt.prepare_camera(sc, W, H, s, d)
    argb, rad = rt.render(sc, cam)
    a_o, r_o, _ = O.render(sc.c, cam, nthreads=16)
    differ = (rad.view(np.uint32) != r_o.view(np.uint32)).any(-1)
    tag = "ok " if not differ.any() and np.array_equal(argb, a_o) else "BAD"
    if tag == "BAD":
        bad += 1
        ys, xs = np.nonzero(differ)
        rel = np.abs(rad - r_o) / np.maximum(np.abs(r_o), 1e-3)
        desc += f"  -> {differ.sum()} pixels differ, max rel {rel.max():.3e}, first (x={xs[0]}, y={ys[0]}) got {rad[ys[0], xs[0]]} want {r_o[ys[0], xs[0]]}"
    print(f"{tag} {c:3d} {W}x{H} s{s} d{d} {desc}", flush=True)
print(f"{cases} cases, {bad} bad, {time.time() - t0:.1f} s")
sys.exit(1 if bad else 0)
