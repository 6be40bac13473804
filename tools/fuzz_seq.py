"""development: what a context keeps between frames (level sizes, the launch plan, the captured graph) must never show
in the pixels.  One context per case renders 4-7 frames of ONE shape - the eye moving, standing still, coming back, the
frame whole or as a band of rows - and every frame is compared with the oracle (libm's powf rounding taken out): every
pixel bit-identical.  Scenes of all three kernel classes (tools/fuzz_parity.py's); FUZZ_FUSED=1 keeps the big ones on
the fused pipeline.
usage: python tools/fuzz_seq.py [cases] [seed]"""
import sys, time, os; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch, raytracer_amd as rt
from oracle import oracle_py as O
from test_gpu_parity import render_dev

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
        n_pol, n_l = int(rng.integers(0, 5)), int(rng.choice([1, 1, 2, 3, 8]))
        if n_sph + n_pol == 0:
            n_pol = 1
        seed = int(rng.integers(1, 2**31))
        sc, desc = rt.Scene.synthetic(seed, n_sph, n_pol, n_l, kind), f"syn seed={seed} sph={n_sph} poly={n_pol} lights={n_l} kind={kind}"
    big = cls == 3
    W, H = int(rng.integers(8, 40 if big else 110)), int(rng.integers(8, 30 if big else 80))
    s, d = int(rng.integers(1, 4)), int(rng.integers(0, 11))
    home = (sc.c.eyePos.x, sc.c.eyePos.y, sc.c.eyePos.z)
    ctx = rt.Context()
    ctx.upload(sc)
    if os.environ.get("FUZZ_FUSED"):
        ctx.set_pipeline(rt.PIPELINE_FUSED)
    eyes = [home]
    log = []
    ok = True
    for f in range(int(rng.integers(4, 8))):
        u = rng.random()
        if u < 0.35:      # somewhere new: up over the scene, into it, behind it
            eye = (home[0] + float(rng.normal(0, 3)), home[1] + float(rng.normal(0, 4)), home[2] + float(rng.normal(0, 8)))
            eyes.append(eye)
        elif u < 0.6:     # back where it has been
            eye = eyes[int(rng.integers(0, len(eyes)))]
        else:             # standing still (the frame the graph replays)
            eye = eyes[-1] if f else home
        sc.c.eyePos.x, sc.c.eyePos.y, sc.c.eyePos.z = eye
        cam = rt.prepare_camera(sc, W, H, s, d)
        r0, r1 = 0, H
        if rng.random() < 0.25:
            r0 = int(rng.integers(0, H)); r1 = int(rng.integers(r0 + 1, H + 1))
        try:
            argb, rad, st = render_dev(torch, ctx, cam, r0, r1 - r0)
        except rt.RtHipError as e:
            ok = False
            log.append("frame %d: %s" % (f, str(e)[:100]))
            break
        a_o, r_o, _ = O.render(sc.c, cam, nthreads=16)
        differ = (rad[r0:r1].view(np.uint32) != r_o[r0:r1].view(np.uint32)).any(-1)
        same = not differ.any() and np.array_equal(argb[r0:r1], a_o[r0:r1])
        log.append("%s%d:%d-%d" % ("" if same else "BAD", f, r0, r1))
        if not same:
            ok = False
            ys, xs = np.nonzero(differ)
            log.append("differ=%d at %s eye=%r" % (int(differ.sum()), list(zip((ys + r0).tolist(), xs.tolist()))[:6], eye))
    g = ctx.stats() if hasattr(ctx, "stats") else None
    ctx.close()
    sc.c.eyePos.x, sc.c.eyePos.y, sc.c.eyePos.z = home
    if not ok:
        bad += 1
    print("%s case %d: %s %dx%d s%d d%d frames: %s [%.0f s]" % ("ok " if ok else "BAD", c, desc, W, H, s, d, " ".join(log), time.time() - t0), flush=True)
print("%d bad of %d" % (bad, cases))
sys.exit(1 if bad else 0)
