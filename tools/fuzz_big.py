"""development: random BIG scenes (a sphere grid: the job pipeline, csrc/rt_jobs.inc) as scene files - clouds of 130-1500
spheres, glass of index 0.8 / 0.95 / 1 / 1.25 (rays longer than unit: k_loop behind its groups' balls), mirrors, 0-4
quads near and far, 1-8 lights (k_hits sends the shadow rays of one or two), the eye outside, inside and far from the
cloud - GPU against the oracle with libm's powf rounding taken out: every pixel bit-identical.
FUZZ_DEVICES=1: through rt_hip_render_devices, 1-5 shares of one GPU.  FUZZ_WILD=1: refraction indices of 0, -1, 1e-30, 1e30 and reflection / transmission coefficients below 0 and above 1.
usage: python tools/fuzz_big.py [cases] [seed] [first case rendered]"""
import sys, time, tempfile, os; sys.path.insert(0, '.')
import numpy as np, raytracer_amd as rt
from oracle import oracle_py as O

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
first = int(sys.argv[3]) if len(sys.argv) > 3 else 0
O.set_rounded_pow(True)
bad = 0
t0 = time.time()
tmp = tempfile.mkdtemp()
for c in range(cases):
    n_sph = int(rng.choice([130, 200, 400, 900, 1500]))
    n_l = int(rng.choice([1, 1, 2, 2, 3, 5, 8]))
    n_pol = int(rng.integers(0, 5))
    ext = float(rng.choice([2.0, 6.0, 15.0]))
    eye = [0.0, 4.5, 2.0]
    mode = int(rng.integers(0, 4))
    if mode == 1:   # inside the cloud
        eye = [float(rng.uniform(-ext / 2, ext / 2)), 4.5 + float(rng.uniform(-ext / 4, ext / 4)), -8.0 + float(rng.uniform(-ext / 2, ext / 2))]
    elif mode == 2: # far away
        eye = [float(rng.normal(0, 30)), 4.5 + float(rng.normal(0, 10)), 60.0 + float(rng.normal(0, 30))]
    parts = ["desc\n{\n\t%d\n\t%d\n}\n" % (n_sph + n_pol, n_l),
             "camera\n{\nposition %r %r %r\nlookat 0 4.5 -8\nupvector 0 1 0\nprojection 45.0 1.0 200.0\n}\n" % tuple(eye),
             "world\n{\nambient 1.0 1.0 1.0\nbackground .3 .4 .6\n}\n"]
    for l in range(n_l):
        parts.append("pointlight\n{\nposition %r %r %r\ndiffuse %g %g %g\nrange 150\n}\n"
                     % (float(rng.uniform(-10, 10)), float(rng.uniform(2, 14)), float(rng.uniform(-16, 4)), *rng.uniform(0.3, 1, 3)))
    rmax = ext / 12.0
    for k in range(n_sph):
        x, y, z = rng.uniform(-ext / 2, ext / 2), 4.5 + rng.uniform(-ext / 2, ext / 2), -8.0 + rng.uniform(-ext / 2, ext / 2)
        m = int(rng.integers(0, 4))
        kr, kt = [(0.0, 0.0), (0.5, 0.0), (0.05, 0.8), (0.3, 0.5)][m]
        n = float(rng.choice([0.8, 0.95, 1.0, 1.25]))
        if os.environ.get("FUZZ_WILD"):  # indices and coefficients no material has: zero and negative indices, gains above one
            n = float(rng.choice([0.0, -1.0, 1e-30, 1e30, 1e-12, 1e12, 1e-20, 3e-19, 0.5, 3.0, 1.0]))
            kr, kt = float(rng.choice([0.0, -0.5, 1.5, 0.3])), float(rng.choice([0.0, -0.5, 2.0, 0.6]))
        parts.append("sphere\n{\nname S%d\nposition %r %r %r\nradius %r\ndiffuse %g %g %g\nspecular 1 1 1\n"
                     "ka 0.1 kd 0.4 ks 0.3 ke %d kr %g kt %g n %g\n}\n"
                     % (k, x, y, z, float(rng.uniform(0.02, rmax)), *rng.uniform(0.2, 1, 3), int(rng.choice([2, 20])), kr, kt, n))
    for q in range(n_pol):
        far = rng.random() < 0.5
        h = (60.0 if far else ext)
        y0 = 4.5 - ext / 2 - (0.2 if q == 0 else float(rng.uniform(0.2, 3.0)))
        parts.append("polygon\n{\nname P%d\npoint %g %g %g\npoint %g %g %g\npoint %g %g %g\npoint %g %g %g\ndiffuse 0 1 0\n"
                     "ka 0.1 kd 0.6 ks 0.2 ke 2 kr %g kt 0 n 1\n}\n"
                     % (q, h, y0, -8 + h, h, y0, -8 - h, -h, y0, -8 - h, -h, y0, -8 + h, float(rng.choice([0.0, 0.4]))))
    path = os.path.join(tmp, "s%d.scene" % c)
    open(path, "w").write("".join(parts))
    sc = rt.load_scene(path)
    W, H = int(rng.integers(8, 48)), int(rng.integers(8, 36))
    s, d = int(rng.integers(1, 4)), int(rng.integers(0, 9))
    if c < first:
        continue
    if os.environ.get("FUZZ_DUMP"):
        os.environ["FUZZ_LAST_SCENE"] = path
        os.environ["FUZZ_LAST_FRAME"] = "%d %d %d %d" % (W, H, s, d)
    cam = rt.prepare_camera(sc, W, H, s, d)
    try:
        if os.environ.get("FUZZ_CTX"):  # the context interface: small queues (batches, halved batches), row slabs, the tolerance build never
            import torch
            sys.path.insert(0, "tests")
            from test_gpu_parity import render_dev
            fmode = os.environ.get("FUZZ_CTX")
            qcap = int(rng.choice([1 << 14, 1 << 16, 1 << 18, 1 << 20, 0])); bs = int(rng.choice([0, 0, 1 << 10, 1 << 14, 1 << 30]))
            if fmode == "slabs":
                qcap, bs = 1 << 22, 0
            ctx = rt.Context(queue_cap=qcap)
            ctx.upload(sc)
            if os.environ.get("FUZZ_FUSED"):
                ctx.set_pipeline(rt.PIPELINE_FUSED)
            if bs:
                ctx.set_batch_samples(bs)
            argb = np.zeros((H, W), np.uint32); rad = np.zeros((H, W, 3), np.float32)
            cuts = sorted(set([0, H] + ([] if fmode == "caps" else [int(x) for x in rng.integers(0, H + 1, int(rng.integers(0, 3)))])))
            extra = " cap=%d batch=%d cuts=%s" % (qcap, bs, cuts)
            for r0, r1 in zip(cuts[:-1], cuts[1:]):
                a_, r_, st_ = render_dev(torch, ctx, cam, r0, r1 - r0)
                argb[r0:r1] = a_[r0:r1]; rad[r0:r1] = r_[r0:r1]
            ctx.close()
        elif os.environ.get("FUZZ_DEVICES"):  # rt_hip_render_devices: the frame's 8-row blocks dealt to 1-5 shares of this box's one GPU
            argb, rad = rt.render(sc, cam, devices=[0] * int(rng.integers(1, 6)))
        else:
            argb, rad = rt.render(sc, cam)
    except rt.RtHipError as e:
        if os.environ.get("FUZZ_CTX") and "overflow on a single" in str(e):  # a context of an explicit, tiny capacity: what the interface promises
            print("cap  case %d: %s" % (c, str(e)[-90:]), flush=True)
            continue
        bad += 1
        print("BAD case %d: %dx%d s%d d%d sph=%d lights=%d quads=%d ext=%g eye-mode=%d: %s" % (c, W, H, s, d, n_sph, n_l, n_pol, ext, mode, str(e)[:120]), flush=True)
        continue
    a_o, r_o, _ = O.render(sc.c, cam, nthreads=16)
    differ = (rad.view(np.uint32) != r_o.view(np.uint32)).any(-1)
    ok = not differ.any() and np.array_equal(argb, a_o)
    argb2, rad2 = (argb, rad) if os.environ.get("FUZZ_CTX") else rt.render(sc, cam)  # (the second render: one device, whatever the first was)
    ok2 = np.array_equal(argb2, argb) and np.array_equal(rad2.view(np.uint32), rad.view(np.uint32))
    if not (ok and ok2):
        bad += 1
    print("%s case %d: %dx%d s%d d%d sph=%d lights=%d quads=%d ext=%g eye-mode=%d differ=%d second-render-same=%s [%.0f s]%s"
          % ("ok " if ok and ok2 else "BAD", c, W, H, s, d, n_sph, n_l, n_pol, ext, mode, int(differ.sum()), ok2, time.time() - t0, extra if os.environ.get("FUZZ_CTX") else ""), flush=True)
    if not ok:
        ys, xs = np.nonzero(differ)
        print("    differing pixels (y, x):", list(zip(ys.tolist(), xs.tolist()))[:12], flush=True)
        for y, x in list(zip(ys.tolist(), xs.tolist()))[:4]:
            print("    (%d, %d): GPU %r oracle %r" % (y, x, rad[y, x].tolist(), r_o[y, x].tolist()), flush=True)
        if os.environ.get("FUZZ_DUMP"):
            import shutil
            shutil.copy(path, os.path.join(os.environ["FUZZ_DUMP"], "bad_case_%d.scene" % c))
print("%d bad of %d" % (bad, cases))
sys.exit(1 if bad else 0)
