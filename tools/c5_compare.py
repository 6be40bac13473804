"""development: the config-5 scene with two builds of the library (run once per build: RT_HIP_LIB unset / set),
then the differing pixels.  usage: python tools/c5_compare.py render <tag> | diff <tagA> <tagB>"""
import sys; sys.path.insert(0, '.')
import numpy as np, raytracer_amd as rt
sc = rt.Scene.synthetic(0xC0FFEE05, 16384, 0, 8, 1)
CASES = [(960, 540, 1, 4), (1920, 1080, 2, 16)]
if sys.argv[1] == "render":
    for (W, H, s, d) in CASES:
        cam = rt.prepare_camera(sc, W, H, s, d)
        argb, rad = rt.render(sc, cam)
        np.save(f"gpurun_out/c5_{sys.argv[2]}_{d}.npy", rad)
else:
    from oracle import oracle_py as O
    O.set_rounded_pow(True)
    for (W, H, s, d) in CASES:
        cam = rt.prepare_camera(sc, W, H, s, d)
        a, b = np.load(f"gpurun_out/c5_{sys.argv[2]}_{d}.npy"), np.load(f"gpurun_out/c5_{sys.argv[3]}_{d}.npy")
        differ = (a.view(np.uint32) != b.view(np.uint32)).any(-1)
        print(f"depth {d}: pixels that differ between the builds: {differ.sum()}", np.argwhere(differ)[:6].tolist(), flush=True)
