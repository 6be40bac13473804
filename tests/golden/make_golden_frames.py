#!/usr/bin/env python3
"""Adds whole-frame pins of the big configurations to the fixtures, by RUNNING THE UNTOUCHED REFERENCE
(oracle/_ref/libref_raytracer.so; authoring container only), into tests/golden/frames.npz + frames.json:

  * BASELINE config 3 (the reference's scene, 3840x2160, 4x4 samples, depth 8): P6 md5 + the ld --wrap
    ray / test counters of the full frame;
  * the scenes of configs 4 and 5 at a reduced size (320x180, 1 sample) as FRAMES - 8-bit pixels, float
    radiance and counters - so that the big-scene kernel classes are compared on pictures, not on a few pixels.

    python tests/golden/make_golden_frames.py
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from make_golden import SCENES, make_scene, ppm_md5  # noqa: E402
from oracle import oracle_py as O  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    arrays, meta = {}, {"generator": "tests/golden/make_golden_frames.py", "scenes": SCENES, "hashes": [], "frames": []}
    for (name, W, H, s, d) in [("builtin", 3840, 2160, 4, 8)]:
        t0 = time.time()
        sc = make_scene(name)  # (kept alive: the Scene object owns the arrays its scene_t points to)
        argb, cnt = O.ref_raytrace(sc.c, W, H, s, d)
        meta["hashes"].append({"scene": name, "W": W, "H": H, "sqrtSpp": s, "depth": d, "md5": ppm_md5(argb, W, H), "counts": cnt})
        print("hash", name, W, H, s, d, meta["hashes"][-1]["md5"], cnt, f"{time.time() - t0:.1f} s", flush=True)
    for (name, W, H, s, d) in [("c4", 320, 180, 1, 8), ("c5", 320, 180, 1, 16)]:
        t0 = time.time()
        sc = make_scene(name)
        argb, cnt = O.ref_raytrace(sc.c, W, H, s, d)
        rad = O.ref_radiance(sc.c, W, H, s, d)
        key = f"{name}_{W}x{H}_s{s}_d{d}"
        arrays[key + "_argb"] = argb
        arrays[key + "_rad"] = rad
        meta["frames"].append({"scene": name, "W": W, "H": H, "sqrtSpp": s, "depth": d, "key": key, "counts": cnt,
                               "md5": ppm_md5(argb, W, H)})
        print("frame", key, cnt, f"{time.time() - t0:.1f} s", flush=True)
    np.savez_compressed(os.path.join(OUT, "frames.npz"), **arrays)
    with open(os.path.join(OUT, "frames.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print("wrote frames.npz", os.path.getsize(os.path.join(OUT, "frames.npz")), "bytes")


if __name__ == "__main__":
    main()
