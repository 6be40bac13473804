#!/usr/bin/env python3
"""More float-radiance pins of the two sphere-cloud configurations AT THEIR FULL FRAME SIZE, by RUNNING THE UNTOUCHED
REFERENCE (oracle/_ref/libref_raytracer.so: prepare_scene + get_pixel_color on single pixels; authoring container
only), into tests/golden/pins.npz + pins.json:

  * BASELINE config 4 (1024 spheres + 4 quads, 3840x2160, 4x4 samples, depth 8): 64 rows x 512 pixels;
  * BASELINE config 5 (16 384 spheres, 8 lights, 7680x4320, 8x8 samples, depth 16): 32 rows x 256 pixels.

The pins of a configuration lie on a few rows so that the GPU test renders a few bands of the real frame, not
hundreds; rows and columns are drawn at random (seeded), rows spread over the frame's height.  golden.npz keeps
its 48 + 12 scattered pins.

    python tests/golden/make_golden_pins.py [processes]
"""
import json
import os
import sys
import time
from multiprocessing import Pool

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from make_golden import SCENES, make_scene  # noqa: E402
from oracle import oracle_py as O  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
CONFIGS = [("c4", 3840, 2160, 4, 8, 64, 512), ("c5", 7680, 4320, 8, 16, 32, 256)]


def pins_of(W, H, rows, per_row, seed):
    rng = np.random.RandomState(seed)
    ys = (np.arange(rows) * H // rows + rng.randint(0, H // rows, rows)).astype(np.int32)
    return np.array([(int(rng.randint(0, W)), int(y)) for y in ys for _ in range(per_row)], dtype=np.int32)


def one_row(job):
    name, W, H, s, d, px = job
    sc = make_scene(name)  # (kept alive: the Scene object owns the arrays its scene_t points to)
    return O.ref_radiance(sc.c, W, H, s, d, pixels=[tuple(p) for p in px])


def main():
    procs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    arrays, meta = {}, {"generator": "tests/golden/make_golden_pins.py", "scenes": SCENES, "pins": []}
    with Pool(procs) as pool:
        for (name, W, H, s, d, rows, per_row) in CONFIGS:
            t0 = time.time()
            px = pins_of(W, H, rows, per_row, seed=W + 7 * d + 1)
            step = 64  # (pixels per job: the pool's grain)
            jobs = [(name, W, H, s, d, px[k:k + step]) for k in range(0, len(px), step)]
            rad = np.concatenate(pool.map(one_row, jobs), 0)
            key = f"{name}_{W}x{H}_s{s}_d{d}_pins"
            arrays[key + "_xy"] = px
            arrays[key + "_rad"] = rad
            meta["pins"].append({"scene": name, "W": W, "H": H, "sqrtSpp": s, "depth": d, "key": key, "rows": rows,
                                 "per_row": per_row})
            print("pins", key, len(px), "pixels, mean", rad.mean(0).tolist(), f"{time.time() - t0:.1f} s", flush=True)
    np.savez_compressed(os.path.join(OUT, "pins.npz"), **arrays)
    with open(os.path.join(OUT, "pins.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print("wrote pins.npz", os.path.getsize(os.path.join(OUT, "pins.npz")), "bytes")


if __name__ == "__main__":
    main()
