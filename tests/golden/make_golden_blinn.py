"""Fixtures of the reference's second shading routine, get_shade_color_phongblinn (raytrace.c:574-685),
made by RUNNING the untouched reference build with oracle/_ref/libref_blinn_shim.so ahead of it in the
symbol scope (oracle/ref_blinn_shim.c: get_ray_color's call of get_shade_color_phong lands in the other
routine; no reference source is changed or copied).  Run in the build container (needs /root/reference),
in a process of its own - the shim must be loaded before the reference library:
    python tests/golden/make_golden_blinn.py
writes tests/golden/blinn.npz + blinn.json.  Only data is stored: scene names, sizes, pixel arrays, counts."""
import ctypes as C
import json, os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(HERE, "..", "..")
sys.path.insert(0, ROOT)
SHIM = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_blinn_shim.so"), mode=C.RTLD_GLOBAL)
import raytracer_amd as rt  # noqa: E402
from oracle import oracle_py as O  # noqa: E402
from make_golden import SCENES, make_scene  # noqa: E402


def scene_of(name):
    if name.endswith(".scene"):
        return rt.load_scene(os.path.join(HERE, name))
    return make_scene(name)


def main():
    SHIM.ref_blinn_shim_calls.restype = C.c_ulong
    cases, arrays = [], {}
    for (name, W, H, s) in [("builtin", 96, 72, 2), ("builtin", 67, 45, 3), ("syn64", 96, 64, 2),
                            ("syn300", 64, 48, 1), ("triangle_two_lights.scene", 80, 60, 2),
                            ("whitted.scene", 64, 48, 1)]:
        sc = scene_of(name)
        before = SHIM.ref_blinn_shim_calls()
        argb, cnt = O.ref_raytrace(sc.c, W, H, s, 8)
        nodes = SHIM.ref_blinn_shim_calls() - before
        assert nodes > 0, "the shim is not in front of the reference library"
        assert cnt["ray_tinypush"] == 0  # the routine pushes nothing and spawns nothing
        rad = O.ref_radiance(sc.c, W, H, s, 8)
        key = f"{name.replace('.scene', '')}_{W}x{H}_s{s}"
        arrays[key + "_argb"] = argb
        arrays[key + "_rad"] = rad
        cases.append({"scene": name, "W": W, "H": H, "sqrtSpp": s, "key": key, "counts": cnt, "shade_nodes": int(nodes)})
        print(key, cnt, nodes, flush=True)
    np.savez_compressed(os.path.join(HERE, "blinn.npz"), **arrays)
    json.dump({"glibc": os.confstr("CS_GNU_LIBC_VERSION"), "scenes": {k: v for k, v in SCENES.items()},
               "cases": cases}, open(os.path.join(HERE, "blinn.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
