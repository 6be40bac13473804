"""development (authoring container only: needs /root/reference and gcc): how much the REFERENCE's own frames
move when it is compiled with other flags - the yardstick for the tolerance build of the kernels.
Builds the untouched sources twice into /tmp (-O2, what oracle/_ref holds, and -O3 -march=native
-ffp-contract=fast) and compares radiance / pixels.  Output kept in profiles/r02_ref_flag_sensitivity.txt."""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import raytracer_amd as rt
from oracle import oracle_py as O
from conftest import rgb_bytes

R, OUT = "/root/reference", "/tmp/refnative"
os.makedirs(OUT + "/obj", exist_ok=True)
objs = []
for s in "color geometry object3d plane ray raytrace scene vector4".split():
    subprocess.check_call(["gcc", "-O3", "-march=native", "-ffp-contract=fast", "-fPIC", "-w", "-I" + R, "-c", f"{R}/{s}.c",
                           "-o", f"{OUT}/obj/{s}.o"])
    objs.append(f"{OUT}/obj/{s}.o")
subprocess.check_call(["gcc", "-O2", "-fPIC", "-c", os.path.join(ROOT, "oracle", "ref_wrap.c"), "-o", OUT + "/obj/ref_wrap.o"])
subprocess.check_call(["gcc", "-shared", "-Wl,--wrap=ray_create,--wrap=ray_tinypush,--wrap=ray_intersect_sphere,"
                       "--wrap=ray_intersect_polygon", "-o", OUT + "/libref_native.so"] + objs + [OUT + "/obj/ref_wrap.o", "-lm"])
LIBS = [os.path.join(ROOT, "oracle", "_ref", "libref_raytracer.so"), OUT + "/libref_native.so"]


def both(fn):
    out = []
    for lib in LIBS:
        O._ref, O.REF_SO = None, lib
        out.append(fn())
    return out


print("reference gcc -O2 against reference gcc -O3 -march=native -ffp-contract=fast (x86-64, glibc), same sources")
b = rt.Scene.builtin()
for (W, H, s, d) in [(640, 480, 1, 4), (67, 45, 3, 5), (333, 217, 3, 5), (1920, 1080, 4, 8)]:
    a0, a1 = both(lambda: O.ref_raytrace(b.c, W, H, s, d)[0])
    diff = np.abs(rgb_bytes(a0) - rgb_bytes(a1)).max(-1)
    print(f"builtin {W}x{H} sqrtSpp {s} depth {d}: {int((diff > 1).sum())} of {W * H} pixels differ by more than 1 LSB "
          f"({(diff <= 1).mean():.6f} within)")
for args, (W, H, s, d) in [((3237998084, 64, 4, 1, 0), (96, 64, 2, 4)), ((1766711702, 438, 0, 8, 0), (64, 33, 2, 2)),
                           ((3237998085, 3000, 4, 2, 1), (48, 32, 1, 3))]:
    sc = rt.Scene.synthetic(*args)
    r0, r1 = both(lambda: np.asarray(O.ref_radiance(sc.c, W, H, s, d)).reshape(H, W, 3))
    rel = (np.abs(r0 - r1) / np.maximum(np.abs(r0), 1e-3)).max(-1)
    print(f"synthetic{args} {W}x{H} sqrtSpp {s} depth {d}: radiance within 1e-4 relative on {(rel <= 1e-4).mean():.4f} of the "
          f"pixels; percentiles 50/90/99 of the relative difference {np.percentile(rel, [50, 90, 99])}")
