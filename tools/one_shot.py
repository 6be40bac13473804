#!/usr/bin/env python3
"""The call the reference makes (main.c:159-160): raytrace() with pixels in the caller's pageable buffer.
  warm : the same process calling it frame after frame (ms per call, median of 30)
  cold : a fresh child process - the command-line host raytracer_amd/rt, which is the reference's main.c contract -
         from exec to exit (image written to /dev/shm), and the library's own account of its first call
usage: python tools/one_shot.py [W H sqrtSpp depth]        prints one JSON object"""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def measure(W=1920, H=1080, s=4, d=8, numerics=("exact", "fast"), calls=30):
    out = {"workload": f"builtin scene {W}x{H}, sqrtSpp {s}, depth {d}; raytrace() into a pageable host buffer"}
    for num in numerics:
        env = dict(os.environ, RT_HIP_NUMERICS=num, RT_HIP_TIMES="1")
        img = f"/dev/shm/rt_one_shot_{os.getpid()}.ppm"
        t0 = time.perf_counter()
        p = subprocess.run([os.path.join(ROOT, "raytracer_amd", "rt"), img, os.path.join(ROOT, "tests", "golden", "whitted.scene"), str(W), str(H), str(s), str(d)],
                           capture_output=True, text=True, env=env, cwd=ROOT, timeout=120)
        cold = time.perf_counter() - t0
        if os.path.exists(img):
            os.unlink(img)
        if p.returncode != 0:
            raise RuntimeError(p.stdout + p.stderr)
        first = [l for l in p.stderr.splitlines() if l.startswith("rt_hip_render:")]
        code = ("import os, sys, time, json, numpy as np; sys.path.insert(0, %r); import raytracer_amd as rt\n"
                "sc = rt.Scene.builtin(); buf = np.zeros(%d, dtype=np.uint32); ts = []\n"
                "for i in range(%d):\n"
                "    t = time.perf_counter(); rt.raytrace(buf, sc, 45.0, %d / %d, 1.0, 200.0, %d, %d, %d, %d); ts.append(time.perf_counter() - t)\n"
                "print(json.dumps(sorted(ts[5:])))\n" % (ROOT, W * H, calls + 5, W, H, W, H, s, d))
        env2 = dict(os.environ, RT_HIP_NUMERICS=num)
        q = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env2, cwd=ROOT, timeout=300)
        if q.returncode != 0:
            raise RuntimeError(q.stdout + q.stderr)
        ts = json.loads([l for l in q.stdout.splitlines() if l.startswith("[")][-1])
        out[num] = {"cold_process_ms": cold * 1e3, "cold_first_call": first[0] if first else None,
                    "warm_call_ms_p50": ts[len(ts) // 2] * 1e3, "warm_call_ms_min": ts[0] * 1e3}
    return out


if __name__ == "__main__":
    a = [int(x) for x in sys.argv[1:5]]
    print(json.dumps(measure(*a) if len(a) == 4 else measure()))
