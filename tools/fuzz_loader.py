"""development: the scene-file loader (csrc/rt_host.c: rt_load_scene) on mutated scene files, compiled with
gcc -fsanitize=address,undefined (host code only, no GPU).  usage: python tools/fuzz_loader.py [files] [seed]"""
import os, random, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOKS = ['{', '}', 'sphere', 'polygon', 'point', 'camera', 'pointlight', 'position', 'radius', 'diffuse', '1e39', '-',
        'nan', 'inf', '0', '1.0f', '#', '\n', '"', 'desc', 'world', 'ambient', 'kt', 'n', 'ke', 'lookat', 'upvector',
        'projection', '999999999999999999999', '.', 'e', '+', 'specular', 'background', 'name']
HARNESS = r'''
#include "rt_hip.h"
#include <string.h>
int main(int argc, char **argv)
{
    for (int i = 1; i < argc; ++i) {
        rt_scene_t sc;
        memset(&sc, 0, sizeof sc);
        if (rt_load_scene(argv[i], &sc))
            rt_scene_free(&sc);
    }
    return 0;
}
'''


def mutate(rng, t):
    for _ in range(rng.randint(1, 20)):
        op, i = rng.random(), rng.randrange(len(t) + 1)
        if op < 0.3:
            t = t[:i] + rng.choice(TOKS) + t[i:]
        elif op < 0.6:
            t = t[:i] + t[min(len(t), i + rng.randint(1, 40)):]
        elif op < 0.8:
            t = t[:i] + ' '.join(rng.choice(TOKS) for _ in range(rng.randint(1, 6))) + t[i:]
        elif op < 0.9:
            t = t[:i] + t[i:i + 200] * rng.randint(2, 30) + t[i:]
        else:
            t = t[:i]
    return t


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    base = [open(os.path.join(ROOT, "tests", "golden", f)).read() for f in ("whitted.scene", "triangle_two_lights.scene")]
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "h.c"), "w").write(HARNESS)
        subprocess.check_call(["gcc", "-g", "-O1", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
                               "-I" + os.path.join(ROOT, "include"), os.path.join(d, "h.c"),
                               os.path.join(ROOT, "raytracer_amd", "csrc", "rt_host.c"), "-lm", "-o", os.path.join(d, "h")])
        files = []
        for k in range(n):
            f = os.path.join(d, f"{k}.scene")
            open(f, "w").write(mutate(rng, rng.choice(base)))
            files.append(f)
        bad = 0
        for i in range(0, n, 100):
            p = subprocess.run([os.path.join(d, "h")] + files[i:i + 100], capture_output=True, text=True)
            if p.returncode != 0 or "ERROR" in p.stderr or "runtime error" in p.stderr:
                bad += 1
                print(p.returncode, p.stderr[-1500:])
        print(f"{n} files, {bad} batches with findings")
        return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
