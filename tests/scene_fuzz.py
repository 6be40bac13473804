"""Random scene FILES for rt_load_scene (used by tests/test_gpu_parity.py and tools/fuzz_polygons.py):
arbitrary polygons - convex n-gons in random planes, big horizontal "floors", non-convex stars, slightly
non-planar ones, either vertex order (which decides whether the reference's F >= 0 mirrors the plane,
geometry.c:29-32) - plus a few spheres and one or two lights."""
import numpy as np


def random_polygon(rng, kind):
    n = int(rng.integers(3, 9))
    c = np.array([rng.normal(0, 3), rng.uniform(-1, 5), rng.uniform(-20, -4)])
    a = rng.normal(size=3); a /= np.linalg.norm(a)
    b = np.cross(a, rng.normal(size=3)); b /= np.linalg.norm(b)
    if kind == "floor":
        a, b, c = np.array([1.0, 0, 0]), np.array([0, 0, -1.0]), np.array([rng.normal(0, 2), rng.uniform(-1, 0.5), -15.0])
    r = rng.uniform(1.0, 6.0) * (4 if kind == "floor" else 1)
    ang = np.sort(rng.uniform(0, 2 * np.pi, n))
    if np.diff(np.append(ang, ang[0] + 2 * np.pi)).max() > np.pi * 0.95:
        ang = np.linspace(0, 2 * np.pi, n, endpoint=False) + rng.uniform(0, 1)
    rad = np.full(n, r)
    if kind == "star":
        rad[::2] *= 0.45
    if rng.random() < 0.5:
        ang = ang[::-1]
    pts = [c + rad[i] * (np.cos(ang[i]) * a + np.sin(ang[i]) * b) for i in range(n)]
    if kind == "warped":
        pts[int(rng.integers(0, n))] += np.cross(a, b) * rng.uniform(0.01, 0.3)
    return pts


def random_scene_text(rng, cloud=0):
    """-> (text of a scene file, list of polygon kinds); cloud > 0: that many small spheres on top, mirrors and glass
    among them (the scene is then searched through a grid: the polygons meet the job pipeline)"""
    lines = ["camera", "{", f"position {rng.normal(0, 1):.4f} {rng.uniform(1, 5):.4f} {rng.uniform(0, 4):.4f}",
             f"lookat {rng.normal(0, 1):.4f} {rng.uniform(0, 3):.4f} -8", "upvector 0 1 0", "projection 45.0 1.0 200.0", "}",
             "world", "{", "ambient 0.7 0.7 0.8", "background 0.1 0.1 0.3", "}"]
    for _ in range(int(rng.integers(1, 3))):
        lines.append(f"pointlight {{ position {rng.normal(0, 5):.3f} {rng.uniform(4, 12):.3f} {rng.uniform(-10, 4):.3f} diffuse 0.9 0.8 0.7 }}")
    kinds = []
    for _ in range(int(rng.integers(1, 5))):
        kind = str(rng.choice(["convex", "convex", "floor", "star", "warped"]))
        kinds.append(kind)
        pts = random_polygon(rng, kind)
        mat = f"ka 0.1 kd 0.6 ks 0.3 ke {int(rng.choice([2, 8, 20]))} kr {rng.choice([0, 0.3, 0.6])} kt {rng.choice([0, 0, 0.5])} n 1.2"
        lines.append("polygon { name p " + " ".join(f"point {p[0]:.5f} {p[1]:.5f} {p[2]:.5f}" for p in pts) + " diffuse 0.8 0.7 0.3 " + mat + " }")
    for _ in range(int(rng.integers(0, 4))):
        lines.append(f"sphere {{ name s position {rng.normal(0, 3):.3f} {rng.uniform(0, 4):.3f} {rng.uniform(-14, -4):.3f} radius {rng.uniform(0.4, 1.6):.3f} "
                     f"diffuse 0.9 0.9 0.9 ka 0.1 kd 0.4 ks 0.5 ke 20 kr {rng.choice([0, 0.5])} kt {rng.choice([0, 0.7])} n 1.3 }}")
    for _ in range(cloud):
        lines.append(f"sphere {{ name c position {rng.normal(0, 4):.3f} {rng.uniform(0, 5):.3f} {rng.uniform(-16, -2):.3f} radius {rng.uniform(0.05, 0.5):.3f} "
                     f"diffuse 0.6 0.9 0.7 ka 0.1 kd 0.4 ks 0.5 ke 20 kr {rng.choice([0, 0, 0.5])} kt {rng.choice([0, 0, 0.7])} n {rng.choice([1.3, 0.9])} }}")
    return "\n".join(lines) + "\n", kinds
