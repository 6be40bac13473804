"""ctypes access to the parity oracle (oracle/librt_oracle.so) and, where it was built in the
authoring container, to the untouched reference (oracle/_ref/libref_raytracer.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg.  The product package raytracer_amd never imports this module.
"""
import ctypes as C
import os

import numpy as np

from raytracer_amd import _abi as A

P = C.POINTER

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(_HERE, "librt_oracle.so")
REF_SO = os.path.join(_HERE, "_ref", "libref_raytracer.so")

_COUNT_FIELDS = ["primary_rays", "shadow_rays", "reflection_rays", "transmission_rays", "shade_nodes",
                 "nodes_sphere", "nodes_polygon", "lights_lit", "sphere_tests", "sphere_hits",
                 "polygon_tests", "polygon_rej_parallel", "polygon_rej_behind", "polygon_full",
                 "polygon_hits"]


class Counts(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in _COUNT_FIELDS] + [("nodes_at_depth", C.c_uint64 * 64)]

    def as_dict(self):
        d = {n: int(getattr(self, n)) for n in _COUNT_FIELDS}
        d["nodes_at_depth"] = [int(v) for v in self.nodes_at_depth]
        return d


_oracle = None


def oracle():
    global _oracle
    if _oracle is None:
        lib = C.CDLL(ORACLE_SO)
        P = C.POINTER
        lib.rt_oracle_prepare_camera.argtypes = [P(A.rt_camera), P(A.rt_scene_t), C.c_uint, C.c_uint,
                                                 C.c_uint, C.c_float, C.c_float, C.c_float, C.c_uint]
        lib.rt_oracle_prepare_camera.restype = None
        lib.rt_oracle_render.argtypes = [P(A.rt_scene), P(A.rt_camera), C.c_void_p, C.c_void_p,
                                         C.c_uint32, C.c_uint32, P(Counts), C.c_int]
        lib.rt_oracle_render.restype = None
        lib.rt_oracle_pixel.argtypes = [P(A.rt_scene), P(A.rt_camera), C.c_uint32, C.c_uint32,
                                        P(C.c_float * 3)]
        lib.rt_oracle_pixel.restype = None
        f3 = C.c_float * 3
        lib.rt_oracle_intersect_sphere.argtypes = [P(f3), P(f3), P(A.rt_sphere), P(f3), P(C.c_float)]
        lib.rt_oracle_intersect_polygon.argtypes = [P(f3), P(f3), P(A.rt_polygon), P(f3), P(C.c_float)]
        lib.rt_oracle_object_color.argtypes = [P(A.rt_object3d), P(f3), P(f3)]
        lib.rt_oracle_object_color.restype = None
        lib.rt_oracle_counts_size.restype = C.c_size_t
        assert lib.rt_oracle_counts_size() == C.sizeof(Counts)
        _oracle = lib
    return _oracle


def set_rounded_pow(on):
    """diagnostic: correctly rounded specular power instead of libm's powf (rt_oracle.c)"""
    oracle().rt_oracle_set_rounded_pow(int(bool(on)))


def set_shading(model):
    """0: get_shade_color_phong (what the reference calls, raytrace.c:706); 1: its unused older routine
    get_shade_color_phongblinn (raytrace.c:574-685)"""
    oracle().rt_oracle_set_shading(int(model))


def tone(rgb, op, ld_max, l_max, key_pixel=-1, wide_sums=False):
    """rt_oracle_tone (apply_tone, raytrace.c:792-871) on a copy of rgb[..., 3]; returns (rgb, sums).
    wide_sums: diagnostic variant with the two frame-wide sums in double (see rt_oracle.c)."""
    lib = oracle()
    lib.rt_oracle_set_tone_wide_sums(int(bool(wide_sums)))
    lib.rt_oracle_tone.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_float, C.c_float, C.c_int64, C.c_void_p]
    lib.rt_oracle_tone.restype = None
    out = np.ascontiguousarray(rgb, dtype=np.float32).copy()
    sums = np.zeros(2, dtype=np.float32)
    lib.rt_oracle_tone(out.ctypes.data, out.size // 3, op, ld_max, l_max, key_pixel, sums.ctypes.data)
    return out, sums


def ref_tone(rgb, scene_t, op, ld_max, l_max, key_pixel=-1):
    """The reference's own tone step on a copy of rgb[..., 3]: apply_tone for Ward (op 1); for Reinhard
    (op 2) apply_tone's loops around the reference's reinhard_tone, which apply_tone itself never calls."""
    lib = ref()
    out = np.ascontiguousarray(rgb, dtype=np.float32).copy()
    n = out.size // 3
    keep = (scene_t.ldMax, scene_t.lMax)
    scene_t.ldMax, scene_t.lMax = ld_max, l_max
    try:
        if op == 1:
            lib.apply_tone.argtypes = [C.c_void_p, C.c_uint, C.c_int, P(A.rt_scene_t)]
            lib.apply_tone.restype = None
            lib.apply_tone(out.ctypes.data, n, int(key_pixel), C.byref(scene_t))
            return out
        libm = C.CDLL("libm.so.6")
        libm.logf.argtypes = [C.c_float]; libm.logf.restype = C.c_float
        lib.get_luminance.argtypes = [C.c_void_p]; lib.get_luminance.restype = C.c_float
        lib.color_scale.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_uint]; lib.color_scale.restype = C.c_void_p
        lib.reinhard_tone.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_float, C.c_float, C.c_float, C.c_int]
        lib.reinhard_tone.restype = None
        flat = out.reshape(-1, 3)
        lbuf = np.zeros(n, dtype=np.float32)
        total, total_log = np.float32(0), np.float32(0)
        for i in range(n):  # raytrace.c:851-859 with the reference's own helpers
            p = flat[i].ctypes.data
            lib.color_scale(p, p, l_max, 0)
            lbuf[i] = lib.get_luminance(p)
            total = np.float32(total + lbuf[i])
            total_log = np.float32(total_log + np.float32(libm.logf(float(np.float32(np.float32(.00001) + lbuf[i])))))
        lib.reinhard_tone(out.ctypes.data, lbuf.ctypes.data, n, ld_max, float(total), float(total_log), int(key_pixel))
        inv = float(np.float32(1.0) / np.float32(ld_max))
        for i in range(n):  # :865-868
            p = flat[i].ctypes.data
            lib.color_scale(p, p, inv, 1)
        return out
    finally:
        scene_t.ldMax, scene_t.lMax = keep


def prepare_camera(scene_t, W, H, sqrt_spp, depth, fov_y=45.0, aspect=None, near_z=1.0):
    cam = A.rt_camera()
    if aspect is None:
        aspect = np.float32(W) / np.float32(H)
    oracle().rt_oracle_prepare_camera(C.byref(cam), C.byref(scene_t), W, H, sqrt_spp, float(fov_y),
                                      float(aspect), float(near_z), depth)
    return cam


def render(scene_t, cam, row0=0, nrows=None, nthreads=1):
    """Returns (argb[H,W] uint32, radiance[H,W,3] float32, counts dict); rows outside the slab are 0."""
    W, H = cam.frameBufferWidth, cam.frameBufferHeight
    if nrows is None:
        nrows = H - row0
    argb = np.zeros((H, W), dtype=np.uint32)
    rad = np.zeros((H, W, 3), dtype=np.float32)
    cnt = Counts()
    view = A.scene_view(scene_t)
    oracle().rt_oracle_render(C.byref(view), C.byref(cam), argb.ctypes.data, rad.ctypes.data, row0, nrows,
                              C.byref(cnt), nthreads)
    return argb, rad, cnt.as_dict()


def pixel(scene_t, cam, x, y):
    out = (C.c_float * 3)()
    view = A.scene_view(scene_t)
    oracle().rt_oracle_pixel(C.byref(view), C.byref(cam), x, y, C.byref(out))
    return np.array(out[:], dtype=np.float32)


def _f3(v):
    return (C.c_float * 3)(*[float(t) for t in v])


def intersect_sphere(origin, direction, sphere):
    pt, d = (C.c_float * 3)(), C.c_float()
    hit = oracle().rt_oracle_intersect_sphere(C.byref(_f3(origin)), C.byref(_f3(direction)), C.byref(sphere),
                                              C.byref(pt), C.byref(d))
    return hit, np.array(pt[:], dtype=np.float32), np.float32(d.value)


def intersect_polygon(origin, direction, poly):
    pt, d = (C.c_float * 3)(), C.c_float()
    hit = oracle().rt_oracle_intersect_polygon(C.byref(_f3(origin)), C.byref(_f3(direction)), C.byref(poly),
                                               C.byref(pt), C.byref(d))
    return hit, np.array(pt[:], dtype=np.float32), np.float32(d.value)


def object_color(obj, pt):
    out = (C.c_float * 3)()
    oracle().rt_oracle_object_color(C.byref(obj), C.byref(_f3(pt)), C.byref(out))
    return np.array(out[:], dtype=np.float32)


# ---------------------------------------------------------------------------------------------
# the untouched reference (only in the authoring container: /root/reference never travels)
# ---------------------------------------------------------------------------------------------

class RefRay(C.Structure):  # ray_t, ray.h:19-27 (36 B on x86-64)
    _fields_ = [("origin", A.rt_vec4), ("direction", A.rt_vec4), ("magnitude", C.c_float)]


_ref = None


def have_ref():
    return os.path.exists(REF_SO)


def ref():
    global _ref
    if _ref is None:
        lib = C.CDLL(REF_SO)
        P = C.POINTER
        lib.parse_scene.argtypes = [C.c_char_p, P(A.rt_scene_t)]
        lib.parse_scene.restype = C.c_int
        lib.free_scene.argtypes = [P(A.rt_scene_t)]
        lib.prepare_scene.argtypes = [P(A.rt_scene_t), C.c_uint, C.c_uint, C.c_uint, C.c_float, C.c_float,
                                      C.c_float]
        lib.prepare_scene.restype = None
        lib.get_pixel_color.argtypes = [P(A.rt_color), C.c_uint, C.c_uint, P(A.rt_scene_t)]
        lib.get_pixel_color.restype = C.c_void_p
        lib.raytrace.argtypes = [C.c_void_p, P(A.rt_scene_t), C.c_float, C.c_float, C.c_float, C.c_float,
                                 C.c_uint, C.c_uint, C.c_uint, C.c_uint]
        lib.raytrace.restype = None
        lib.poly_plane.argtypes = [P(A.rt_plane), P(A.rt_polygon)]
        lib.poly_plane.restype = C.c_void_p
        lib.ref_counts_get.argtypes = [P(C.c_uint64 * 6)]
        lib.ref_intersect_sphere.argtypes = [P(RefRay), P(A.rt_sphere), P(A.rt_vec4), P(C.c_float)]
        lib.ref_intersect_polygon.argtypes = [P(RefRay), P(A.rt_polygon), P(A.rt_vec4), P(C.c_float)]
        lib.get_object_color.argtypes = [P(A.rt_color), P(A.rt_object3d), P(A.rt_vec4)]
        lib.get_object_color.restype = C.c_void_p
        _ref = lib
    return _ref


def ref_set_depth(depth):
    C.c_int.in_dll(ref(), "MAX_DEPTH").value = depth


def ref_raytrace(scene_t, W, H, sqrt_spp, depth, fov_y=45.0, near_z=1.0):
    """The reference's raytrace() (main.c:159-160 argument pattern) -> (argb, counts[6])."""
    lib = ref()
    buf = np.zeros((H, W), dtype=np.uint32)
    lib.ref_counts_reset()
    lib.raytrace(buf.ctypes.data, C.byref(scene_t), fov_y, float(np.float32(W) / np.float32(H)), near_z,
                 200.0, W, H, sqrt_spp, depth)
    c = (C.c_uint64 * 6)()
    lib.ref_counts_get(C.byref(c))
    names = ["ray_create", "ray_tinypush", "sphere_tests", "sphere_hits", "polygon_tests", "polygon_hits"]
    return buf, dict(zip(names, [int(v) for v in c]))


def ref_radiance(scene_t, W, H, sqrt_spp, depth, pixels=None, fov_y=45.0, near_z=1.0):
    """Float radiance through the reference's prepare_scene + get_pixel_color.
    pixels: iterable of (x, y) or None for the full frame [H,W,3]."""
    lib = ref()
    ref_set_depth(depth)
    lib.prepare_scene(C.byref(scene_t), W, H, sqrt_spp, fov_y, float(np.float32(W) / np.float32(H)), near_z)
    col = A.rt_color()
    if pixels is None:
        out = np.zeros((H, W, 3), dtype=np.float32)
        for y in range(H):
            for x in range(W):
                col.r = col.g = col.b = 0.0
                lib.get_pixel_color(C.byref(col), x, y, C.byref(scene_t))
                out[y, x] = (col.r, col.g, col.b)
        return out
    out = np.zeros((len(pixels), 3), dtype=np.float32)
    for k, (x, y) in enumerate(pixels):
        col.r = col.g = col.b = 0.0
        lib.get_pixel_color(C.byref(col), int(x), int(y), C.byref(scene_t))
        out[k] = (col.r, col.g, col.b)
    return out
