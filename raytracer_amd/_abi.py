"""ctypes mirror of include/rt_hip.h (the C-ABI of librt_hip.so).

The structures are layout-compatible with the reference's x86-64 structs (vector4.h:14-29,
color.h:17-22, plane.h:14-29, geometry.h:22-48, materials.h:24-34, object3d.h:17-35,
light.h:17-26, scene.h:22-64); tests/test_abi.py pins sizes and offsets.
"""
import ctypes as C

RT_GEOMETRY_SPHERE = 0x01
RT_GEOMETRY_POLYGON = 0x02
RT_E_INVALID = -10001
RT_E_OVERFLOW = -10002
RT_E_NOSCENE = -10003


class rt_vec4(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float), ("w", C.c_float)]


class rt_color(C.Structure):
    _fields_ = [("r", C.c_float), ("g", C.c_float), ("b", C.c_float)]


class rt_plane(C.Structure):
    _fields_ = [("A", C.c_float), ("B", C.c_float), ("C", C.c_float), ("F", C.c_float)]


class rt_sphere(C.Structure):
    _fields_ = [("center", rt_vec4), ("radius", C.c_float)]


class rt_polygon(C.Structure):
    _fields_ = [("plane", rt_plane), ("nVerticies", C.c_uint32), ("vertex", C.POINTER(rt_vec4))]


class rt_material(C.Structure):
    _fields_ = [("colors", rt_color * 2), ("phong_ka", C.c_float), ("phong_kd", C.c_float),
                ("phong_ks", C.c_float), ("phong_ke", C.c_float), ("kr", C.c_float),
                ("kt", C.c_float), ("n", C.c_float)]


class _geom(C.Union):
    _fields_ = [("sphr_obj", rt_sphere), ("poly_obj", rt_polygon)]


class rt_object3d(C.Structure):
    _anonymous_ = ("g",)
    _fields_ = [("geometryType", C.c_uint16), ("debugName", C.c_char_p), ("material", rt_material),
                ("g", _geom)]


class rt_pointlight(C.Structure):
    _fields_ = [("position", rt_vec4), ("color", rt_color), ("range", C.c_float)]


class rt_scene_t(C.Structure):
    _fields_ = [("eyePos", rt_vec4), ("lookAt", rt_vec4), ("upVec", rt_vec4),
                ("U", rt_vec4), ("V", rt_vec4), ("N", rt_vec4),
                ("frameBufferWidth", C.c_uint32), ("frameBufferHeight", C.c_uint32),
                ("sqrtSpp", C.c_uint32), ("viewDistance", C.c_float),
                ("viewPlaneHalfWidth", C.c_float), ("viewPlaneHalfHeight", C.c_float),
                ("sppWidth", C.c_float), ("bgColor", rt_color), ("ambientLightColor", rt_color),
                ("ldMax", C.c_float), ("lMax", C.c_float), ("nLights", C.c_uint32),
                ("nObjects", C.c_uint32), ("lights", C.POINTER(rt_pointlight)),
                ("objects", C.POINTER(rt_object3d))]


class rt_scene(C.Structure):
    _fields_ = [("bgColor", rt_color), ("ambientLightColor", rt_color), ("nLights", C.c_uint32),
                ("nObjects", C.c_uint32), ("lights", C.POINTER(rt_pointlight)),
                ("objects", C.POINTER(rt_object3d))]


class rt_camera(C.Structure):
    _fields_ = [("eyePos", rt_vec4), ("U", rt_vec4), ("V", rt_vec4), ("N", rt_vec4),
                ("frameBufferWidth", C.c_uint32), ("frameBufferHeight", C.c_uint32),
                ("sqrtSpp", C.c_uint32), ("viewDistance", C.c_float),
                ("viewPlaneHalfWidth", C.c_float), ("viewPlaneHalfHeight", C.c_float),
                ("sppWidth", C.c_float), ("depth", C.c_uint32)]


class rt_framebuffer(C.Structure):
    _fields_ = [("argb", C.POINTER(C.c_uint32)), ("radiance", C.POINTER(C.c_float)),
                ("width", C.c_uint32), ("height", C.c_uint32), ("row0", C.c_uint32),
                ("nrows", C.c_uint32)]


_U64 = C.c_uint64
_STAT_U64 = [
    "primary_rays", "shadow_rays", "reflection_rays", "transmission_rays", "shade_nodes",
    "tree_nodes", "queued_rays"]
_STAT_U64_B = [
    "sphere_tests", "sphere_hits", "polygon_tests", "polygon_rej_parallel", "polygon_rej_behind",
    "polygon_full", "polygon_hits", "lights_lit", "nodes_sphere", "nodes_polygon",
    "polygon_screen_unsure", "polygon_screen_mismatch",
    "k0_sphere_tests", "k0_sphere_hits", "k0_polygon_tests", "k0_polygon_rej_parallel",
    "k0_polygon_rej_behind", "k0_polygon_full", "k0_shade_nodes", "k0_nodes_sphere",
    "k0_nodes_polygon", "k0_lights_lit", "k0_shadow_rays", "k0_reflection_rays",
    "k0_transmission_rays"]


class rt_hip_stats(C.Structure):
    _fields_ = ([(n, _U64) for n in _STAT_U64]
                + [("batches", C.c_uint32), ("overflow_retries", C.c_uint32)]
                + [(n, _U64) for n in _STAT_U64_B]
                + [("k0_ms", C.c_float), ("k0_launches", C.c_uint32), ("primary_rays_culled", _U64),
                   ("planned_launches", C.c_uint32), ("plan_builds", C.c_uint32),
                   ("work_mib", C.c_uint32), ("reserved_", C.c_uint32),
                   ("graph_launches", C.c_uint32), ("graph_captures", C.c_uint32)])

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


def scene_view(scene_t: rt_scene_t) -> rt_scene:
    """The part of scene_t the trace loop reads (what raytrace() forwards to rt_hip_render)."""
    v = rt_scene()
    v.bgColor = scene_t.bgColor
    v.ambientLightColor = scene_t.ambientLightColor
    v.nLights = scene_t.nLights
    v.nObjects = scene_t.nObjects
    v.lights = scene_t.lights
    v.objects = scene_t.objects
    return v
