"""Loader for the in-tree HIP shared library.  There is no CPU fallback: if librt_hip.so is
missing the import fails loudly (build it with `python -c 'import __graft_entry__ as g; g.build()'`
or `make -C raytracer_amd/csrc`)."""
import ctypes as C
import os

from . import _abi as A

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RT_HIP_LIB") or os.path.join(_HERE, "librt_hip.so")  # override: development A/B builds

# every symbol include/rt_hip.h declares
EXPORTS = [
    "raytrace", "rt_hip_render", "rt_camera_prepare", "rt_hip_ctx_create", "rt_hip_ctx_destroy",
    "rt_hip_scene_upload", "rt_hip_render_rows", "rt_hip_render_interleaved", "rt_hip_finish", "rt_hip_render_stats",
    "rt_hip_ctx_set_timing", "rt_hip_ctx_set_batch_samples", "rt_hip_ctx_set_shading", "rt_hip_ctx_set_tail_stream", "rt_hip_last_error", "rt_hip_version",
    "rt_poly_plane", "rt_scene_builtin", "rt_parse_scene", "rt_scene_free", "rt_scene_synthetic",
    "rt_write_image", "write_image", "rt_hip_selftest_arith", "rt_load_scene", "rt_hip_deinterleave",
    "rt_hip_render_devices", "rt_hip_tone", "rt_hip_tone_sums", "rt_hip_tone_apply", "rt_hip_selftest_grid",
    "rt_hip_tone_static", "rt_hip_ctx_set_numerics", "rt_hip_ctx_set_specialization", "rt_hip_ctx_set_jit", "rt_hip_ctx_jit_status", "rt_hip_ctx_set_launch_plan", "rt_hip_ctx_set_graph", "rt_launch_plan_order",
    "rt_hip_ctx_set_pipeline",
]

_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP extension is not built. There is no CPU fallback; "
            "run __graft_entry__.build() (hipcc --offload-arch=gfx950).")
    try:  # share torch's HIP runtime (same SONAME) when torch is in the process
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is optional for the C-ABI itself
        pass
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    if os.environ.get("RT_HIP_LIB"):  # development A/B against an older build: tolerate entry points it lacks
        class _Missing:
            argtypes = restype = None
        real = lib

        class _Tolerant:
            def __getattr__(self, name):
                try:
                    return getattr(real, name)
                except AttributeError:
                    return _Missing()
        lib = _Tolerant()
    u32, u64, f32, vp, i32 = C.c_uint32, C.c_uint64, C.c_float, C.c_void_p, C.c_int
    P = C.POINTER
    lib.raytrace.argtypes = [vp, P(A.rt_scene_t), f32, f32, f32, f32, C.c_uint, C.c_uint, C.c_uint, C.c_uint]
    lib.raytrace.restype = None
    lib.rt_hip_render.argtypes = [P(A.rt_scene), P(A.rt_camera), P(A.rt_framebuffer)]
    lib.rt_hip_render.restype = i32
    lib.rt_hip_render_devices.argtypes = [P(A.rt_scene), P(A.rt_camera), P(A.rt_framebuffer), P(C.c_int), u32]
    lib.rt_hip_render_devices.restype = i32
    lib.rt_hip_selftest_grid.argtypes = [vp, vp, u32, i32, vp]
    lib.rt_hip_selftest_grid.restype = i32
    lib.rt_hip_tone.argtypes = [vp, vp, u64, i32, f32, f32, C.c_int64, vp]
    lib.rt_hip_tone.restype = i32
    lib.rt_hip_tone_static.argtypes = [vp, vp, u64, i32, f32, f32, C.c_int64, vp]
    lib.rt_hip_tone_static.restype = i32
    lib.rt_hip_tone_sums.argtypes = [vp, u64, f32, C.c_int64, vp, vp]
    lib.rt_hip_tone_sums.restype = i32
    lib.rt_hip_tone_apply.argtypes = [vp, vp, u64, i32, f32, f32, vp, u64, i32, vp]
    lib.rt_hip_tone_apply.restype = i32
    lib.rt_camera_prepare.argtypes = [P(A.rt_camera), P(A.rt_scene_t), C.c_uint, C.c_uint, C.c_uint, f32, f32, f32, C.c_uint]
    lib.rt_camera_prepare.restype = None
    lib.rt_hip_ctx_create.argtypes = [P(vp), i32, u64]
    lib.rt_hip_ctx_create.restype = i32
    lib.rt_hip_ctx_destroy.argtypes = [vp]
    lib.rt_hip_ctx_destroy.restype = None
    lib.rt_hip_scene_upload.argtypes = [vp, P(A.rt_scene)]
    lib.rt_hip_scene_upload.restype = i32
    lib.rt_hip_render_rows.argtypes = [vp, P(A.rt_camera), vp, vp, u32, u32, vp]
    lib.rt_hip_render_rows.restype = i32
    lib.rt_hip_render_interleaved.argtypes = [vp, P(A.rt_camera), vp, vp, u32, u32, vp]
    lib.rt_hip_render_interleaved.restype = i32
    lib.rt_hip_deinterleave.argtypes = [vp, vp, u32, u32, u32, vp]
    lib.rt_hip_deinterleave.restype = i32
    lib.rt_hip_finish.argtypes = [vp, P(A.rt_hip_stats)]
    lib.rt_hip_finish.restype = i32
    lib.rt_hip_render_stats.argtypes = [vp, P(A.rt_camera), vp, vp, u32, u32, vp, P(A.rt_hip_stats)]
    lib.rt_hip_render_stats.restype = i32
    lib.rt_hip_ctx_set_timing.argtypes = [vp, i32]
    lib.rt_hip_ctx_set_timing.restype = None
    lib.rt_hip_ctx_set_batch_samples.argtypes = [vp, u64]
    lib.rt_hip_ctx_set_batch_samples.restype = None
    lib.rt_hip_ctx_set_shading.argtypes = [vp, i32]
    lib.rt_hip_ctx_set_shading.restype = i32
    lib.rt_hip_ctx_set_numerics.argtypes = [vp, i32]
    lib.rt_hip_ctx_set_numerics.restype = i32
    lib.rt_hip_ctx_set_specialization.argtypes = [vp, i32]
    lib.rt_hip_ctx_set_specialization.restype = None
    lib.rt_hip_ctx_set_jit.argtypes = [vp, i32]
    lib.rt_hip_ctx_set_jit.restype = i32
    lib.rt_hip_ctx_jit_status.argtypes = [vp, C.c_char_p, C.c_size_t, C.POINTER(C.c_double)]
    lib.rt_hip_ctx_jit_status.restype = i32
    lib.rt_hip_ctx_set_launch_plan.argtypes = [vp, i32]
    lib.rt_hip_ctx_set_launch_plan.restype = None
    lib.rt_hip_ctx_set_graph.argtypes = [vp, i32]
    lib.rt_hip_ctx_set_graph.restype = None
    lib.rt_hip_ctx_set_pipeline.argtypes = [vp, i32]
    lib.rt_hip_ctx_set_pipeline.restype = i32
    lib.rt_launch_plan_order.argtypes = [vp, u32, u32, vp]
    lib.rt_launch_plan_order.restype = None
    lib.rt_hip_ctx_set_tail_stream.argtypes = [vp, vp]
    lib.rt_hip_ctx_set_tail_stream.restype = i32
    lib.rt_hip_last_error.restype = C.c_char_p
    lib.rt_hip_version.restype = C.c_char_p
    lib.rt_hip_selftest_arith.argtypes = [vp, vp, vp, u32, vp, vp, vp]
    lib.rt_hip_selftest_arith.restype = i32
    lib.rt_poly_plane.argtypes = [P(A.rt_plane), P(A.rt_polygon)]
    lib.rt_poly_plane.restype = None
    lib.rt_scene_builtin.argtypes = [P(A.rt_scene_t)]
    lib.rt_scene_builtin.restype = i32
    lib.rt_parse_scene.argtypes = [C.c_char_p, P(A.rt_scene_t)]
    lib.rt_parse_scene.restype = i32
    lib.rt_load_scene.argtypes = [C.c_char_p, P(A.rt_scene_t)]
    lib.rt_load_scene.restype = i32
    lib.rt_scene_free.argtypes = [P(A.rt_scene_t)]
    lib.rt_scene_free.restype = None
    lib.rt_scene_synthetic.argtypes = [P(A.rt_scene_t), u64, u32, u32, u32, i32]
    lib.rt_scene_synthetic.restype = i32
    lib.rt_write_image.argtypes = [C.c_char_p, vp, C.c_uint, C.c_uint]
    lib.rt_write_image.restype = i32
    lib.write_image.argtypes = [C.c_char_p, vp, C.c_uint, C.c_uint]
    lib.write_image.restype = i32
    _lib = lib
    return lib


class RtHipError(RuntimeError):
    pass


def check(rc, what):
    if rc != 0:
        msg = load().rt_hip_last_error().decode(errors="replace")
        raise RtHipError(f"{what} failed with code {rc}: {msg}")
