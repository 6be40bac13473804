"""Host-side mirror of the reference's interface for the trace-and-shade path.

Names follow the reference: parse_scene / free_scene (scene.h:67-70), raytrace (raytrace.h:17-20),
write_image (output.h:15-16).  Everything computes through librt_hip.so (HIP kernels); nothing
here renders on the CPU.
"""
import ctypes as C
import os

import numpy as np

from . import _abi as A
from ._lib import RtHipError, check, load


class Scene:
    """Owns an rt_scene_t (the reference's scene_t) built by the library's host helpers."""

    def __init__(self):
        self.c = A.rt_scene_t()
        self._owned = False

    @classmethod
    def builtin(cls):
        s = cls()
        if not load().rt_scene_builtin(C.byref(s.c)):
            raise MemoryError("rt_scene_builtin")
        s._owned = True
        return s

    @classmethod
    def synthetic(cls, seed, n_spheres, n_polys=4, n_lights=1, kind=0):
        s = cls()
        if not load().rt_scene_synthetic(C.byref(s.c), seed, n_spheres, n_polys, n_lights, kind):
            raise ValueError("rt_scene_synthetic: bad arguments")
        s._owned = True
        return s

    def view(self):
        return A.scene_view(self.c)

    def free(self):
        if self._owned:
            load().rt_scene_free(C.byref(self.c))
            self._owned = False

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def load_scene(filename):
    """rt_load_scene: really reads the grammar of the reference's `scene` file."""
    s = Scene()
    if not load().rt_load_scene(os.fsencode(filename), C.byref(s.c)):
        raise ValueError(f"rt_load_scene: cannot load {filename!r} (the reason was printed to stdout)")
    s._owned = True
    return s


def parse_scene(filename):
    """scene.c:24 - returns a Scene (the reference ignores the file's contents) or raises."""
    s = Scene()
    if not load().rt_parse_scene(str(filename).encode(), C.byref(s.c)):
        raise OSError(f"Error opening file {{{filename}}} for reading.")
    s._owned = True
    return s


def free_scene(scene):
    scene.free()


def prepare_camera(scene, width, height, sqrt_spp, depth, fov_y=45.0, aspect=None, near_z=1.0):
    """prepare_scene raytrace.c:72-85 with main.c:159-160's projection defaults."""
    cam = A.rt_camera()
    if aspect is None:
        aspect = np.float32(width) / np.float32(height)
    load().rt_camera_prepare(C.byref(cam), C.byref(scene.c), width, height, sqrt_spp,
                             float(fov_y), float(aspect), float(near_z), depth)
    return cam


def raytrace(buffer, scene, fovY, aspectRatio, nearZ, farZ, width, height, samplesPerPixelSq, depth):
    """raytrace.h:17-20 - fills the caller's uint32 buffer (numpy, W*H) with 0xFFRRGGBB pixels."""
    buf = np.ascontiguousarray(buffer, dtype=np.uint32)
    if buf.size != width * height:
        raise ValueError("buffer must hold width*height uint32 pixels")
    load().raytrace(buf.ctypes.data, C.byref(scene.c), fovY, aspectRatio, nearZ, farZ, width, height,
                    samplesPerPixelSq, depth)
    if buf is not buffer:
        np.copyto(buffer.reshape(-1), buf.reshape(-1))
    return buffer


def render(scene, cam, row0=0, nrows=None, want_radiance=True, devices=None):
    """rt_hip_render (or, with a device list, rt_hip_render_devices) on host arrays: returns
    (argb[H,W] uint32, radiance[H,W,3] float32 or None)."""
    W, H = cam.frameBufferWidth, cam.frameBufferHeight
    if nrows is None:
        nrows = H - row0
    argb = np.zeros((H, W), dtype=np.uint32)
    rad = np.zeros((H, W, 3), dtype=np.float32) if want_radiance else None
    fb = A.rt_framebuffer()
    fb.argb = argb.ctypes.data_as(C.POINTER(C.c_uint32))
    fb.radiance = rad.ctypes.data_as(C.POINTER(C.c_float)) if want_radiance else None
    fb.width, fb.height, fb.row0, fb.nrows = W, H, row0, nrows
    view = scene.view()
    if devices is not None:  # rt_hip_render_devices: whole frames over several GPUs of this process
        devs = (C.c_int * len(devices))(*devices)
        check(load().rt_hip_render_devices(C.byref(view), C.byref(cam), C.byref(fb), devs, len(devices)),
              "rt_hip_render_devices")
        return argb, rad
    check(load().rt_hip_render(C.byref(view), C.byref(cam), C.byref(fb)), "rt_hip_render")
    return argb, rad


TONE_WARD, TONE_REINHARD = 1, 2
SHADE_PHONG, SHADE_PHONGBLINN = 0, 1
NUMERICS_EXACT, NUMERICS_FAST = 0, 1
JIT_OFF, JIT_AUTO, JIT_NOW, JIT_NOW_ANY = 0, 1, 2, 3
PIPELINE_AUTO, PIPELINE_FUSED = 0, 1


def tone(d_radiance, d_argb, npixels, op, ld_max, l_max, key_pixel=-1, stream=0):
    """rt_hip_tone: apply_tone (raytrace.c:792-871) on a DEVICE radiance frame, in place; d_argb may be 0."""
    check(load().rt_hip_tone(d_radiance, d_argb or None, npixels, op, ld_max, l_max, key_pixel, stream or None),
          "rt_hip_tone")


def tone_static(d_radiance, d_argb, npixels, op, ld_max, l_max, key_pixel=-1, stream=0):
    """rt_hip_tone_static: rt_hip_tone without any allocation (static scratch block in the library)."""
    check(load().rt_hip_tone_static(d_radiance, d_argb or None, npixels, op, ld_max, l_max, key_pixel, stream or None),
          "rt_hip_tone_static")


def tone_sums(d_radiance, npixels, l_max, key_pixel, d_sums, stream=0):
    check(load().rt_hip_tone_sums(d_radiance, npixels, l_max, key_pixel, d_sums, stream or None), "rt_hip_tone_sums")


def tone_apply(d_radiance, d_argb, npixels, op, ld_max, l_max, d_sums, total_pixels, use_key=False, stream=0):
    check(load().rt_hip_tone_apply(d_radiance, d_argb or None, npixels, op, ld_max, l_max, d_sums, total_pixels,
                                   int(bool(use_key)), stream or None), "rt_hip_tone_apply")


def write_image(filename, buffer, width, height):
    """output.h:15-16 - raw P6; returns 1 on success, 0 on failure like the reference."""
    buf = np.ascontiguousarray(buffer, dtype=np.uint32)
    return load().rt_write_image(str(filename).encode(), buf.ctypes.data, width, height)


class Context:
    """Device-resident renderer: scene and frame buffer stay in HBM (bench / multi-GPU path)."""

    def __init__(self, device=-1, queue_cap=0):
        self._h = C.c_void_p()
        check(load().rt_hip_ctx_create(C.byref(self._h), device, queue_cap), "rt_hip_ctx_create")

    def upload(self, scene):
        view = scene.view()
        check(load().rt_hip_scene_upload(self._h, C.byref(view)), "rt_hip_scene_upload")

    def set_timing(self, on):
        """False / 0: off; True / 1: HIP events around k_primary's dispatch; 2: the kernel's own start and end (rt_hip_ctx_set_timing)"""
        load().rt_hip_ctx_set_timing(self._h, int(on))

    def set_shading(self, model):
        """SHADE_PHONG (get_shade_color_phong, what the reference calls) or SHADE_PHONGBLINN (its unused
        get_shade_color_phongblinn, raytrace.c:574-685)"""
        check(load().rt_hip_ctx_set_shading(self._h, int(model)), "rt_hip_ctx_set_shading")

    def set_numerics(self, mode):
        """NUMERICS_EXACT (the reference's rounding, bit for bit) or NUMERICS_FAST (the north star's tolerance:
        fused multiply-adds, 1-ulp rcp / rsq / sqrt) - rt_hip_ctx_set_numerics"""
        check(load().rt_hip_ctx_set_numerics(self._h, int(mode)), "rt_hip_ctx_set_numerics")

    def set_graph(self, on):
        """frames whose launches repeat are captured once and replayed as one HIP graph launch (rt_hip_ctx_set_graph)"""
        load().rt_hip_ctx_set_graph(self._h, int(bool(on)))

    def set_launch_plan(self, on):
        """False: k_primary's blocks always in raster order; True (the default): slowest first, from the times the
        tiles of the last frame of this shape took (rt_hip_ctx_set_launch_plan)"""
        load().rt_hip_ctx_set_launch_plan(self._h, int(bool(on)))

    def set_pipeline(self, mode):
        """PIPELINE_AUTO (big scenes through the job pipeline, csrc/rt_jobs.inc) or PIPELINE_FUSED (every scene through
        the fused primary / bounce kernels) - rt_hip_ctx_set_pipeline"""
        check(load().rt_hip_ctx_set_pipeline(self._h, int(mode)), "rt_hip_ctx_set_pipeline")

    def set_jit(self, mode):
        """JIT_OFF / JIT_AUTO / JIT_NOW / JIT_NOW_ANY: trace kernels compiled at run time for the scene's shape (rt_hip_ctx_set_jit)"""
        check(load().rt_hip_ctx_set_jit(self._h, int(mode)), "rt_hip_ctx_set_jit")

    def jit_status(self):
        """(state, compile seconds, compiler log): 0 nothing asked for, 1 compiling, 2 ready, -1 failed"""
        log = C.create_string_buffer(4096)
        sec = C.c_double(0.0)
        st = load().rt_hip_ctx_jit_status(self._h, log, len(log), C.byref(sec))
        return st, sec.value, log.value.decode(errors="replace")

    def set_specialization(self, on):
        """False: generic kernels also for the reference's scene shape (rt_hip_ctx_set_specialization)"""
        load().rt_hip_ctx_set_specialization(self._h, int(bool(on)))

    def set_tail_stream(self, stream):
        """hipStream_t (int) for the latency-bound tail of every batch, 0 = none (rt_hip_ctx_set_tail_stream)"""
        check(load().rt_hip_ctx_set_tail_stream(self._h, stream or None), "rt_hip_ctx_set_tail_stream")

    def set_batch_samples(self, n):
        load().rt_hip_ctx_set_batch_samples(self._h, int(n))

    def render_rows(self, cam, d_argb, d_radiance=0, row0=0, nrows=None, stream=0):
        if nrows is None:
            nrows = cam.frameBufferHeight - row0
        check(load().rt_hip_render_rows(self._h, C.byref(cam), d_argb, d_radiance or None, row0, nrows,
                                        stream or None), "rt_hip_render_rows")

    def render_interleaved(self, cam, d_argb_local, d_radiance_local=0, phase=0, nphases=1, stream=0):
        """8-row blocks b with b % nphases == phase, into a compact local buffer (multi-GPU split)."""
        check(load().rt_hip_render_interleaved(self._h, C.byref(cam), d_argb_local, d_radiance_local or None,
                                               phase, nphases, stream or None), "rt_hip_render_interleaved")

    def finish(self):
        st = A.rt_hip_stats()
        check(load().rt_hip_finish(self._h, C.byref(st)), "rt_hip_finish")
        return st

    def render_stats(self, cam, d_argb, d_radiance=0, row0=0, nrows=None, stream=0):
        if nrows is None:
            nrows = cam.frameBufferHeight - row0
        st = A.rt_hip_stats()
        check(load().rt_hip_render_stats(self._h, C.byref(cam), d_argb, d_radiance or None, row0, nrows,
                                         stream or None, C.byref(st)), "rt_hip_render_stats")
        return st

    def close(self):
        if self._h:
            load().rt_hip_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


__all__ = ["Scene", "Context", "parse_scene", "free_scene", "prepare_camera", "raytrace", "render",
           "write_image", "RtHipError"]
