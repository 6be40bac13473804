"""raytracer_amd - MI355X-native trace-and-shade path behind the reference's raytrace() boundary."""
from .api import (Context, RtHipError, Scene, free_scene, load_scene, parse_scene, prepare_camera, raytrace,
                  render, tone, tone_apply, tone_static, tone_sums, write_image, TONE_WARD, TONE_REINHARD,
                  SHADE_PHONG, SHADE_PHONGBLINN, NUMERICS_EXACT, NUMERICS_FAST, PIPELINE_AUTO, PIPELINE_FUSED,
                  JIT_OFF, JIT_AUTO, JIT_NOW, JIT_NOW_ANY)

__all__ = ["Scene", "Context", "parse_scene", "load_scene", "free_scene", "prepare_camera", "raytrace", "render",
           "write_image", "RtHipError", "tone", "tone_static", "tone_sums", "tone_apply", "TONE_WARD", "TONE_REINHARD",
           "SHADE_PHONG", "SHADE_PHONGBLINN", "NUMERICS_EXACT", "NUMERICS_FAST", "PIPELINE_AUTO", "PIPELINE_FUSED",
           "JIT_OFF", "JIT_AUTO", "JIT_NOW", "JIT_NOW_ANY"]
