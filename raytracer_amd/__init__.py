"""raytracer_amd - MI355X-native trace-and-shade path behind the reference's raytrace() boundary."""
from .api import (Context, RtHipError, Scene, free_scene, load_scene, parse_scene, prepare_camera, raytrace,
                  render, write_image)

__all__ = ["Scene", "Context", "parse_scene", "load_scene", "free_scene", "prepare_camera", "raytrace", "render",
           "write_image", "RtHipError"]
