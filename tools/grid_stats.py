import sys, os
sys.path.insert(0, os.getcwd())
import raytracer_amd as rt
from bench import CONFIGS
for cfg in ("c4", "c5s"):
    scene_args = CONFIGS[cfg][0]
    sc = rt.Scene.synthetic(*scene_args)
    ctx = rt.Context(); ctx.upload(sc)
