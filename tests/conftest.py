import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(GOLDEN_DIR, "golden.json")) as f:
        meta = json.load(f)
    arrays = np.load(os.path.join(GOLDEN_DIR, "golden.npz"), allow_pickle=False)
    return meta, arrays


@pytest.fixture(scope="session")
def pins():
    """tests/golden/pins.npz: 32 768 + 8192 pixels of configs 4 and 5 at full frame size, from the untouched reference
    (tests/golden/make_golden_pins.py)."""
    with open(os.path.join(GOLDEN_DIR, "pins.json")) as f:
        meta = json.load(f)
    arrays = np.load(os.path.join(GOLDEN_DIR, "pins.npz"), allow_pickle=False)
    return meta, arrays


@pytest.fixture(scope="session")
def make_scene(golden):
    """name -> raytracer_amd.Scene, from the scene table stored with the fixtures."""
    import raytracer_amd as rt
    meta, _ = golden

    def _make(name):
        a = meta["scenes"][name]
        return rt.Scene.builtin() if a is None else rt.Scene.synthetic(**a)
    return _make


def blinn_cases():
    """(case, scene, expected argb, expected radiance) of tests/golden/blinn.npz: the reference's unused
    second shading routine (raytrace.c:574-685), fixtures made by tests/golden/make_golden_blinn.py."""
    import json
    import raytracer_amd as rt
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    meta = json.load(open(os.path.join(here, "blinn.json")))
    arrays = np.load(os.path.join(here, "blinn.npz"))
    for c in meta["cases"]:
        if c["scene"].endswith(".scene"):
            sc = rt.load_scene(os.path.join(here, c["scene"]))
        else:
            a = meta["scenes"][c["scene"]]
            sc = rt.Scene.builtin() if a is None else rt.Scene.synthetic(**a)
        yield c, sc, arrays[c["key"] + "_argb"], arrays[c["key"] + "_rad"]


def rgb_bytes(argb):
    return np.stack([(argb >> 16) & 255, (argb >> 8) & 255, argb & 255], -1).astype(np.int32)


def ppm_md5(argb, W, H):
    import hashlib
    rgb = rgb_bytes(argb).astype(np.uint8)
    return hashlib.md5(b"P6\n%d %d\n255\n" % (W, H) + rgb.tobytes()).hexdigest()


def parity_report(argb, rad, argb_ref, rad_ref):
    """north_star tolerances: fraction of pixels within +-1 LSB (8-bit) and within 1e-4 relative
    radiance error (denominator floored at 1e-3 so black channels do not divide by zero)."""
    lsb_ok = (np.abs(rgb_bytes(argb) - rgb_bytes(argb_ref)).max(-1) <= 1).mean()
    rel = (np.abs(rad - rad_ref) / np.maximum(np.abs(rad_ref), 1e-3)).max(-1)
    return float(lsb_ok), float((rel <= 1e-4).mean()), float(np.median(rel)), float((rad == rad_ref).all(-1).mean())
