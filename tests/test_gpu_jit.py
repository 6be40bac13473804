"""GPU tests of the trace kernels compiled at run time for a scene's shape (csrc/rt_jit.cpp, rt_hip_ctx_set_jit):
the same source as the library's kernels with the scene's counts as constants, so the exact build's frames may not
differ in any bit, nor any counter; the tolerance build contracts differently and is held to its tolerance."""
import time

import numpy as np
import pytest

import raytracer_amd as rt
from test_gpu_parity import render_dev

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch


def counters(st):
    return (st.primary_rays, st.shadow_rays, st.reflection_rays, st.transmission_rays, st.shade_nodes)


def test_reference_scene_through_run_time_kernels(torch_cuda):
    """The reference's scene, for which the library has kernels of its own, through kernels compiled at run time
    instead (RT_JIT_NOW_ANY): identical frame and counters (exact build), tolerance build within 1e-4."""
    sc = rt.Scene.builtin()
    cam = rt.prepare_camera(sc, 640, 360, 3, 8)
    for numerics in (rt.NUMERICS_EXACT, rt.NUMERICS_FAST):
        plain = rt.Context()
        plain.upload(sc)
        plain.set_numerics(numerics)
        plain.set_jit(rt.JIT_OFF)
        want_a, want_r, st0 = render_dev(torch_cuda, plain, cam)
        ctx = rt.Context()
        ctx.upload(sc)
        ctx.set_numerics(numerics)
        ctx.set_jit(rt.JIT_NOW_ANY)
        for k in range(3):  # (the second and third frames follow a launch plan and take k_subtree)
            a, r, st = render_dev(torch_cuda, ctx, cam)
            state, seconds, log = ctx.jit_status()
            assert state == 2, (state, log)
            if numerics == rt.NUMERICS_EXACT:
                assert np.array_equal(a, want_a) and np.array_equal(r.view(np.uint32), want_r.view(np.uint32)), k
                assert counters(st) == counters(st0)
            else:
                ok = np.abs(r - want_r) <= 1e-4 * np.maximum(np.abs(want_r), 1e-3)
                assert ok.all(axis=2).mean() >= 0.999, k
        ctx.close()
        plain.close()


def test_random_small_scenes_through_run_time_kernels(torch_cuda):
    """24 random scenes of the small class (0-40 spheres, 0-4 quads, 1-3 lights, moved eyes, ragged frames, depth
    0-10), each rendered by the generic kernels and by kernels compiled for its counts: no bit may differ."""
    rng = np.random.default_rng(20261112)
    compiled = 0
    for case in range(24):
        n_sph = int(rng.integers(0, 40))
        n_pol = int(rng.integers(0, 5)) or int(n_sph == 0)
        args = (int(rng.integers(1, 2**31)), n_sph, n_pol, int(rng.choice([1, 1, 2, 3])), int(rng.integers(0, 2)))
        sc = rt.Scene.synthetic(*args)
        if rng.random() < 0.5:
            sc.c.eyePos.x += float(rng.normal(0, 2))
            sc.c.eyePos.z += float(rng.normal(0, 6))
        W, H = int(rng.integers(1, 150)), int(rng.integers(1, 100))
        s, d = int(rng.choice([1, 2, 3, 4])), int(rng.integers(0, 11))
        cam = rt.prepare_camera(sc, W, H, s, d)
        plain = rt.Context(queue_cap=1 << 21)
        plain.upload(sc)
        plain.set_jit(rt.JIT_OFF)
        ctx = rt.Context(queue_cap=1 << 21)
        ctx.upload(sc)
        ctx.set_jit(rt.JIT_NOW)
        for k in range(2):
            want_a, want_r, st0 = render_dev(torch_cuda, plain, cam)
            a, r, st = render_dev(torch_cuda, ctx, cam)
            what = f"case {case} render {k}: {W}x{H} s{s} d{d} synthetic{args}"
            assert np.array_equal(a, want_a) and np.array_equal(r.view(np.uint32), want_r.view(np.uint32)), what
            assert counters(st) == counters(st0), what
        state, seconds, log = ctx.jit_status()
        assert state in (0, 2), (state, log)  # (0: a scene too big for the small class takes the library's kernels)
        compiled += state == 2
        ctx.close()
        plain.close()
    assert compiled >= 20


def test_default_policy_compiles_beside_the_frames(torch_cuda):
    """RT_JIT_AUTO (the default): nothing is compiled for the first RT_JIT_AFTER frames of a scene; then the compile
    runs on a thread of its own while frames go on, and the frames after it show the same pixels."""
    sc = rt.Scene.synthetic(77, 5, 2, 2, 0)
    cam = rt.prepare_camera(sc, 320, 200, 2, 6)
    ctx = rt.Context()
    ctx.upload(sc)
    want_a, want_r, st0 = render_dev(torch_cuda, ctx, cam)
    assert ctx.jit_status()[0] == 0
    for k in range(12):
        a, r, st = render_dev(torch_cuda, ctx, cam)
        assert np.array_equal(a, want_a) and np.array_equal(r.view(np.uint32), want_r.view(np.uint32)), k
    t0 = time.time()
    while ctx.jit_status()[0] == 1 and time.time() - t0 < 60:
        time.sleep(0.05)
    state, seconds, log = ctx.jit_status()
    assert state == 2 and 0.0 < seconds < 60.0, (state, seconds, log)
    for k in range(3):
        a, r, st = render_dev(torch_cuda, ctx, cam)
        assert np.array_equal(a, want_a) and np.array_equal(r.view(np.uint32), want_r.view(np.uint32)), k
        assert counters(st) == counters(st0)
    ctx.close()
