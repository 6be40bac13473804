"""Edges of the input space, each against the oracle bit for bit (libm's powf rounding taken out): frames of extreme
shapes, very many lights, very many spheres, contexts on concurrent host threads.  Small versions of
tools/extreme_frames_probe.py, many_lights_probe.py, many_spheres_probe.py, threads_probe.py."""
import threading

import numpy as np
import pytest

import raytracer_amd as rt
from oracle import oracle_py as O
from test_gpu_parity import render_dev

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


@pytest.fixture()
def rounded_pow():
    O.set_rounded_pow(True)
    yield
    O.set_rounded_pow(False)


def same(argb, rad, a_o, r_o, rows=slice(None)):
    return np.array_equal(rad[rows].view(np.uint32), r_o[rows].view(np.uint32)) and np.array_equal(argb[rows], a_o[rows])


@pytest.mark.parametrize("name", ["builtin", "c4"])
@pytest.mark.parametrize("W,H,s,d,r0,n", [(65536, 9, 1, 3, 0, 9), (7, 200000, 1, 4, 199990, 10), (1, 1, 1, 8, 0, 1),
                                          (2, 2, 16, 5, 0, 2), (1, 9, 31, 2, 3, 5)])
def test_frames_of_extreme_shapes(torch_cuda, make_scene, rounded_pow, name, W, H, s, d, r0, n):
    """65 536 pixels wide, 200 000 rows tall (a band near the end), one pixel, 16 x 16 and 31 x 31 samples a pixel."""
    sc = make_scene(name)
    cam = rt.prepare_camera(sc, W, H, s, d)
    ctx = rt.Context()
    ctx.upload(sc)
    argb, rad, st = render_dev(torch_cuda, ctx, cam, r0, n)
    ctx.close()
    assert st.primary_rays == W * n * s * s
    a_o, r_o, _ = O.render(sc.c, cam, row0=r0, nrows=n, nthreads=16)
    assert same(argb, rad, a_o, r_o, slice(r0, r0 + n))


@pytest.mark.parametrize("n_sph,n_l", [(5, 65), (5, 257), (300, 130), (2500, 65)])
def test_more_lights_than_a_visibility_word_has_bits(torch_cuda, rounded_pow, tmp_path, n_sph, n_l):
    """65 to 257 lights (scene files: rt_scene_synthetic stops at 64) over 5, 300 and 2500 spheres: the fused kernels'
    visibility words beyond the first, and the big scenes leaving the job pipeline (32 lights at most) for them."""
    rng = np.random.default_rng(n_sph + n_l)
    parts = ["camera\n{\nposition 0 4.5 2\nlookat 0 4.5 -8\nupvector 0 1 0\nprojection 45.0 1.0 200.0\n}\n",
             "world\n{\nambient 1.0 1.0 1.0\nbackground .3 .4 .6\n}\n"]
    for _ in range(n_l):
        parts.append("pointlight\n{\nposition %r %r %r\ndiffuse %g %g %g\nrange 150\n}\n"
                     % (float(rng.uniform(-10, 10)), float(rng.uniform(2, 14)), float(rng.uniform(-16, 4)), *(rng.uniform(0.3, 1, 3) / n_l * 3)))
    for k in range(n_sph):
        kr, kt = [(0.0, 0.0), (0.5, 0.0), (0.05, 0.8), (0.3, 0.5)][int(rng.integers(0, 4))]
        parts.append("sphere\n{\nname S%d\nposition %r %r %r\nradius %r\ndiffuse %g %g %g\nspecular 1 1 1\nka 0.1 kd 0.4 ks 0.3 ke 20 kr %g kt %g n 1.25\n}\n"
                     % (k, float(rng.uniform(-3, 3)), 4.5 + float(rng.uniform(-3, 3)), -8.0 + float(rng.uniform(-3, 3)),
                        float(rng.uniform(0.05, 0.5)), *rng.uniform(0.2, 1, 3), kr, kt))
    parts.append("polygon\n{\nname P\npoint 20 0 12\npoint 20 0 -28\npoint -20 0 -28\npoint -20 0 12\ndiffuse 0 1 0\nka 0.1 kd 0.6 ks 0.2 ke 2 kr 0.4 kt 0 n 1\n}\n")
    path = tmp_path / "lights.scene"
    path.write_text("".join(parts))
    sc = rt.load_scene(str(path))
    assert sc.c.nLights == n_l and sc.c.nObjects == n_sph + 1
    cam = rt.prepare_camera(sc, 40, 28, 2, 4)
    argb, rad = rt.render(sc, cam)
    a_o, r_o, _ = O.render(sc.c, cam, nthreads=16)
    assert same(argb, rad, a_o, r_o)


def test_a_hundred_thousand_spheres(torch_cuda, rounded_pow):
    """Six times config 5's sphere count on a small frame (tools/many_spheres_probe.py goes to 2.2 M)."""
    sc = rt.Scene.synthetic(101234, 100000, 2, 2, 1)
    cam = rt.prepare_camera(sc, 48, 32, 1, 6)
    argb, rad = rt.render(sc, cam)
    a_o, r_o, _ = O.render(sc.c, cam, nthreads=16)
    assert same(argb, rad, a_o, r_o)


def test_contexts_on_concurrent_host_threads(torch_cuda, rounded_pow):
    """Six host threads, a context and a stream each, scenes of every kernel class: uploads, grid builds and renders
    overlap (ctypes drops the GIL in the calls); every frame is the oracle's."""
    torch = torch_cuda
    rng = np.random.default_rng(5)
    jobs = [[(rt.Scene.synthetic(int(rng.integers(1, 2**31)), int(rng.choice([3, 30, 200, 2500])), int(rng.integers(1, 4)),
                                 int(rng.choice([1, 2, 8])), int(rng.integers(0, 2))),
              int(rng.integers(16, 64)), int(rng.integers(16, 48)), int(rng.integers(1, 3)), int(rng.integers(0, 7)))
             for _ in range(5)] for _ in range(6)]
    out = [[] for _ in jobs]

    def work(t):
        st = torch.cuda.Stream()
        for sc, W, H, s, d in jobs[t]:
            cam = rt.prepare_camera(sc, W, H, s, d)
            ctx = rt.Context()
            ctx.upload(sc)
            argb = torch.zeros((H, W), dtype=torch.int32, device="cuda")
            rad = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
            st.wait_stream(torch.cuda.current_stream())  # (the fills above ran on torch's current stream, not on st)
            for _ in range(2):
                ctx.render_rows(cam, argb.data_ptr(), rad.data_ptr(), 0, H, st.cuda_stream)
                ctx.finish()
            st.synchronize()
            out[t].append((argb.cpu().numpy().view(np.uint32), rad.cpu().numpy(), cam))
            ctx.close()

    threads = [threading.Thread(target=work, args=(t,)) for t in range(len(jobs))]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    for t in range(len(jobs)):
        assert len(out[t]) == len(jobs[t])  # (a thread that died left fewer)
        for (sc, W, H, s, d), (a, r, cam) in zip(jobs[t], out[t]):
            a_o, r_o, _ = O.render(sc.c, cam, nthreads=16)
            differ = (r.view(np.uint32) != r_o.view(np.uint32)).any(-1)
            assert same(a, r, a_o, r_o), (t, W, H, s, d, sc.c.nObjects, sc.c.nLights, int(differ.sum()), np.argwhere(differ)[:5].tolist())
