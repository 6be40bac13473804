"""k_primary's launch plan (rt_hip_ctx_set_launch_plan): a context times the tiles of a frame and launches the
blocks of the next frames of that shape slowest first.  The plan is a permutation of the raster: pixels,
radiance bits and ray counters must not depend on it - checked here against a context that never plans, for
still and moving views, both numerics builds, slabs, interleaved blocks and big (grid) scenes."""
import copy

import numpy as np
import pytest

import raytracer_amd as rt
from test_gpu_parity import render_dev, rays_of

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


def counters(st):
    return (st.primary_rays, st.shade_nodes, st.reflection_rays, st.transmission_rays, st.primary_rays_culled)


@pytest.mark.parametrize("numerics", [rt.NUMERICS_EXACT, rt.NUMERICS_FAST])
@pytest.mark.parametrize("name,W,H,s,d", [("builtin", 640, 360, 2, 6), ("builtin", 333, 217, 3, 4),
                                          ("syn300", 256, 160, 1, 5), ("syn3000", 160, 96, 2, 3)])
def test_planned_frames_equal_raster_frames(torch_cuda, make_scene, numerics, name, W, H, s, d):
    sc = make_scene(name)
    cam = rt.prepare_camera(sc, W, H, s, d)
    plain = rt.Context(queue_cap=1 << 22)
    plain.upload(sc)
    plain.set_numerics(numerics)
    plain.set_launch_plan(False)
    ctx = rt.Context(queue_cap=1 << 22)
    ctx.upload(sc)
    ctx.set_numerics(numerics)
    seen = []
    for _ in range(4):
        # (frame for frame: the tolerance build's first frame of a shape, folded level by level, and its later ones,
        # whose deep levels k_subtree finishes, may differ in the last bit of a few pixels - include/rt_hip.h)
        want_a, want_r, st0 = render_dev(torch_cuda, plain, cam)
        assert st0.planned_launches == 0 and st0.plan_builds == 0
        a, r, st = render_dev(torch_cuda, ctx, cam)
        assert np.array_equal(a, want_a) and np.array_equal(r.view(np.uint32), want_r.view(np.uint32))
        assert counters(st) == counters(st0)
        assert st.batches == 1
        seen.append((st.planned_launches, st.plan_builds))
    # the first frame is timed and the plan built at its end; the others follow it, nothing is rebuilt
    # (big scenes render through the job pipeline, rt_jobs.inc: persistent lanes take rays from a cursor, there is
    # no launch order to plan - the frames must still be identical)
    big = name in ("syn300", "syn3000")
    assert seen == ([(0, 0)] * 4 if big else [(0, 1), (1, 0), (1, 0), (1, 0)]), seen
    # switched off: back to the raster, same frame
    ctx.set_launch_plan(False)
    a, r, st = render_dev(torch_cuda, ctx, cam)
    assert (st.planned_launches, st.plan_builds) == (0, 0)
    assert np.array_equal(a, want_a) and np.array_equal(r.view(np.uint32), want_r.view(np.uint32))
    ctx.close()
    plain.close()


def test_moving_view_keeps_the_plan_and_refreshes_it(torch_cuda, make_scene):
    """a plan taken on another view is still a permutation: it is used as it is and rebuilt every 8th frame"""
    sc = make_scene("builtin")
    base = rt.prepare_camera(sc, 480, 270, 2, 5)
    plain = rt.Context(queue_cap=1 << 22)
    plain.upload(sc)
    plain.set_launch_plan(False)
    ctx = rt.Context(queue_cap=1 << 22)
    ctx.upload(sc)
    builds = []
    for f in range(18):
        cam = copy.copy(base)
        cam.eyePos.x = base.eyePos.x + 0.05 * f
        cam.eyePos.y = base.eyePos.y + 0.02 * f
        want_a, want_r, st0 = render_dev(torch_cuda, plain, cam)
        a, r, st = render_dev(torch_cuda, ctx, cam)
        assert np.array_equal(a, want_a) and np.array_equal(r.view(np.uint32), want_r.view(np.uint32)), f
        assert rays_of(st) == rays_of(st0)
        assert st.planned_launches == (0 if f == 0 else 1)
        builds.append(st.plan_builds)
    assert builds == [1] + [0] * 7 + [1] + [0] * 7 + [1, 0], builds
    ctx.close()
    plain.close()


def test_plan_follows_frame_shape_slabs_and_interleaved_blocks(torch_cuda, make_scene):
    torch = torch_cuda
    sc = make_scene("builtin")
    ctx = rt.Context(queue_cap=1 << 22)
    ctx.upload(sc)
    plain = rt.Context(queue_cap=1 << 22)
    plain.upload(sc)
    plain.set_launch_plan(False)
    # frames of different shapes on one context, each twice: a plan of other blocks is dropped, never followed
    for W, H, s in [(320, 200, 2), (200, 320, 1), (320, 200, 2), (64, 8, 3), (31, 7, 2)]:
        cam = rt.prepare_camera(sc, W, H, s, 4)
        want_a, want_r, _ = render_dev(torch, plain, cam)
        for k in range(2):
            a, r, st = render_dev(torch, ctx, cam)
            assert np.array_equal(a, want_a) and np.array_equal(r.view(np.uint32), want_r.view(np.uint32)), (W, H, k)
            assert st.planned_launches == k
    # a slab of rows is a frame shape of its own
    cam = rt.prepare_camera(sc, 320, 200, 2, 4)
    want_a, want_r, _ = render_dev(torch, plain, cam)
    for k in range(3):
        a, r, st = render_dev(torch, ctx, cam, row0=40, nrows=100)
        assert np.array_equal(a[40:140], want_a[40:140]) and np.array_equal(r[40:140].view(np.uint32), want_r[40:140].view(np.uint32))
        assert st.planned_launches == (1 if k else 0)
    # cyclic blocks of a rank (the multi-GPU partition)
    W, H, nph = 200, 117, 2
    cam = rt.prepare_camera(sc, W, H, 2, 5)
    want_a, want_r, _ = render_dev(torch, plain, cam)
    kmax = ((H + 7) // 8 + nph - 1) // nph
    for k in range(3):
        parts = []
        for ph in range(nph):
            la = torch.zeros((kmax * 8, W), dtype=torch.int32, device="cuda")
            ctx.render_interleaved(cam, la.data_ptr(), 0, ph, nph)
            st = ctx.finish()
            parts.append(la)
        ga = torch.stack(parts).view(nph, kmax, 8, W).permute(1, 0, 2, 3).reshape(-1, W)[:H]
        assert np.array_equal(ga.cpu().numpy().view(np.uint32), want_a), k
    ctx.close()
    plain.close()


def test_plan_is_not_used_for_frames_cut_into_batches_or_instrumented(torch_cuda, make_scene):
    torch = torch_cuda
    sc = make_scene("builtin")
    cam = rt.prepare_camera(sc, 320, 200, 2, 4)
    ctx = rt.Context(queue_cap=1 << 22)
    ctx.upload(sc)
    want_a, want_r, st0 = render_dev(torch, ctx, cam)
    _, _, st1 = render_dev(torch, ctx, cam)
    assert st1.planned_launches == 1
    ctx.set_batch_samples(320 * 64 * 4)  # 64 rows per batch: four batches
    a, r, st = render_dev(torch, ctx, cam)
    assert st.batches == 4 and st.planned_launches == 0 and st.plan_builds == 0
    assert np.array_equal(a, want_a) and np.array_equal(r.view(np.uint32), want_r.view(np.uint32))
    ctx.set_batch_samples(0)
    a, r, st = render_dev(torch, ctx, cam, stats=True)
    assert st.planned_launches == 0 and st.plan_builds == 0
    assert np.array_equal(a, want_a)
    a, r, st = render_dev(torch, ctx, cam)
    assert st.planned_launches == 1 and np.array_equal(a, want_a)
    ctx.close()


def test_plan_survives_an_overflow_retry(torch_cuda, make_scene):
    """a batch that overflows its queues is rendered again in halves: no plan for those, none built from them"""
    torch = torch_cuda
    sc = make_scene("builtin")
    cam = rt.prepare_camera(sc, 640, 480, 2, 8)
    big = rt.Context(queue_cap=1 << 23)
    big.upload(sc)
    big.set_launch_plan(False)
    want_a, want_r, _ = render_dev(torch, big, cam)
    big.close()
    ctx = rt.Context(queue_cap=1 << 18)  # too small for the frame in one batch
    ctx.upload(sc)
    ctx.set_batch_samples(1 << 40)  # ... but planned as one
    for k in range(3):
        a, r, st = render_dev(torch, ctx, cam)
        assert np.array_equal(a, want_a) and np.array_equal(r.view(np.uint32), want_r.view(np.uint32)), k
        if st.overflow_retries:
            assert st.plan_builds == 0
    ctx.close()


def test_random_scenes_planned_and_through_either_pipeline(torch_cuda):
    """80 random cases - scenes of all three kernel classes, lights off, the eye moved at random (also into the
    sphere cloud), ragged frames, 1-4 sample rows, depth 0-10 - rendered three times by a context that plans its
    launches and, big scenes, runs the job pipeline (csrc/rt_jobs.inc), against a context that does neither: the
    raster order, the fused kernels for every scene.  No pixel may differ in any bit (exact build; the tolerance
    build contracts differently in the two pipelines: most pixels within 1e-4 there), no counter at all, and no frame from the next."""
    rng = np.random.default_rng(20261111)
    for case in range(80):
        cls = int(rng.integers(0, 4))
        if cls == 0:
            sc, desc = rt.Scene.builtin(), "builtin"
        else:
            n_sph = int([rng.integers(0, 40), rng.integers(40, 600), rng.integers(2049, 3200)][cls - 1])
            n_pol = int(rng.integers(0, 5)) or int(n_sph == 0)
            n_l = int(rng.choice([1, 1, 2, 3, 8, 33])) if cls < 3 else int(rng.integers(1, 4))
            args = (int(rng.integers(1, 2**31)), n_sph, n_pol, n_l, int(rng.integers(0, 2)))
            sc, desc = rt.Scene.synthetic(*args), f"synthetic{args}"
        if rng.random() < 0.1:
            sc.c.nLights = 0
        if rng.random() < 0.5:
            sc.c.eyePos.x += float(rng.normal(0, 2))
            sc.c.eyePos.y += float(rng.normal(0, 1))
            sc.c.eyePos.z += float(rng.normal(0, 6))
        big = cls == 3
        W, H = int(rng.integers(1, 60 if big else 150)), int(rng.integers(1, 40 if big else 100))
        s, d = int(rng.choice([1, 2, 2, 3, 4, 4])), int(rng.integers(0, 11))
        cam = rt.prepare_camera(sc, W, H, s, d)
        numerics = rt.NUMERICS_FAST if rng.random() < 0.3 else rt.NUMERICS_EXACT
        plain = rt.Context(queue_cap=1 << 21)
        plain.upload(sc)
        plain.set_numerics(numerics)
        plain.set_launch_plan(False)
        plain.set_pipeline(rt.PIPELINE_FUSED)
        ctx = rt.Context(queue_cap=1 << 21)
        ctx.upload(sc)
        ctx.set_numerics(numerics)
        jobs = sc.c.nObjects > 128 + 4  # (a grid scene: more than 128 spheres)
        first = None
        for k in range(3):
            want_a, want_r, st0 = render_dev(torch_cuda, plain, cam)
            a, r, st = render_dev(torch_cuda, ctx, cam)
            what = f"case {case} render {k}: {W}x{H} s{s} d{d} {desc} numerics {numerics}"
            if numerics == rt.NUMERICS_EXACT or not jobs:
                assert np.array_equal(a, want_a) and np.array_equal(r.view(np.uint32), want_r.view(np.uint32)), what
                assert counters(st) == counters(st0), what
            else:
                # (a sphere cloud through the tolerance build: the two pipelines round a few operations differently - the job
                # pipeline makes a sample's ray a second time for its child rays, contracted as that kernel's compiler saw
                # fit - and depth-9 trees amplify a flipped hit; both are as far from the exact build as each other, see
                # tools/fast_pipelines_probe.py and tests/test_gpu_fast.py for what the tolerance build is held to there)
                err = np.abs(r - want_r) / np.maximum(np.abs(want_r), 1e-3)
                assert (err <= 1e-4).all(axis=2).mean() >= 0.7 and (err <= 1e-2).all(axis=2).mean() >= 0.95, what
            if jobs:  # frame after frame the same bits: the job pipeline's atomics order nothing that shows
                first = r.copy() if first is None else first
                assert np.array_equal(r.view(np.uint32), first.view(np.uint32)), what
        ctx.close()
        plain.close()


def test_frames_as_graphs_show_the_same_pixels(torch_cuda):
    """rt_hip_ctx_set_graph: once the launches of a frame repeat they are captured and replayed as one graph launch.
    Same pixels and counters as the plain launches; one capture for a run of equal frames, a new one when the view
    changes, the plain way again for the first frames of a new shape and - sampled - with timing on."""
    torch = torch_cuda
    sc = rt.Scene.builtin()
    W, H, s, d = 640, 360, 2, 8
    cam = rt.prepare_camera(sc, W, H, s, d)
    plain = rt.Context()
    plain.upload(sc)
    want, want_rad, st0 = render_dev(torch, plain, cam)
    g = rt.Context()
    g.upload(sc)
    g.set_graph(True)
    stream = torch.cuda.Stream()
    argb = torch.zeros((H, W), dtype=torch.int32, device="cuda")
    rad = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()

    def frame(c):
        argb.zero_(); rad.zero_()
        torch.cuda.synchronize()
        g.render_rows(c, argb.data_ptr(), rad.data_ptr(), 0, H, stream.cuda_stream)
        st = g.finish()
        return argb.cpu().numpy().view(np.uint32), rad.cpu().numpy(), st

    launches = captures = 0
    for i in range(8):
        a, r, st = frame(cam)
        assert np.array_equal(a, want) and np.array_equal(r.view(np.uint32), want_rad.view(np.uint32)), i
        assert (st.shade_nodes, st.reflection_rays, st.transmission_rays) == (st0.shade_nodes, st0.reflection_rays, st0.transmission_rays)
        launches += st.graph_launches
        captures += st.graph_captures
    assert launches >= 4 and 1 <= captures <= 2, (launches, captures)   # (the launch plan's arrival changes the launches once)
    # another view: the plain way while its level counts are unknown, then a graph of its own
    cam2 = rt.prepare_camera(sc, W, H, s, 5)
    want2, want2_rad, _ = render_dev(torch, plain, cam2)
    seen = []
    for i in range(5):
        a, r, st = frame(cam2)
        assert np.array_equal(a, want2) and np.array_equal(r.view(np.uint32), want2_rad.view(np.uint32)), i
        seen.append((st.graph_launches, st.graph_captures))
    assert seen[0] == (0, 0) and seen[-1] == (1, 0) and sum(c for _, c in seen) <= 2, seen
    # timing on: every 8th frame goes the plain way between its events
    g.set_timing(True)
    timed = 0
    for i in range(16):
        a, r, st = frame(cam2)
        assert np.array_equal(a, want2)
        assert st.graph_launches + st.k0_launches == 1
        timed += st.k0_launches
    assert timed == 2
    g.close()
    plain.close()
