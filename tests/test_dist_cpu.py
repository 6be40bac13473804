"""The N > 1 path on CPU: two gloo ranks split a frame into cyclic 8-row blocks, "render" their
blocks (the oracle stands in for the GPU renderer - allowed in tests only), all-gather and
de-interleave; the result must equal the single-process frame."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, W, H, spp, depth, outdir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import raytracer_amd as rt
    from oracle import oracle_py as O
    from raytracer_amd.dist import FrameGather, RowPlan
    sc = rt.Scene.builtin()
    cam = O.prepare_camera(sc.c, W, H, spp, depth)
    plan = RowPlan(H, world)
    fg = FrameGather(plan, W, torch.int32, "cpu", rank=rank)
    fr = FrameGather(plan, W, torch.float32, "cpu", trailing=(3,), rank=rank)
    for y, n, ly in plan.rows_of(rank):
        a, r, _ = O.render(sc.c, cam, y, n)
        fg.local[ly:ly + n] = torch.from_numpy(a[y:y + n].view(np.int32))
        fr.local[ly:ly + n] = torch.from_numpy(r[y:y + n])
    frame = fg.gather().numpy().view(np.uint32)
    rad = fr.gather().numpy()
    np.save(os.path.join(outdir, f"argb{rank}.npy"), frame)
    np.save(os.path.join(outdir, f"rad{rank}.npy"), rad)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("W,H", [(96, 64), (70, 45)])   # 45 rows: 6 blocks, the last one ragged
def test_two_ranks_rebuild_the_frame(tmp_path, W, H):
    sys.path.insert(0, ROOT)
    import raytracer_amd as rt
    from oracle import oracle_py as O
    world, spp, depth = 2, 2, 3
    mp.spawn(_worker, args=(world, _free_port(), W, H, spp, depth, str(tmp_path)), nprocs=world, join=True)
    sc = rt.Scene.builtin()
    a, r, _ = O.render(sc.c, O.prepare_camera(sc.c, W, H, spp, depth))
    for rank in range(world):
        assert np.array_equal(np.load(tmp_path / f"argb{rank}.npy"), a)
        assert np.array_equal(np.load(tmp_path / f"rad{rank}.npy"), r)


def test_row_plan_covers_every_row_once():
    from raytracer_amd.dist import RowPlan, deinterleave
    for H in (1, 7, 8, 9, 45, 1080, 2160, 4320):
        for world in (1, 2, 3, 4, 8):
            plan = RowPlan(H, world)
            seen = np.zeros(H, dtype=int)
            for r in range(world):
                for y, n, ly in plan.rows_of(r):
                    seen[y:y + n] += 1
                    assert ly + n <= plan.local_rows
            assert (seen == 1).all()
            # de-interleave of "local row -> frame row" labels
            g = torch.full((world * plan.local_rows,), -1, dtype=torch.int64)
            for r in range(world):
                for y, n, ly in plan.rows_of(r):
                    g[r * plan.local_rows + ly: r * plan.local_rows + ly + n] = torch.arange(y, y + n)
            assert torch.equal(deinterleave(g, plan), torch.arange(H))


def _pipeline_worker(rank, world, port, W, H, nframes, inflight, gather_stream, outdir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from raytracer_amd.dist import FramePipeline, RowPlan
    plan = RowPlan(H, world)

    def pixel(frame, y, x):  # what "rendering" frame `frame` puts at (x, y)
        return (frame * 1000003 + y * 4099 + x) & 0x7FFFFFFF

    class Stats:
        overflow_retries = 0
    finished = []

    def render(i, k, local, stream):
        # only this rank's own rows are written - into its slice of the slot's gather buffer (in place)
        assert local.data_ptr() == pipe.gathers[k].gathered[rank * plan.local_rows:].data_ptr()
        local.fill_(-7)  # rows of the slot's previous frame must not survive
        xs = torch.arange(W, dtype=torch.int64)
        for y, n, ly in plan.rows_of(rank):
            for d in range(n):
                local[ly + d] = ((i * 1000003 + (y + d) * 4099 + xs) & 0x7FFFFFFF).to(torch.int32)

    def finish(i, k):
        finished.append(i)
        return Stats()

    pipe = FramePipeline(plan, W, torch.int32, "cpu", rank, render, finish, inflight=inflight,
                         gather_stream=gather_stream)
    ok = True
    want_rows = torch.arange(H, dtype=torch.int64)[:, None] * 4099 + torch.arange(W, dtype=torch.int64)[None, :]
    for i in range(nframes):  # bench.py's loop: collect frame i - inflight, then submit frame i
        if i >= inflight:
            pipe.collect(i - inflight)
            got = pipe.frame(i - inflight)
            ok = ok and torch.equal(got.to(torch.int64), ((i - inflight) * 1000003 + want_rows) & 0x7FFFFFFF)
        pipe.submit(i)
    for i in range(max(0, nframes - inflight), nframes):
        pipe.collect(i)
        ok = ok and torch.equal(pipe.frame(i).to(torch.int64), (i * 1000003 + want_rows) & 0x7FFFFFFF)
    assert finished == list(range(nframes))
    # a frame whose batch overflowed after the gather took its rows is refused
    Stats.overflow_retries = 1
    pipe.submit(nframes)
    try:
        pipe.collect(nframes)
        refused = False
    except RuntimeError:
        refused = True
    np.save(os.path.join(outdir, f"ok{rank}.npy"), np.array([ok, refused]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("H,inflight,gather_stream", [(45, 2, "side"), (64, 3, "same"), (5, 2, "side"),
                                                     (45, 6, "side"), (64, 6, "same")])   # six: bench.py's N > 1 default
def test_frames_in_flight_through_the_shared_pipeline(tmp_path, H, inflight, gather_stream):
    """raytracer_amd.dist.FramePipeline - the submit / collect loop bench.py times - with two gloo ranks:
    several frames in flight, each rank rendering into ITS slice of the slot's gather buffer (the in-place
    all-gather), slots re-used frame after frame, a ragged last block (45 rows), a rank without any block
    (5 rows over 2 ranks), and the six frames in flight bench.py keeps on more than one rank.  Every rank must see
    every frame complete and in order."""
    world, W, nframes = 2, 40, max(7, 2 * inflight + 3)   # every slot is used at least twice
    mp.spawn(_pipeline_worker, args=(world, _free_port(), W, H, nframes, inflight, gather_stream, str(tmp_path)),
             nprocs=world, join=True)
    for rank in range(world):
        assert np.load(tmp_path / f"ok{rank}.npy").tolist() == [True, True]
