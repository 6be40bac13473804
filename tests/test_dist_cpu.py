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
