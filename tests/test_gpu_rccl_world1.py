"""The multi-rank path on the one GPU there is: a process group of ONE rank over RCCL (backend "nccl"), so that every
collective call of an N-GPU run - the in-place all-gather of the frame slices, the barriers and all-reduces of bench.py -
executes on hardware at least in its degenerate form.  (Real scaling needs the 8-GPU node only the driver has.)"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import os, sys
sys.path.insert(0, %r)
import numpy as np, torch, torch.distributed as dist
import raytracer_amd as rt
from raytracer_amd.dist import FrameGather, RowPlan
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29531")
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
sc = rt.Scene.builtin(); W, H = 320, 203
cam = rt.prepare_camera(sc, W, H, 2, 6)
ctx = rt.Context(); ctx.upload(sc)
want = torch.zeros((H, W), dtype=torch.int32, device=dev)
ctx.render_rows(cam, want.data_ptr(), 0, 0, H, 0); ctx.finish()
fg = FrameGather(RowPlan(H, 1), W, torch.int32, dev, rank=0, force=True)
st = torch.cuda.current_stream().cuda_stream
ctx.render_interleaved(cam, fg.local.data_ptr(), 0, 0, 1, st); ctx.finish()
frame = fg.gather(collective=True)          # all_gather_into_tensor over RCCL, in place, then the de-interleave kernel
torch.cuda.synchronize()
t = torch.ones(3, dtype=torch.float64, device=dev); dist.all_reduce(t); dist.barrier()
assert torch.equal(frame, want), "gathered frame differs"
assert float(t.sum()) == 3.0
dist.destroy_process_group()
print("RCCL_WORLD1_OK")
"""


def _run(cmd, env=None, timeout=300):
    e = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    e.update(env or {})
    return subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, env=e, timeout=timeout)


def test_frame_gather_over_rccl_with_a_world_of_one():
    p = _run([sys.executable, "-c", SCRIPT % ROOT])
    assert p.returncode == 0 and "RCCL_WORLD1_OK" in p.stdout, p.stdout[-2000:] + p.stderr[-2000:]


def test_bench_multi_rank_path_with_a_world_of_one():
    """bench.py's N > 1 code path (process group, FramePipeline with the gather on its side stream, max / sum
    all-reduces of the timing) with one rank: the line must come out and say what it is."""
    p = _run([sys.executable, "bench.py", "--config", "c1", "--steps", "8", "--warmup", "2", "--no-cpu-baseline",
              "--no-one-shot", "--no-generic"], env={"RT_BENCH_RCCL_WORLD1": "1"})
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert "world of one" in line["data"] and line["n_gpus"] == 1 and line["value"] > 0
