"""bench.py's one-line JSON contract, on the GPU box: a short N = 1 run, and the one-GPU rehearsal of a
rank of an N = 2 run (same code path as under torch.distributed.run, the collective left out)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(args, env_extra=None):
    env = dict(os.environ)
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True,
                       timeout=170, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


@pytest.mark.parametrize("tail", ["on", "off"])
def test_single_gpu_line_has_the_contract_fields(tail):
    d = run_bench(["--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--tail-stream", tail, "--numerics", "exact"] +
                  ([] if tail == "on" else ["--no-one-shot"]))
    if tail == "on":   # the blocking raytrace() call in child processes: cold from exec to exit, warm per call
        for num in ("exact", "fast"):
            o = d["one_shot"][num]
            assert 0.2 < o["warm_call_ms_min"] <= o["warm_call_ms_p50"] < 2.0 and 20.0 < o["cold_process_ms"] < 5000.0
    assert d["config"]["streams"] == ("shared + tail stream" if tail == "on" else "shared")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert (d["n_gpus"], d["steps"], d["warmup"], d["unit"], d["scaling"], d["dtype"]) == (1, 6, 2, "Mrays/s", "weak", "f32")
    assert d["config"]["workload"].startswith("c2:") and d["config"]["numerics"] == "exact"
    assert 0 < d["config"]["rays_traced_per_step"] < d["config"]["rays_per_step"]   # sky tiles are never generated
    assert d["generic_kernels"]["k_primary_ms"] > d["roofline"]["avg_launch_ms"] * 0.9
    alone = d["roofline"]["one_frame_at_a_time"]   # the kernel without a neighbour: not slower than beside one
    # (six timed frames on a GPU whose clocks may still be settling: a loose bound)
    assert 0 < alone["k_primary_ms"] <= d["roofline"]["avg_launch_ms"] * 1.35 and 0.05 < alone["frac"] < 1.0
    assert d["step_ms"]["p50"] <= d["step_ms"]["p95"] <= d["step_ms"]["max"]
    assert 0 < d["config"]["device_mib_per_context"] <= 1024      # sized from what the frame uses (0.08 entries a sample)
    assert 0 < d["roofline"]["frac_traced"] < d["roofline"]["frac"]
    assert alone["ms_per_step_one_frame_at_a_time"] > d["ms_per_step"] * 0.9
    assert d["vs_baseline"] is None and d["higher_is_better"] is True
    assert "workload" in d["config"] and "1920x1080" in d["config"]["workload"]
    assert d["config"]["rays_per_step"] == 57693197          # config 2's ray count, equal to the reference's
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert 0.05 < r["frac"] < 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert d["value"] > 2e4 and abs(d["value"] - d["config"]["rays_per_step"] / d["ms_per_step"] / 1e3) < 1e-3 * d["value"]


@pytest.mark.parametrize("gather,tail", [("side", "on"), ("same", "on"), ("side", "off")])
def test_rank_of_a_two_gpu_run_rehearsed_on_one_gpu(gather, tail):
    d = run_bench(["--gpus", "2", "--steps", "6", "--warmup", "2", "--gather-stream", gather, "--tail-stream", tail],
                  {"RT_BENCH_REHEARSE_WORLD": "2", "RT_BENCH_REHEARSE_RANK": "1"})
    assert d["n_gpus"] == 2 and "REHEARSAL" in d["data"]
    assert d["value"] > 1e4
    # N > 1 renders BASELINE config 3 by default: the 3840x2160 frame, fixed, split over the ranks
    assert d["scaling"] == "strong" and d["config"]["workload"].startswith("c3:") and "3840x2160" in d["config"]["workload"]
    assert d["config"]["numerics"] == "fast"
    # more than one rank: every frame in flight on a stream of its own, six of them (a rank's share does not fill its GPU)
    assert d["config"]["streams"] == "one per frame" and d["config"]["frames_in_flight"] == 6


def test_rank_of_a_multi_gpu_run_on_the_shared_stream():
    d = run_bench(["--gpus", "4", "--steps", "6", "--warmup", "2", "--shared-stream"],
                  {"RT_BENCH_REHEARSE_WORLD": "4", "RT_BENCH_REHEARSE_RANK": "2"})
    assert d["n_gpus"] == 4 and d["config"]["streams"] == "shared + tail stream" and d["config"]["frames_in_flight"] == 3
    assert d["value"] > 1e4


def test_other_workloads_and_weak_scaling_are_selectable():
    d = run_bench(["--config", "c1", "--steps", "4", "--warmup", "1", "--no-cpu-baseline", "--no-one-shot"])
    assert "640x480" in d["config"]["workload"] and d["n_gpus"] == 1 and 0 < d["roofline"]["frac"] < 1.0
    d = run_bench(["--config", "c4", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"])
    assert "1024 spheres" in d["config"]["workload"] and "3840x2160" in d["config"]["workload"]
    assert d["value"] > 1e3 and "generic_kernels" not in d
    assert d["config"]["numerics"] == "exact"   # the sphere clouds default to the build that is pinned there
    d = run_bench(["--gpus", "2", "--config", "c2", "--scaling", "weak", "--steps", "4", "--warmup", "1"],
                  {"RT_BENCH_REHEARSE_WORLD": "2", "RT_BENCH_REHEARSE_RANK": "0"})
    assert d["scaling"] == "weak" and "1920x1080" not in d["config"]["workload"]
    d = run_bench(["--gpus", "8", "--config", "c2", "--steps", "4", "--warmup", "1"],
                  {"RT_BENCH_REHEARSE_WORLD": "8", "RT_BENCH_REHEARSE_RANK": "3"})
    assert d["scaling"] == "strong" and "1920x1080" in d["config"]["workload"] and d["n_gpus"] == 8
