#!/usr/bin/env python3
"""bench.py - throughput of the HIP trace-and-shade path on BASELINE.json's workloads.

    python bench.py --gpus N --steps K --warmup W [--config c2|c3|c4|c5s|c1] [--scaling strong|weak]
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one full render of the frame: every rank renders its row share with the scene and the frame
buffer resident in HBM, then (N > 1) the frame-buffer slices are collected with one RCCL all-gather.

Workloads (BASELINE.json `configs`):
    c2   builtin Whitted scene 1920x1080, 4x4 supersampling, depth 8     the N = 1 default (the metric's config)
    c3   the same scene at 3840x2160                                      the N > 1 default: a FIXED frame split
                                                                          over the ranks ("strong" scaling), as
                                                                          the reference's Cell build splits one
                                                                          image over its workers
    c1   640x480, 1 spp, depth 4 (the reference's CPU-runnable case)
    c4   1024 random spheres + 4 quads, 3840x2160, 16 spp, depth 8
    c5s  the scene of config 5 (16 384 spheres, 8 lights, depth 16) at a REDUCED size, 1920x1080 with 2x2
         samples - config 5 itself (7680x4320, 64 spp) is sized for 8 GPUs
--config c2 --gpus 8 is the metric's "1080p at 8 GPUs".  --scaling weak grows the frame with N instead (N x the
pixels of the config's frame at its aspect ratio; per-GPU work fixed).

Rank 0 prints ONE JSON line.  `value` = rays the REFERENCE traces for the frame (primary + shadow +
reflection + transmission, counted by the kernels and equal to the reference's own counters) of all ranks /
max-over-ranks time; `config.rays_traced_per_step` leaves out the primary rays of pixel tiles that were shown to
see nothing and were never generated.  `roofline` prices the dominant kernel (k_primary) with the algorithmic
FLOP constants of SURVEY.md section 8d against the FP32 vector peak: reference-equivalent work / time, NOT a
utilisation - `roofline.executed_frac` (vector lane operations really issued / what the SIMDs can issue) is
that.  `cpu_baseline` times the reference (or the oracle port) on the host.

Three frames are in flight (a render context each), on ONE render stream - k_primary and the wide
bounce levels, back to back - and ONE tail stream for every frame's sparse deep end (k_subtree), folds, resolve
and publish, which drain beside the next frame's kernels.  `roofline` times k_primary inside that timed region;
`roofline.one_frame_at_a_time` is the same kernel with each frame alone on the GPU.
With more than one rank (a rank's share of the frame does not fill its GPU, and half of a frame's kernels are bound
by latency whatever their size) SIX frames are in flight, each on a stream of its own (--overlap --inflight 6;
--shared-stream: the one-rank arrangement); the k_primary of such a run shares the GPU with its neighbours' kernels.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402,F401
import torch  # noqa: E402

import raytracer_amd as rt  # noqa: E402
from raytracer_amd.dist import FramePipeline, RowPlan  # noqa: E402

PEAK_FP32_TFLOPS = 157.3  # MI355X vector FP32 peak, MI355X_MICROARCH.md "Chip-level parameters"
PMC_FILE = os.path.join(ROOT, "profiles", "r03_pmc.json")
KERNEL_SOURCES = ["rt_kernels.hip", "rt_jobs.inc", "rt_arith.h", "rt_device.h", "rt_dev.h"]


def kernels_sha():
    """what the counters of PMC_FILE are stamped with: a hash of the kernel sources they were collected on"""
    import hashlib
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        with open(os.path.join(ROOT, "raytracer_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]

# name -> (scene, W, H, sqrtSpp, depth, default steps, default warmup); scene None = the reference's hard-coded one,
# else rt_scene_synthetic's arguments (seed, spheres, quads, lights, kind) - SURVEY.md section 8d
CONFIGS = {
    "c1": (None, 640, 480, 1, 4, 200, 30),
    "c2": (None, 1920, 1080, 4, 8, 200, 30),
    "c3": (None, 3840, 2160, 4, 8, 60, 10),
    "c4": ((0xC0FFEE04, 1024, 4, 1, 0), 3840, 2160, 4, 8, 6, 2),
    "c5s": ((0xC0FFEE05, 16384, 0, 8, 1), 1920, 1080, 2, 16, 4, 1),
    "syn64": ((0xC0FFEE64, 64, 4, 1, 0), 1920, 1080, 4, 8, 30, 5),   # a small scene that is NOT the reference's: 64 spheres, 4 quads
    "c5": ((0xC0FFEE05, 16384, 0, 8, 1), 7680, 4320, 8, 16, 2, 1),   # BASELINE config 5 at full size: a job for 8 GPUs
}


def grown_frame(W, H, n_gpus):
    """--scaling weak: n_gpus x the pixels of the WxH frame at its aspect ratio; height a multiple of 8*n_gpus."""
    if n_gpus == 1:
        return W, H
    h = int(round(H * math.sqrt(n_gpus) / (8 * n_gpus))) * 8 * n_gpus
    w = int(round(h * W / H / 32)) * 32
    return w, h


def algorithmic_flops_k0(st, samples, nv=4):
    """FLOPs the reference would spend on the work k_primary covers, from the
    instrumented pass's counts of ray-object tests by outcome."""
    sph = 20 * st["k0_sphere_tests"] + 15 * st["k0_sphere_hits"]
    poly_full = 19 + 27 * nv + 1
    poly = (12 * st["k0_polygon_rej_parallel"] + 13 * st["k0_polygon_rej_behind"]
            + poly_full * st["k0_polygon_full"])
    gen = 50 * samples
    nodes = 51 * st["k0_nodes_sphere"] + 45 * st["k0_nodes_polygon"]
    lights = 28 * st["k0_shadow_rays"] + 36 * st["k0_lights_lit"]
    spawn = 33 * st["k0_reflection_rays"] + 45 * st["k0_transmission_rays"]
    return sph + poly + gen + nodes + lights + spawn


def pmc_for(workload_key, numerics):
    """Counters of k_primary from the committed rocprofv3 --pmc passes (profiles/r02_pmc.json: separate runs,
    FETCH_SIZE doubled as the MI355X guide prescribes), only for the workload and build they were collected on."""
    try:
        t = json.load(open(PMC_FILE))
    except (OSError, ValueError):
        return None
    if t.get("kernels_sha") != kernels_sha():
        return None  # counters of other kernels than the ones timed here: not reported
    return t.get(f"{workload_key}:{numerics}")


def cpu_all_cores(O, sc, W, H, sqrt_spp, depth):
    """the oracle port with its row-parallel OpenMP loop on the host cores this job
    may use (OMP_NUM_THREADS on the GPU box, else the CPU count, at most 16), two full frames."""
    cores = max(1, min(16, int(os.environ.get("OMP_NUM_THREADS", "0")) or (os.cpu_count() or 1)))
    cam = O.prepare_camera(sc.c, W, H, sqrt_spp, depth)
    t0 = time.perf_counter()
    rays = 0
    for _ in range(2):
        _, _, cnt = O.render(sc.c, cam, 0, H, nthreads=cores)
        rays += cnt["primary_rays"] + cnt["shadow_rays"] + cnt["reflection_rays"] + cnt["transmission_rays"]
    dt = time.perf_counter() - t0
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": f"2 full {W}x{H} frames, oracle/rt_oracle.c -O2 -fopenmp, {dt:.2f} s"}


def cpu_baseline(sc, builtin, W, H, sqrt_spp, depth):
    """The reference (oracle/_ref, when it travelled; whole frames of its own scene only) or the oracle port on
    ONE host core - the reference is single-threaded - on a bounded sample of the same workload."""
    from oracle import oracle_py as O
    if builtin and O.have_ref() and W * H * sqrt_spp * sqrt_spp <= 40e6:
        total, frames, rays = 0.0, 0, 0
        while total < 10.0 and frames < 4:  # whole frames until about 10 s of CPU work are done
            t0 = time.perf_counter()
            _, cnt = O.ref_raytrace(sc.c, W, H, sqrt_spp, depth)
            total += time.perf_counter() - t0
            frames += 1
            rays += cnt["ray_create"] + (cnt["ray_tinypush"] - (cnt["ray_create"] - W * H * sqrt_spp * sqrt_spp))
        return {"value": rays / total / 1e6, "unit": "Mrays/s", "cores": 1, "kind": "reference",
                "sample": f"{frames} full {W}x{H} frames, sqrtSpp {sqrt_spp}, depth {depth}, gcc -O2 build of the "
                          f"untouched reference, single-threaded like the reference, {total:.2f} s",
                "all_cores": cpu_all_cores(O, sc, W, H, sqrt_spp, depth)}
    # big frames / synthetic scenes: row slabs from the middle of the frame, until ~10 s of CPU work are done
    cam = O.prepare_camera(sc.c, W, H, sqrt_spp, depth)
    rows, total, rays, used = 1, 0.0, 0, 0
    row0 = H // 2
    while total < 10.0 and row0 + rows <= H and used < 64:
        t0 = time.perf_counter()
        _, _, cnt = O.render(sc.c, cam, row0, rows, nthreads=1)
        total += time.perf_counter() - t0
        rays += cnt["primary_rays"] + cnt["shadow_rays"] + cnt["reflection_rays"] + cnt["transmission_rays"]
        row0 += rows
        used += rows
        rows = min(rows * 2, 16)
    return {"value": rays / total / 1e6, "unit": "Mrays/s", "cores": 1, "kind": "port",
            "sample": f"{used} rows from the middle of the {W}x{H} frame, sqrtSpp {sqrt_spp}, depth {depth}, "
                      f"oracle/rt_oracle.c -O2, {total:.2f} s"}


def small_scene_line(dev, numerics, numerics_name, cfg="syn64", frames=20):
    """A small scene that is NOT the reference's (64 spheres, 4 quads, one light at 1080p x 16, depth 8), one frame at
    a time, with kernels compiled at run time for its counts: k_primary's time and its reference-equivalent FLOP rate."""
    scene_args, W, H, sqrt_spp, depth, _, _ = CONFIGS[cfg]
    sc = rt.Scene.synthetic(*scene_args)
    cam = rt.prepare_camera(sc, W, H, sqrt_spp, depth)
    g = rt.Context(device=dev.index)
    g.upload(sc)
    g.set_numerics(numerics)
    g.set_jit(rt.JIT_NOW)
    frame = torch.zeros((H, W), dtype=torch.int32, device=dev)
    st_alg = g.render_stats(cam, frame.data_ptr(), 0, 0, H, 0).as_dict()
    flops = algorithmic_flops_k0(st_alg, W * H * sqrt_spp * sqrt_spp)
    ms, n, rays, t0 = 0.0, 0, 0, 0.0
    for i in range(6 + frames):
        if i == 6:
            g.set_timing(2)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        g.render_rows(cam, frame.data_ptr(), 0, 0, H, 0)
        st = g.finish()
        if i >= 6:
            ms += st.k0_ms
            n += st.k0_launches
            rays += st.primary_rays + st.shadow_rays + st.reflection_rays + st.transmission_rays
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    state, sec, _ = g.jit_status()
    g.close()
    kms = ms / max(n, 1)
    return {"workload": f"{cfg}: synthetic scene (seed {scene_args[0]:#x}: {scene_args[1]} spheres, {scene_args[2]} quads, "
                        f"{scene_args[3]} light) {W}x{H}, sqrtSpp {sqrt_spp}, depth {depth}; {frames} frames, one at a time",
            "numerics": numerics_name, "ms_per_frame": dt / frames * 1e3, "mrays_per_s": rays / dt / 1e6,
            "k_primary_ms": kms, "algorithmic_flop_per_launch": flops,
            "frac": flops / (kms * 1e-3) / 1e12 / PEAK_FP32_TFLOPS if kms else 0.0,
            "run_time_kernels": {"state": state, "compile_seconds": sec}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)  # the clocks settle over the first ~20 frames
    ap.add_argument("--config", choices=sorted(CONFIGS), default=None,
                    help="workload (module docstring); default c2 on one GPU, c3 on several")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="N > 1: strong (default) = the config's frame, fixed, split over the ranks; weak = the frame "
                         "grows to N x the pixels")
    ap.add_argument("--sqrt-spp", type=int, default=None)
    ap.add_argument("--depth", type=int, default=None)
    ap.add_argument("--inflight", type=int, default=None,
                    help="frames submitted ahead (each with its own render context) on ONE render stream plus ONE tail "
                         "stream: three keep the host two frames ahead of the GPU, so that the next k_primary is queued "
                         "when the render stream falls free (two: 0.228 ms per frame, three: 0.225, four: 0.220)")
    ap.add_argument("--render-priority", choices=["high", "normal"], default="normal",
                    help="HIP stream priority of the shared render stream (k_primary and the wide bounce levels); the tail "
                         "stream has the normal one.  normal (round 3: 0.183 ms per frame, k_primary 0.1095 ms in the pipeline): the "
                         "two share alike; high (0.190 ms, 0.1085 ms): the tail of frame i only fills what k_primary of frame i+1 "
                         "leaves - its kernels then take 2.5 times as long as alone and the tail stream, not the render stream, "
                         "sets the period (round 2, with a heavier tail: high 0.190, normal 0.194)")
    ap.add_argument("--launch-plan", choices=["on", "off"], default="on",
                    help="k_primary's launch plan (rt_hip_ctx_set_launch_plan): each context times the tiles of its first "
                         "frame and launches the blocks of the next ones slowest first; off: raster order, top-down "
                         "(k_primary alone 0.099 against 0.124 ms, 0.106 against 0.125 ms in the pipelined run)")
    ap.add_argument("--shared-stream", action="store_true",
                    help="with more than one rank, or a workload other than config 2: the arrangement of config 2 on one GPU (frames "
                         "on ONE render stream + a tail stream, three in flight) instead of --overlap with six in flight.  A rank's share of a frame spread over N GPUs "
                         "does not fill its GPU, and half of its kernels are bound by latency, whatever their size: frames "
                         "that may overlap freely make up for it (a rank of 8 / 4 / 2 on config 3, rehearsed on one GPU: "
                         "0.118 / 0.192 / 0.336 ms per frame against 0.151 / 0.200 / 0.382)")
    ap.add_argument("--overlap", action="store_true",
                    help="give every in-flight frame its own stream (kernels of neighbouring frames overlap; the "
                         "event-timed k_primary duration then includes the sharing and is not a clean roofline input)")
    ap.add_argument("--tail-stream", choices=["on", "off"], default="on",
                    help="enqueue the latency-bound tail of every frame (sparse deep levels, folds, resolve) on a "
                         "second stream, so that it drains beside the next frame's k_primary (rt_hip_ctx_set_tail_stream)")
    ap.add_argument("--gather-stream", choices=["side", "same"], default="side",
                    help="N > 1: run the all-gather + de-interleave on a second stream behind events, so that they "
                         "overlap the next frame's render (default), or on the render stream (the frames then wait "
                         "for their gathers: step = render + collective)")
    ap.add_argument("--numerics", choices=["fast", "exact"], default=None,
                    help="fast (the default on the reference's scene, c1-c3): the tolerance build of the trace kernels (FMA, "
                         "1-ulp rcp / rsq / sqrt; pixels within the north star's +-1 LSB / 1e-4 there); exact (the default "
                         "on the sphere clouds, c4 / c5s, where the tolerance build is NOT within that bar: rays graze "
                         "hundreds of spheres and one flipped hit is another picture of the pixel): every operation "
                         "rounded like the reference's x86 build (pixels bit-identical to the reference's)")
    ap.add_argument("--graph", choices=["on", "off"], default="off",
                    help="frames as HIP graphs (rt_hip_ctx_set_graph): one graph launch a frame instead of a dozen kernel "
                         "launches, where every frame has a stream of its own. Off: on ROCm 7.2 the graph launch costs "
                         "the host more than the launches it replaces (a rank of eight on config 3: submit 69 us a "
                         "frame against 57, 0.1137 ms a frame either way)")
    ap.add_argument("--jit", choices=["auto", "now", "off"], default=None,
                    help="trace kernels compiled at run time for the scene's counts (rt_hip_ctx_set_jit; small scenes other "
                         "than the reference's, for which the library has such kernels built in).  Default: now for such "
                         "scenes (every context compiles before its first frame, so that the timed region runs one set of "
                         "kernels), else auto")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-one-shot", action="store_true", help="skip the child processes that time the blocking raytrace() call")
    ap.add_argument("--no-generic", action="store_true", help="skip the short extra run with the generic kernels")
    ap.add_argument("--no-small-scene", action="store_true", help="skip the short extra run on the 64-sphere scene")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # rehearsal of one rank of an N-GPU run on a single GPU (no process group: the gather becomes a
    # local copy into this rank's slot); the printed line is marked and is not a scaling result
    fake = int(os.environ.get("RT_BENCH_REHEARSE_WORLD", "0"))
    if fake > 1 and world == 1:
        world, rank, args.gpus = fake, int(os.environ.get("RT_BENCH_REHEARSE_RANK", "0")), fake
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    # RT_BENCH_RCCL_WORLD1=1 (one GPU): the path of many ranks with a world of ONE - process group, barriers, the in-place
    # all-gather, the de-interleave, the all-reduces: every RCCL call of an N-GPU run executes, on the hardware there is
    rccl1 = world == 1 and not fake and bool(os.environ.get("RT_BENCH_RCCL_WORLD1"))
    if (world > 1 and not fake) or rccl1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    cfg = args.config or ("c2" if world == 1 else "c3")
    scene_args, W, H, sqrt_spp, depth, def_steps, def_warm = CONFIGS[cfg]
    sqrt_spp = args.sqrt_spp or sqrt_spp
    depth = depth if args.depth is None else args.depth
    steps = def_steps if args.steps is None else args.steps
    warmup = def_warm if args.warmup is None else args.warmup
    scaling = "weak" if world == 1 else args.scaling  # (one GPU: the per-GPU work is the config's, either way)
    if world > 1 and args.scaling == "weak":
        W, H = grown_frame(W, H, world)
    builtin = scene_args is None
    if args.numerics is None:
        args.numerics = "fast" if builtin else "exact"
    scene = rt.Scene.builtin() if builtin else rt.Scene.synthetic(*scene_args)
    if args.jit is None:
        args.jit = "now" if (not builtin and scene_args[1] <= 128) else "auto"
    cam = rt.prepare_camera(scene, W, H, sqrt_spp, depth)
    plan = RowPlan(H, world)
    numerics = rt.NUMERICS_FAST if args.numerics == "fast" else rt.NUMERICS_EXACT

    # `inflight` frames are submitted ahead: each has its own render context (ray queues, node pool,
    # counters) and frame buffers.  By default they share one HIP stream: the GPU runs the frames back
    # to back while the host collects frame k's counters and submits frame k+2.
    # Frames that may overlap freely (a stream each, six in flight) give the highest throughput on every workload: a
    # rank's share of a frame over N GPUs and the small frames do not fill the GPU, and the 4K frames have long
    # latency-bound tails (config 3 on one GPU 0.775 -> 0.631 ms, config 4 55.9 -> 52.1 ms, config 5's scene 54.8 -> 46.8,
    # config 1 0.056 -> 0.024).  Config 2 on one GPU - the configuration the metric is quoted on - gains 3-5 % that
    # way (0.180-0.184 against 0.190 ms) and keeps ONE high-priority render stream instead, so that the event-timed
    # k_primary of the roofline is the kernel and not the kernel plus its neighbours (0.106 against 0.18-0.21 ms).
    if (world > 1 or cfg != "c2") and not args.shared_stream:
        args.overlap = True
    # (the sphere clouds: two frames in flight - a 4K frame of config 4 in one batch holds 42 GB of hits, six of them
    # would fill the GPU, and two already hide each other's sparse levels: 44.3 ms a frame against 44.7 with six;
    # RT_HIP_QUEUE_CAP=33554432 renders it in twelve batches out of 5.9 GB, at 48.1 ms)
    inflight = max(1, args.inflight if args.inflight is not None else ((6 if builtin else 2) if args.overlap else 3))
    tail = torch.cuda.Stream(device=dev) if args.tail_stream == "on" and not args.overlap else None
    ctxs = []
    for _ in range(inflight):
        # queue capacity sized by the first render of this rank's share (development: RT_BENCH_QUEUE_CAP fixes it)
        c = rt.Context(device=local_rank, queue_cap=int(os.environ.get("RT_BENCH_QUEUE_CAP", "0")))
        c.upload(scene)
        c.set_numerics(numerics)
        c.set_launch_plan(args.launch_plan == "on")
        c.set_graph(args.graph == "on")
        if args.jit != "auto":  # (auto: the library's policy - compile beside the frames after the eighth of a scene)
            c.set_jit({"off": rt.JIT_OFF, "now": rt.JIT_NOW}[args.jit])
        if tail is not None:
            c.set_tail_stream(tail.cuda_stream)
        ctxs.append(c)

    def render(i, k, target, stream):
        if world == 1 and not rccl1:
            ctxs[k].render_rows(cam, target.data_ptr(), 0, 0, H, stream.cuda_stream)
        else:
            ctxs[k].render_interleaved(cam, target.data_ptr(), 0, rank, world, stream.cuda_stream)

    def finish(i, k):
        return ctxs[k].finish()  # waits for the render, checks the queues, reads the counters

    pipe = FramePipeline(plan, W, torch.int32, dev, rank, render, finish, inflight=inflight,  # 0xFFRRGGBB bit patterns
                         gather_stream=args.gather_stream, done_stream=tail, collective=not fake,
                         render_streams=[torch.cuda.Stream(device=dev) for _ in range(inflight)] if args.overlap else None,
                         render_priority=args.render_priority, force_collective=rccl1)

    # instrumented pass (outside the timed region, exact kernels): algorithmic work of this rank's share
    full = torch.zeros((H, W), dtype=torch.int32, device=dev)
    if world == 1:
        st_alg = ctxs[0].render_stats(cam, full.data_ptr(), 0, 0, H, 0).as_dict()
    else:
        # the instrumented kernels render contiguous rows: add up this rank's own 8-row blocks (a slab
        # would not do - the top slab is all sky while the timed cyclic blocks are a fair sample)
        st_alg = {}
        for y, n, _ in plan.rows_of(rank):
            d = ctxs[0].render_stats(cam, full.data_ptr(), 0, y, n, 0).as_dict()
            for key, v in d.items():
                if isinstance(v, (int, float)):
                    st_alg[key] = st_alg.get(key, 0) + v
    del full
    my_rows = sum(n for _, n, _ in plan.rows_of(rank)) if world > 1 else H

    # set-up, before the W warm-up steps: every context renders the frame twice - a context sizes its queues from
    # its first frame, times its tiles for k_primary's launch plan, and plans the sparse levels of the tree
    # (k_subtree, k_fold_deep) from the second on
    # ... and goes on, for at most 30 ms, until the GPU's clocks have settled (the first ~20 frames of a process run
    # 5 % slow: a short --warmup would otherwise be timed on a GPU that is still ramping up)
    t_setup = time.perf_counter()
    i = 0
    while i < 2 * inflight or (i < 64 and time.perf_counter() - t_setup < 0.03):
        pipe.submit(i)
        pipe.collect(i, sizing=i < inflight)  # (a context's first frame: rendered in batches, some of them twice)
        i += 1
    for i in range(warmup):
        pipe.submit(i)
        pipe.collect(i)
    # k_primary's events: on the shared render stream the kernel's own start and end (what rocprofv3 reports); with a
    # stream per frame - small frames, ranks of a split frame: the host is what bounds those - the cheaper pair around
    # the dispatch (there the kernel shares the GPU with its neighbours' kernels and its time says little anyway)
    timing_mode = 0 if os.environ.get("RT_BENCH_NO_K0_EVENTS") else (1 if args.overlap else 2)
    for c in ctxs:
        c.set_timing(timing_mode)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    acc = {"k0_ms": 0.0, "k0_n": 0, "rays": 0, "shadow": 0, "culled": 0, "mib": 0, "batches": 0, "retries": 0}
    done_at = []  # when each frame of the timed region was collected

    def tally(st):
        done_at.append(time.perf_counter())
        acc["mib"] = max(acc["mib"], st.work_mib)
        acc["batches"] += st.batches
        acc["retries"] += st.overflow_retries
        acc["k0_ms"] += st.k0_ms
        acc["k0_n"] += st.k0_launches
        acc["rays"] += st.primary_rays + st.shadow_rays + st.reflection_rays + st.transmission_rays
        acc["shadow"] += st.shadow_rays
        acc["culled"] += st.primary_rays_culled

    host = [0.0, 0.0]  # (development, RT_BENCH_HOST_TIMES=1: seconds the host spent in collect / in submit)
    for i in range(steps):
        ta = time.perf_counter()
        if i >= inflight:
            tally(pipe.collect(i - inflight))
        tb = time.perf_counter()
        pipe.submit(i)
        host[0] += tb - ta
        host[1] += time.perf_counter() - tb
    for i in range(max(0, steps - inflight), steps):
        tally(pipe.collect(i))
    if os.environ.get("RT_BENCH_HOST_TIMES"):
        print("host per frame: collect %.1f us, submit %.1f us" % (host[0] / steps * 1e6, host[1] / steps * 1e6), file=sys.stderr)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    for c in ctxs:
        c.set_timing(False)

    tot = torch.tensor([dt, float(acc["rays"]), float(acc["shadow"]), float(acc["culled"])], dtype=torch.float64, device=dev)
    if dist is not None:
        tmax = tot.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        dt_max = float(tmax[0])
    else:
        dt_max = dt
    rays_all, shadow_all, culled_all = float(tot[1]), float(tot[2]), float(tot[3])

    if rank == 0 or fake:  # (a rehearsal prints whichever rank it plays)
        samples = W * my_rows * sqrt_spp * sqrt_spp
        flops_k0 = algorithmic_flops_k0(st_alg, samples)
        k0_avg_ms = acc["k0_ms"] / max(acc["k0_n"], 1)
        launches_per_step = max(acc["k0_n"] / max(steps, 1), 1)
        achieved = flops_k0 / launches_per_step / (k0_avg_ms * 1e-3) / 1e12 if k0_avg_ms > 0 else 0.0
        scene_name = ("builtin Whitted scene" if builtin else
                      f"synthetic scene (seed {scene_args[0]:#x}: {scene_args[1]} spheres, {scene_args[2]} quads, "
                      f"{scene_args[3]} lights)")
        split = "" if world == 1 else (f", cyclic 8-row blocks over {world} GPUs + RCCL all-gather"
                                       + (", the frame fixed" if scaling == "strong" else ", the frame grown with N"))
        grid_scene = (not builtin) and scene_args[1] > 128
        out = {
            "metric": "Mrays/s (primary+secondary) Whitted 1080p 16spp at 1/8 GPUs; % FP32 VALU peak",
            "value": rays_all / dt_max / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": steps,
            "warmup": warmup,
            "ms_per_step": dt_max / steps * 1e3,
            # spread of the timed region: intervals between consecutive frames' completions on rank 0
            "step_ms": (lambda d: {"p50": float(np.percentile(d, 50)), "p95": float(np.percentile(d, 95)),
                                   "max": float(d.max())} if len(d) else None)(np.diff(np.array(done_at)) * 1e3),
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": ("synthetic (the reference's hard-coded Whitted scene, scene.c:86-258)" if builtin else
                     "synthetic (seeded random scene, rt_scene_synthetic)")
                    + (f"; REHEARSAL of rank {rank} of {world} on one GPU, no collective" if fake else "")
                    + ("; REHEARSAL of the multi-rank path with a world of one: every RCCL call runs" if rccl1 else ""),
            "config": {"workload": f"{cfg}: {scene_name} {W}x{H}, sqrtSpp {sqrt_spp} ({sqrt_spp * sqrt_spp} spp), "
                                   f"depth {depth}{split}",
                       "rays_per_step": rays_all / steps,
                       # the sub-total without shadow rays, and its rate
                       "rays_per_step_without_shadow": (rays_all - shadow_all) / steps,
                       "mrays_per_s_without_shadow": (rays_all - shadow_all) / dt_max / 1e6,
                       # rays really generated and traced: the primary rays of tiles shown to see nothing are not
                       "rays_traced_per_step": (rays_all - culled_all) / steps,
                       "mrays_traced_per_s": (rays_all - culled_all) / dt_max / 1e6,
                       "numerics": args.numerics,
                       "frames_in_flight": inflight,
                       "batches_per_frame": acc["batches"] / max(steps, 1), "overflow_retries": acc["retries"],
                       "device_mib_per_context": acc["mib"],
                       "device_mib": acc["mib"] * inflight + (W * my_rows * 4 * max(2, inflight) >> 20),
                       "streams": "one per frame" if args.overlap else ("shared + tail stream" if tail is not None else "shared"),
                       "render_stream_priority": args.render_priority,
                       "k_primary_launch_plan": args.launch_plan,
                       "parallelism": "1 GPU" if world == 1 else f"cyclic 8-row blocks over {world} GPUs"},
            "roofline": {"bound": "fp32_valu", "kernel": "k_primary", "achieved": achieved,
                         "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_FP32_TFLOPS,
                         "definition": "reference-equivalent FLOPs of the work k_primary covers (SURVEY 8d constants x the "
                                       "instrumented pass's test counts" + (", i.e. the tests the sphere grid makes, not the "
                                       "reference's loop over every sphere" if grid_scene else "") + ") / launch time: the "
                                       "kernel skips tiles that see nothing and decides most polygon tests by a cheap screen, "
                                       "so this is not a utilisation - executed_frac is",
                         "traffic": None, "algorithmic_flop_per_launch": flops_k0 / launches_per_step,
                         "avg_launch_ms": k0_avg_ms},
        }
        if not grid_scene and k0_avg_ms > 0:
            # the same rate without the primary rays the kernel never generated (tiles shown to see nothing):
            # each priced as the reference's miss - ray generation, every sphere rejected, every polygon behind
            n_sph = 2 if builtin else scene_args[1]
            n_pol = 1 if builtin else scene_args[2]
            culled_flops = culled_all / max(steps, 1) / max(world, 1) * (50 + 20 * n_sph + 13 * n_pol)
            tr = (flops_k0 - culled_flops) / launches_per_step / (k0_avg_ms * 1e-3) / 1e12
            out["roofline"]["frac_traced"] = tr / PEAK_FP32_TFLOPS
        if grid_scene:
            out["roofline"]["kernel"] = "k_hits + k_trace(level 0): the primary rays and the rays of their hits"
        pmc = pmc_for(cfg, args.numerics) if world == 1 and (W, H, sqrt_spp, depth) == CONFIGS[cfg][1:5] else None
        if pmc:
            r = out["roofline"]
            r["traffic"] = pmc.get("hbm_bytes_per_launch")
            r["traffic_unit"] = "bytes per launch (PMC, profiles/r03_pmc.json, kernels_sha %s)" % kernels_sha()
            lane_ops = pmc.get("valu_lane_ops_per_launch")
            if lane_ops and k0_avg_ms > 0:
                # what the SIMDs can issue: one vector instruction (64 lanes) per 2 cycles per SIMD = peak / 2 lane-ops/s
                r["executed_valu_lane_ops_per_launch"] = lane_ops
                r["executed_frac"] = lane_ops / (k0_avg_ms * 1e-3) / (PEAK_FP32_TFLOPS * 1e12 / 2)
            fp = pmc.get("fp_lane_flops_per_launch")
            if fp and k0_avg_ms > 0:
                # floating-point work really executed: 2 x FMA + MUL + ADD + transcendental instructions x active lanes
                r["fp_executed_frac"] = fp / (k0_avg_ms * 1e-3) / (PEAK_FP32_TFLOPS * 1e12)
            for key in ("valu_busy", "lds_conflict_rate", "waves_per_simd"):
                if key in pmc:
                    r[key] = pmc[key]
            if r["traffic"] and k0_avg_ms > 0:  # the secondary bound, HBM
                gbs = r["traffic"] / (k0_avg_ms * 1e-3) / 1e9
                r["secondary"] = {"bound": "hbm", "achieved": gbs, "peak": 8000.0, "unit": "GB/s", "frac": gbs / 8000.0}
        if world == 1 and not args.no_generic:
            # 30 frames, one at a time: k_primary alone on the GPU (in the timed region above it shares the GPU with
            # the previous frame's tail, which the tail stream lets in: throughput wins, the kernel's own time loses)
            def one_at_a_time(g, warm=10):
                g.set_timing(2)
                frame = pipe.gathers[0].frame
                ms, n, tg = 0.0, 0, 0.0
                for i in range(warm + 30):
                    if i == warm:
                        torch.cuda.synchronize()
                        tg = time.perf_counter()
                    g.render_rows(cam, frame.data_ptr(), 0, 0, H, pipe.streams[0].cuda_stream)
                    st = g.finish()
                    if i >= warm:
                        ms += st.k0_ms
                        n += st.k0_launches
                torch.cuda.synchronize()
                tg = (time.perf_counter() - tg) / 30
                g.set_timing(False)
                kms = ms / max(n, 1)
                return {"k_primary_ms": kms,
                        "frac": flops_k0 / launches_per_step / (kms * 1e-3) / 1e12 / PEAK_FP32_TFLOPS if kms else 0.0,
                        "ms_per_step_one_frame_at_a_time": tg * 1e3}
            g = ctxs[0]
            g.set_tail_stream(0)
            out["roofline"]["one_frame_at_a_time"] = dict(
                one_at_a_time(g), note="30 frames, each alone on the GPU, one stream: the kernel without a neighbour")
            if builtin:
                # the same frames with the kernels any OTHER scene gets: first the generic ones (counts are run-time
                # values), which such a scene's first frames take, then the ones compiled at run time for the scene's
                # counts (rt_jit.cpp), which its later frames take.  A context of its own each: a launch plan is made of
                # the times the tiles took with the kernels that will follow it.
                def other_kernels(jit_mode):
                    h = rt.Context(device=local_rank)
                    h.upload(scene)
                    h.set_numerics(numerics)
                    h.set_launch_plan(args.launch_plan == "on")
                    h.set_specialization(False)
                    h.set_jit(jit_mode)
                    r = one_at_a_time(h, warm=60)  # (the GPU sat idle while the context was made and the kernels compiled: its clocks take ~40 frames to come back)
                    r["compile_seconds"] = h.jit_status()[1] if jit_mode != rt.JIT_OFF else None
                    h.close()
                    return r
                out["generic_kernels"] = dict(
                    other_kernels(rt.JIT_OFF), note="30 frames, one at a time, with the library's kernels for any scene shape "
                                                    "(rt_hip_ctx_set_specialization(ctx, 0), rt_hip_ctx_set_jit(ctx, RT_JIT_OFF))")
                out["run_time_kernels"] = dict(
                    other_kernels(rt.JIT_NOW_ANY), note="30 frames, one at a time, with kernels compiled at run time for this scene's "
                                                        "counts as for any other small scene (rt_hip_ctx_set_jit(ctx, RT_JIT_NOW_ANY): "
                                                        "the library's own kernels for the reference's shape are not used)")
            elif not grid_scene and args.jit != "off":
                st_, sec_, _ = g.jit_status()
                out["run_time_kernels"] = {"state": st_, "compile_seconds": sec_,
                                           "note": "the timed region ran kernels compiled at run time for this scene's counts"}
                g.set_jit(rt.JIT_OFF)
                out["generic_kernels"] = dict(
                    one_at_a_time(g), note="30 frames, one at a time, with the library's kernels for any scene shape")
                g.set_jit(rt.JIT_NOW)
            if builtin and cfg == "c2" and not args.no_small_scene:
                out["small_scene"] = small_scene_line(dev, numerics, args.numerics)
        if world == 1 and builtin and not args.no_one_shot:
            # the call the reference's main() makes - raytrace() with the pixels in the caller's pageable buffer - timed in
            # child processes: a cold process from exec to exit, and the same process calling it frame after frame
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import one_shot
            try:
                out["one_shot"] = one_shot.measure(W, H, sqrt_spp, depth)
            except Exception as e:  # noqa: BLE001 - the bench line must still print
                out["one_shot"] = {"error": str(e)[-300:]}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(scene, builtin, W, H, sqrt_spp, depth)
        print(json.dumps(out), flush=True)
    for c in ctxs:
        c.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
