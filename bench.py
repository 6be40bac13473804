#!/usr/bin/env python3
"""bench.py - throughput of the HIP trace-and-shade path on BASELINE.json's headline workload.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one full render of the frame: every rank renders its row share with the scene and the
frame buffer resident in HBM, then (N > 1) the frame-buffer slices are collected with one RCCL
all-gather.  N = 1 renders BASELINE config 2 (builtin Whitted scene, 1920x1080, 4x4 supersampling,
depth 8).  N > 1 keeps the per-GPU work fixed (weak scaling): the frame grows to N x the pixels at
16:9 (N = 4 is BASELINE config 3, 3840x2160).

Rank 0 prints ONE JSON line.  `value` = rays (primary + shadow + reflection + transmission, counted
by the kernels) of all ranks / max-over-ranks time.  `roofline` prices the dominant kernel
(k_primary) with the algorithmic FLOP constants of SURVEY.md section 8d against the FP32 vector peak;
`cpu_baseline` times the reference (or the oracle port) on one host core.
"""
import argparse
import ctypes as C
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import raytracer_amd as rt  # noqa: E402
from raytracer_amd.dist import FrameGather, RowPlan  # noqa: E402

PEAK_FP32_TFLOPS = 157.3  # MI355X vector FP32 peak, MI355X_MICROARCH.md "Chip-level parameters"


def frame_for(n_gpus):
    """Frame size with n_gpus x the pixels of 1920x1080 at 16:9; height a multiple of 8*n_gpus."""
    if n_gpus == 1:
        return 1920, 1080
    if n_gpus == 4:
        return 3840, 2160  # BASELINE config 3 (2160 = 4 * 540)
    h = int(round(1080 * math.sqrt(n_gpus) / (8 * n_gpus))) * 8 * n_gpus
    w = int(round(h * 16 / 9 / 32)) * 32
    return w, h


def algorithmic_flops_k0(st, samples, n_lights, nv=4):
    """FLOPs the reference would spend on the work k_primary covers."""
    sph = 20 * st["k0_sphere_tests"] + 15 * st["k0_sphere_hits"]
    poly_full = 19 + 27 * nv + 1
    poly = (12 * st["k0_polygon_rej_parallel"] + 13 * st["k0_polygon_rej_behind"]
            + poly_full * st["k0_polygon_full"])
    gen = 50 * samples
    nodes = 51 * st["k0_nodes_sphere"] + 45 * st["k0_nodes_polygon"]
    lights = 28 * st["k0_shadow_rays"] + 36 * st["k0_lights_lit"]
    spawn = 33 * st["k0_reflection_rays"] + 45 * st["k0_transmission_rays"]
    return sph + poly + gen + nodes + lights + spawn


def measured_traffic(W, H, sqrt_spp, depth):
    """HBM bytes per k_primary launch from the committed PMC passes (profiles/r01_traffic.json:
    FETCH_SIZE / WRITE_SIZE collected in separate rocprofv3 runs, corrected as the MI355X guide
    prescribes).  Only for the workload they were collected on; otherwise None."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))
    except OSError:
        return None
    w = t["workload"]
    if (w["W"], w["H"], w["sqrt_spp"], w["depth"]) != (W, H, sqrt_spp, depth):
        return None
    return t["k_primary"]["bytes_per_launch"]


def executed_lane_ops(W, H, sqrt_spp, depth):
    """SURVEY 8(d) asks for "executed" next to "algorithmic": vector lane operations k_primary really issues
    per launch (SQ_INSTS_VALU x 64 from the committed PMC pass), same workload condition as above."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))
    except OSError:
        return None
    w = t["workload"]
    if (w["W"], w["H"], w["sqrt_spp"], w["depth"]) != (W, H, sqrt_spp, depth):
        return None
    return t["k_primary"].get("executed_valu_lane_ops_per_launch")


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


def cpu_baseline(W, H, sqrt_spp, depth, budget_rows):
    """Reference (oracle/_ref, when it travelled) or the oracle port on ONE host core, on a row
    slab of the same frame.  Returns the JSON object."""
    from oracle import oracle_py as O
    sc = rt.Scene.builtin()
    row0 = (H - budget_rows) // 2
    if O.have_ref():
        # the untouched reference renders whole frames only: time a frame with H*budget/H rows is not
        # possible, so render the full frame at reduced supersampling-invariant size instead
        # whole frames until about 10 s of CPU work are done (bounded: at most 4)
        total, frames, rays = 0.0, 0, 0
        while total < 10.0 and frames < 4:
            t0 = time.perf_counter()
            _, cnt = O.ref_raytrace(sc.c, W, H, sqrt_spp, depth)
            total += time.perf_counter() - t0
            frames += 1
            rays += cnt["ray_create"] + (cnt["ray_tinypush"] - (cnt["ray_create"] - W * H * sqrt_spp * sqrt_spp))
        return {"value": rays / total / 1e6, "unit": "Mrays/s", "cores": 1, "kind": "reference",
                "sample": f"{frames} full {W}x{H} frames, sqrtSpp {sqrt_spp}, depth {depth}, gcc -O2 build of the "
                          f"untouched reference, single-threaded like the reference, {total:.2f} s",
                "all_cores": cpu_all_cores(O, sc, W, H, sqrt_spp, depth)}
    cam = O.prepare_camera(sc.c, W, H, sqrt_spp, depth)
    t0 = time.perf_counter()
    _, _, cnt = O.render(sc.c, cam, row0, budget_rows, nthreads=1)
    dt = time.perf_counter() - t0
    rays = (cnt["primary_rays"] + cnt["shadow_rays"] + cnt["reflection_rays"] + cnt["transmission_rays"])
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": 1, "kind": "port",
            "sample": f"rows [{row0},{row0 + budget_rows}) of the {W}x{H} frame, sqrtSpp {sqrt_spp}, "
                      f"depth {depth}, oracle/rt_oracle.c -O2, {dt:.2f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=30)  # the clocks settle over the first ~20 frames
    ap.add_argument("--sqrt-spp", type=int, default=4)
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--inflight", type=int, default=2,
                    help="frames submitted ahead (each with its own render context); they share ONE HIP stream, "
                         "so kernels of different frames never overlap - only the host's submit/collect latency "
                         "between frames is hidden")
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
    ap.add_argument("--numerics", choices=["fast", "exact"], default="fast",
                    help="fast (default): the tolerance build of the trace kernels (FMA, 1-ulp rcp / rsq / sqrt; pixels "
                         "within the north star's +-1 LSB / 1e-4); exact: every operation rounded like the reference's "
                         "x86 build (pixels bit-identical to the reference's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
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
    if world > 1 and not fake:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    W, H = frame_for(world)
    sqrt_spp, depth = args.sqrt_spp, args.depth
    scene = rt.Scene.builtin()
    cam = rt.prepare_camera(scene, W, H, sqrt_spp, depth)
    plan = RowPlan(H, world)

    # `inflight` frames are submitted ahead: each has its own render context (ray queues, node pool,
    # counters) and frame buffers.  By default they share one HIP stream: the GPU runs the frames back
    # to back while the host collects frame k's counters and submits frame k+2.
    inflight = max(1, args.inflight)
    ctxs, streams, gathers = [], [], []
    shared = torch.cuda.Stream(device=dev)
    tail = torch.cuda.Stream(device=dev) if args.tail_stream == "on" and not args.overlap else None
    for _ in range(inflight):
        c = rt.Context(device=local_rank)
        c.upload(scene)
        c.set_numerics(rt.NUMERICS_FAST if args.numerics == "fast" else rt.NUMERICS_EXACT)
        if tail is not None:
            c.set_tail_stream(tail.cuda_stream)
        ctxs.append(c)
        streams.append(torch.cuda.Stream(device=dev) if args.overlap else shared)
        gathers.append(FrameGather(plan, W, torch.int32, dev, rank=rank))  # 0xFFRRGGBB bit patterns

    # N > 1: the in-place all-gather of frame k and its de-interleave copy (the library's streaming
    # kernel) run on a second stream behind an event, overlapping the render of frame k+1
    # (--gather-stream same keeps them on the render stream).
    comm = (torch.cuda.Stream(device=dev) if args.gather_stream == "side" else shared) if world > 1 else None
    rendered = [torch.cuda.Event() for _ in range(inflight)]
    gathered = [torch.cuda.Event() for _ in range(inflight)]

    def submit(i):
        k = i % inflight
        st, fg, ctx = streams[k], gathers[k], ctxs[k]
        with torch.cuda.stream(st):
            if world == 1:
                ctx.render_rows(cam, fg.frame.data_ptr(), 0, 0, H, st.cuda_stream)
                return
            if i >= inflight:
                st.wait_event(gathered[k])  # the previous gather out of this frame's buffers is done
            ctx.render_interleaved(cam, fg.local.data_ptr(), 0, rank, world, st.cuda_stream)
            rendered[k].record(tail if tail is not None else st)  # (the frame is complete on the tail stream)
        with torch.cuda.stream(comm):
            comm.wait_event(rendered[k])
            fg.gather(collective=not fake)  # in-place RCCL all-gather of the local slices + de-interleave
            gathered[k].record(comm)

    def collect(i):
        k = i % inflight
        st = ctxs[k].finish()  # waits for the render, checks the queues, reads the counters
        return st

    # instrumented pass (outside the timed region): algorithmic work of this rank's share
    tmp = torch.zeros((plan.local_rows, W), dtype=torch.int32, device=dev)
    if world == 1:
        st_alg = ctxs[0].render_stats(cam, gathers[0].frame.data_ptr(), 0, 0, H, 0).as_dict()
    else:
        # the instrumented kernels render contiguous rows: add up this rank's own 8-row blocks (a slab
        # would not do - the top slab is all sky while the timed cyclic blocks are a fair sample)
        full = torch.zeros((H, W), dtype=torch.int32, device=dev)
        st_alg = {}
        for y, n, _ in plan.rows_of(rank):
            d = ctxs[0].render_stats(cam, full.data_ptr(), 0, y, n, 0).as_dict()
            for k, v in d.items():
                if isinstance(v, (int, float)):
                    st_alg[k] = st_alg.get(k, 0) + v
        del full
    del tmp
    my_rows = sum(n for _, n, _ in plan.rows_of(rank)) if world > 1 else H

    for i in range(args.warmup):
        submit(i)
        collect(i)
    for c in ctxs:
        c.set_timing(True)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    k0_ms, k0_n, rays_rank, shadow_rank = 0.0, 0, 0, 0
    for i in range(args.steps):
        if i >= inflight:
            st = collect(i - inflight)
            k0_ms += st.k0_ms
            k0_n += st.k0_launches
            rays_rank += (st.primary_rays + st.shadow_rays + st.reflection_rays + st.transmission_rays)
            shadow_rank += st.shadow_rays
        submit(i)
    for i in range(max(0, args.steps - inflight), args.steps):
        st = collect(i)
        k0_ms += st.k0_ms
        k0_n += st.k0_launches
        rays_rank += (st.primary_rays + st.shadow_rays + st.reflection_rays + st.transmission_rays)
        shadow_rank += st.shadow_rays
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    for c in ctxs:
        c.set_timing(False)

    tot = torch.tensor([dt, float(rays_rank), float(shadow_rank)], dtype=torch.float64, device=dev)
    if dist is not None:
        tmax = tot.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        dt_max, rays_all, shadow_all = float(tmax[0]), float(tot[1]), float(tot[2])
    else:
        dt_max, rays_all, shadow_all = dt, float(rays_rank), float(shadow_rank)

    if rank == 0 or fake:  # (a rehearsal prints whichever rank it plays)
        samples = W * my_rows * sqrt_spp * sqrt_spp
        flops_k0 = algorithmic_flops_k0(st_alg, samples, scene.c.nLights)
        k0_avg_ms = k0_ms / max(k0_n, 1)
        launches_per_step = k0_n / max(args.steps, 1)
        achieved = flops_k0 / max(launches_per_step, 1) / (k0_avg_ms * 1e-3) / 1e12 if k0_avg_ms > 0 else 0.0
        out = {
            "metric": "Mrays/s (primary+secondary) Whitted 1080p 16spp at 1/8 GPUs; % FP32 VALU peak",
            "value": rays_all / dt_max / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt_max / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic (the reference's hard-coded Whitted scene, scene.c:86-258)"
                    + (f"; REHEARSAL of rank {rank} of {world} on one GPU, no collective" if fake else ""),
            "config": {"workload": f"builtin Whitted scene {W}x{H}, sqrtSpp {sqrt_spp} ({sqrt_spp * sqrt_spp} spp), "
                                   f"depth {depth}" + ("" if world == 1 else f", cyclic 8-row blocks over {world} GPUs + RCCL all-gather"),
                       "rays_per_step": rays_all / args.steps,
                       # the sub-total without shadow rays, and its rate
                       "rays_per_step_without_shadow": (rays_all - shadow_all) / args.steps,
                       "mrays_per_s_without_shadow": (rays_all - shadow_all) / dt_max / 1e6,
                       "numerics": args.numerics,
                       "frames_in_flight": inflight, "streams": "one per frame" if args.overlap else ("shared + tail stream" if tail is not None else "shared"),
                       "parallelism": "1 GPU" if world == 1 else f"cyclic 8-row blocks over {world} GPUs"},
            "roofline": {"bound": "fp32_valu", "kernel": "k_primary", "achieved": achieved,
                         "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_FP32_TFLOPS,
                         "traffic": measured_traffic(W, H, sqrt_spp, depth) if world == 1 else None,
                         "traffic_unit": "bytes per launch (PMC, profiles/r01_traffic.json)",
                         "algorithmic_flop_per_launch": flops_k0 / max(launches_per_step, 1),
                         "executed_valu_lane_ops_per_launch": executed_lane_ops(W, H, sqrt_spp, depth) if world == 1 else None,
                         "avg_launch_ms": k0_avg_ms},
        }
        # the secondary bound, HBM, from the measured bytes of the same kernel
        tr = out["roofline"]["traffic"]
        if tr and k0_avg_ms > 0:
            gbs = tr / (k0_avg_ms * 1e-3) / 1e9
            out["roofline"]["secondary"] = {"bound": "hbm", "achieved": gbs, "peak": 8000.0, "unit": "GB/s",
                                            "frac": gbs / 8000.0}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(W, H, sqrt_spp, depth, budget_rows=H)
        print(json.dumps(out), flush=True)
    for c in ctxs:
        c.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
