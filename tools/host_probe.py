"""development: where the host's time goes in the pipelined frame loop (submit vs collect), config 2.
python tools/host_probe.py [inflight] [fast|exact]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as rt

inflight = int(sys.argv[1]) if len(sys.argv) > 1 else 2
mode = rt.NUMERICS_FAST if (sys.argv[2:] or ["fast"])[0] == "fast" else rt.NUMERICS_EXACT
sc = rt.Scene.builtin()
cam = rt.prepare_camera(sc, 1920, 1080, 4, 8)
main, side = torch.cuda.Stream(), torch.cuda.Stream()
ctxs, bufs = [], []
for _ in range(inflight):
    c = rt.Context(); c.upload(sc); c.set_numerics(mode); c.set_tail_stream(side.cuda_stream)
    ctxs.append(c); bufs.append(torch.zeros((1080, 1920), dtype=torch.int32, device="cuda"))
def run(n):
    ts = tc = 0.0
    for i in range(inflight):
        ctxs[i].render_rows(cam, bufs[i].data_ptr(), stream=main.cuda_stream)
    t0 = time.perf_counter()
    for i in range(n):
        k = i % inflight
        a = time.perf_counter(); ctxs[k].finish(); b = time.perf_counter()
        ctxs[k].render_rows(cam, bufs[k].data_ptr(), stream=main.cuda_stream); c = time.perf_counter()
        tc += b - a; ts += c - b
    dt = time.perf_counter() - t0
    for c in ctxs: c.finish()
    return dt / n * 1e3, ts / n * 1e3, tc / n * 1e3
run(50)
print("inflight %d: period %.4f ms, submit %.4f ms, collect %.4f ms" % ((inflight,) + run(400)))
