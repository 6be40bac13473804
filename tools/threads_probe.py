"""development: eight host threads, a context each, rendering different scenes at the same time (ctypes drops the GIL
in the calls; uploads, grid builds, run-time compiles and renders overlap) - every frame against the oracle, bit for bit."""
import sys, threading; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch, raytracer_amd as rt
from oracle import oracle_py as O
O.set_rounded_pow(True)
torch.cuda.init()
jobs = []
rng = np.random.default_rng(5)
for t in range(8):
    mine = []
    for k in range(12):
        n_sph = int(rng.choice([3, 30, 200, 2500]))
        sc = rt.Scene.synthetic(int(rng.integers(1, 2**31)), n_sph, int(rng.integers(1, 4)), int(rng.choice([1, 2, 8])), int(rng.integers(0, 2)))
        mine.append((sc, int(rng.integers(16, 64)), int(rng.integers(16, 48)), int(rng.integers(1, 3)), int(rng.integers(0, 7))))
    jobs.append(mine)
out = [[] for _ in jobs]
def work(t):
    st = torch.cuda.Stream()
    for sc, W, H, s, d in jobs[t]:
        cam = rt.prepare_camera(sc, W, H, s, d)
        ctx = rt.Context(); ctx.upload(sc)
        argb = torch.zeros((H, W), dtype=torch.int32, device="cuda"); rad = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
        st.wait_stream(torch.cuda.current_stream())  # (the fills above ran on torch's current stream, not on st)
        for _ in range(3):   # (the third frame of a small scene may run kernels compiled for it meanwhile)
            ctx.render_rows(cam, argb.data_ptr(), rad.data_ptr(), 0, H, st.cuda_stream); ctx.finish()
        st.synchronize()
        out[t].append((argb.cpu().numpy().view(np.uint32), rad.cpu().numpy(), cam))
        ctx.close()
ths = [threading.Thread(target=work, args=(t,)) for t in range(len(jobs))]
[th.start() for th in ths]; [th.join() for th in ths]
bad = 0
for t in range(len(jobs)):
    assert len(out[t]) == len(jobs[t]), "a thread died"
    for (sc, W, H, s, d), (a, r, cam) in zip(jobs[t], out[t]):
        a_o, r_o, _ = O.render(sc.c, cam, nthreads=16)
        ok = np.array_equal(r.view(np.uint32), r_o.view(np.uint32)) and np.array_equal(a, a_o)
        bad += not ok
print(sum(len(j) for j in jobs), "frames,", bad, "bad")
sys.exit(1 if bad else 0)
