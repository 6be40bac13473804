"""development: frame time of config 2 on streams restricted to half of the CUs, for several ways of
choosing the half.  If CUs that share an instruction / scalar cache contend for it, a mask that takes
one CU of every pair is faster than one that takes both CUs of every other pair.
usage: python tools/cumask_probe.py"""
import ctypes, sys, time; sys.path.insert(0, '.')
import torch, raytracer_amd as rt

torch.zeros(1, device='cuda')
path = next(l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l)
hip = ctypes.CDLL(path)
hip.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32,
                                             ctypes.POINTER(ctypes.c_uint32)]
hip.hipStreamSynchronize.argtypes = [ctypes.c_void_p]
hip.hipStreamDestroy.argtypes = [ctypes.c_void_p]
NCU = torch.cuda.get_device_properties(0).multi_processor_count
print("CUs", NCU, flush=True)


def stream_for(pred):
    words = (ctypes.c_uint32 * ((NCU + 31) // 32))()
    n = 0
    for i in range(NCU):
        if pred(i):
            words[i // 32] |= 1 << (i % 32); n += 1
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), len(words), words)
    if rc != 0:
        print("mask refused, rc", rc, flush=True)
        return None, n
    return s, n


sc = rt.Scene.builtin(); W, H, s_, d = 1920, 1080, 4, 8
cam = rt.prepare_camera(sc, W, H, s_, d)
ctx = rt.Context(); ctx.upload(sc)
fb = torch.zeros((H, W), dtype=torch.int32, device='cuda')
masks = [("all", lambda i: True), ("first 128", lambda i: i < 128), ("first 255", lambda i: i < 255)]
for k in (1, 2, 4, 8, 16, 32, 64, 128):
    masks.append((f"(i/{k})%2==0", (lambda k: lambda i: (i // k) % 2 == 0)(k)))
masks.append(("i%4==0 (64 CUs)", lambda i: i % 4 == 0))
masks.append(("(i/2)%4==0 (64 CUs)", lambda i: (i // 2) % 4 == 0))
masks.append(("(i/32)%4==0 (64 CUs)", lambda i: (i // 32) % 4 == 0))
want = sys.argv[1:]  # one or two mask indices per process: every masked stream takes a hardware queue
for name, pred in [masks[int(w)] for w in want] if want else masks[:3]:
    st, n = stream_for(pred)
    if st is None:
        continue
    for it in range(3):
        ctx.render_rows(cam, fb.data_ptr(), 0, 0, H, st.value); ctx.finish()
    t0 = time.perf_counter()
    for it in range(20):
        ctx.render_rows(cam, fb.data_ptr(), 0, 0, H, st.value); ctx.finish()
    dt = (time.perf_counter() - t0) / 20
    print(f"{name:24s} {n:4d} CUs  {dt*1e3:.3f} ms/frame  CU-ms {dt*1e3*n:.1f}", flush=True)
    hip.hipStreamDestroy(st)
