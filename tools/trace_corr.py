"""development: launch-to-start delay of every k_primary in a rocprofv3 --kernel-trace --hip-runtime-trace run:
was the kernel already in its queue when the queue's previous kernel ended?  python tools/trace_corr.py <dir>"""
import csv, glob, re, sys
d = sys.argv[1]
kt = sorted(csv.DictReader(open(glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0])), key=lambda r: int(r["Start_Timestamp"]))
api = {r["Correlation_Id"]: r for r in csv.DictReader(open(glob.glob(d + "/**/*hip_api_trace.csv", recursive=True)[0]))}
prev_end = {}
out = []
for r in kt:
    q = r["Queue_Id"]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if "k_primary<false, false" in r["Kernel_Name"]:
        a = api.get(r["Correlation_Id"])
        if a:
            out.append((s, (s - int(a["Start_Timestamp"])) / 1e3, (s - prev_end.get(q, s)) / 1e3))
    prev_end[q] = e
for s, since_launch, since_prev in out[150:175]:
    print(f"k_primary: started {since_launch:8.1f} us after its hipLaunchKernel call, {since_prev:7.1f} us after the previous kernel of its queue ended")
