"""development: idle time of the render queue in a rocprofv3 --kernel-trace run of bench.py.
python tools/queue_gaps.py <dir>: per kernel kind, the median gap between the end of the queue's previous kernel and its start."""
import csv, glob, re, sys, statistics as st, collections
d = sys.argv[1]
kt = sorted(csv.DictReader(open(glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0])), key=lambda r: int(r["Start_Timestamp"]))
prev = {}
gaps = collections.defaultdict(list); durs = collections.defaultdict(list)
n = len(kt)
for r in kt[n // 3: n - 50]:
    q = r["Queue_Id"]; s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    nm = re.sub(r"\(.*", "", r["Kernel_Name"]); nm = re.sub(r"void |rt_\w+::", "", nm)[:20]
    if q in prev:
        gaps[(q, nm, prev[q][1])].append((s - prev[q][0]) / 1e3)
    durs[(q, nm)].append((e - s) / 1e3)
    prev[q] = (e, nm)
for k, v in sorted(gaps.items()):
    print(f"q{k[0]} {k[1]:22s} after {k[2]:22s} n {len(v):4d} gap median {st.median(v):7.1f} us")
for k, v in sorted(durs.items()):
    print(f"q{k[0]} {k[1]:22s} n {len(v):4d} dur median {st.median(v):7.1f} us")
