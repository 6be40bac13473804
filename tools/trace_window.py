"""development: a window of a rocprofv3 kernel trace around the n-th k_primary launch, per queue.
python tools/trace_window.py <dir> [n] [count]"""
import csv, glob, re, sys
d = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 120
cnt = int(sys.argv[3]) if len(sys.argv) > 3 else 50
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_primary<false, false" in r["Kernel_Name"]]
i0 = idx[min(n, len(idx) - 1)]
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0 - 3:i0 + cnt]:
    nm = re.sub(r"\(.*", "", r["Kernel_Name"]); nm = re.sub(r"void |rt_\w+::", "", nm)[:28]
    s = (int(r["Start_Timestamp"]) - t0) / 1e3; e = (int(r["End_Timestamp"]) - t0) / 1e3
    print(f"{nm:30s} q{r['Queue_Id']} {s:8.1f} {e:8.1f} {e - s:7.1f}")
