import re
#!/usr/bin/env python3
"""Per-kernel timeline of a frame that had the GPU to itself in a rocprofv3 --kernel-trace directory of bench.py: the
last frame but one of the kernels compiled into the library for the reference's scene shape (the one-frame-at-a-time
section of the bench; any frame if there is no such kernel in the trace)."""
import csv, glob, sys
d = sys.argv[1]
import os
rows = list(csv.DictReader(open(max(glob.glob(d + "/**/*kernel_trace.csv", recursive=True), key=os.path.getsize))))
ks = sorted((r for r in rows if "k_" in r["Kernel_Name"]), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(ks) if "k_primary<" in r["Kernel_Name"] and not re.search(r"k_primary<\w+, true,", r["Kernel_Name"])]
own = [i for i in idx if re.search(r"k_primary<\w+, \w+, \w+, 1", ks[i]["Kernel_Name"])]
idx = own if len(own) >= 2 else idx
s, e = idx[-2], idx[-1]
t0 = int(ks[s]["Start_Timestamp"])
for r in ks[s:e]:
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("::")[-1][:28]
    print(f"{name:30s} start {(st - t0) / 1e3:9.1f}  dur {(en - st) / 1e3:8.1f} us  grid {r['Grid_Size_X']}x{r['Grid_Size_Y']} vgpr {r['VGPR_Count']}")
print("frame period", (int(ks[e]["Start_Timestamp"]) - t0) / 1e3, "us")
